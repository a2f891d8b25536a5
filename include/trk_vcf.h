/*
 * trk_vcf.h -- C ABI of the native VCF / BGZF reader in libtrk.so.
 *
 * it replaces, for the hot path, what the reference gets from
 * cyvcf2/htslib (trtools/utils/utils.py:19-67 LoadSingleReader -> cyvcf2.VCF;
 * Variant.genotype.array() consumed at tr_harmonizer.py:860-862; Variant.format(key) consumed
 * through tr_harmonizer.py:561-588) -- decoding straight into the packed batch layout of
 * trk_batch (include/trk.h) instead of one Python object per sample field:
 *
 *   gt      int16 [n, S, P]   allele indices, -1 missing haplotype, -2 ploidy padding
 *   phased  uint8 [n, S]      cyvcf2's phase column ('|' seen in the GT)
 *   planes  int32 / float32 [n, S, ncol]   selected FORMAT fields, cyvcf2 conventions:
 *           Integer missing = INT_MIN, short vectors padded with INT_MIN + 1, Float missing = NaN
 *           (floats go text -> double -> float, as htslib does)
 *
 * BGZF blocks are inflated and records are parsed by a pool of host threads.  Host code only.
 */
#ifndef TRK_VCF_H
#define TRK_VCF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trk_vcf trk_vcf;

enum {
    TRK_VCF_INT = 0,        /* comma separated integers                                      */
    TRK_VCF_FLOAT = 1,      /* comma separated floats                                        */
    TRK_VCF_INT_RANGES = 2, /* integers separated by ',' or '-'  (GangSTR REPCI 'lo-hi,lo-hi') */
    TRK_VCF_MINSUPP = 3     /* HipSTR: min over the GB alleles of the ALLREADS read count
                               ('len|count;...'), 0 when ALLREADS is missing; 1 column
                               (reference dumpSTR/filters.py:519-567)                        */
};

/* Open a plain / gzip / bgzip VCF and read its header.  n_threads <= 0: all cores. */
int trk_vcf_open(const char* path, int n_threads, trk_vcf** out);
void trk_vcf_close(trk_vcf* v);
const char* trk_vcf_last_error(trk_vcf* v);  /* v may be NULL: last open error */
/* header text ('##...' lines and the '#CHROM' line, '\n' separated) */
const char* trk_vcf_header(trk_vcf* v, size_t* len);
int trk_vcf_n_samples(trk_vcf* v);
const char* trk_vcf_sample_name(trk_vcf* v, int i);

/* Continue reading at a BGZF virtual offset (compressed block offset << 16 | offset inside the
 * inflated block) -- the positions a tabix index (.tbi) holds; what cyvcf2's vcf(region) does
 * through htslib (statSTR.py:568-570, load_and_filter_genotypes.py:126-128).  BGZF files only. */
int trk_vcf_seek(trk_vcf* v, uint64_t voffset);

/* Contiguous shard of the records for one of `world` readers of the same file (one process per GPU; SURVEY 8(e):
 * "contiguous locus ranges per rank").  The compressed file is cut at the BGZF block boundaries nearest to
 * rank / world of its size (a plain-text file at those byte offsets); a record belongs to the rank in whose range
 * its line STARTS, so the ranks' batches, concatenated in rank order, are the records of the file in file order, and
 * each rank inflates its own range plus at most the block or two in which its last line ends.  The header stays as
 * read by trk_vcf_open.  Call once, before the first trk_vcf_read_batch.  *begin_off / *end_off: the file offsets of
 * the range.  Returns non-zero (and changes nothing) for inputs that cannot be cut (plain gzip).
 * trk_vcf_counters: {inflated bytes, compressed bytes, BGZF blocks} consumed since the shard was set. */
int trk_vcf_shard(trk_vcf* v, int rank, int world, uint64_t* begin_off, uint64_t* end_off);
void trk_vcf_counters(trk_vcf* v, uint64_t out[3]);

/* Ask for a FORMAT field to be decoded into a plane; returns the plane index (>= 0) or < 0. */
int trk_vcf_select_format(trk_vcf* v, const char* key, int kind, int ncol);

typedef struct {
    int32_t n_records;       /* records decoded by this call (0 = end of file)              */
    int32_t max_ploidy;      /* P of the gt tensor                                          */
    int16_t* gt;             /* [max_records, S, P]  caller-owned                            */
    uint8_t* phased;         /* [max_records, S]     caller-owned, may be NULL              */
    uint8_t* locus_ploidy;   /* [max_records]        caller-owned                            */
    void** planes;           /* [n_selected] caller-owned buffers [max_records, S, ncol]    */
    /* the record lines themselves (for the fixed columns and anything not decoded above):
     * text[line_off[i] .. line_end[i]) is record i (no newline; text[line_end[i]] itself is the line's '\n' or '\r' and
     * readable -- the reader appends a newline to a last line without one -- which trk_vcf_dumpstr_records relies on);
     * field_off[i*10 + k] is the offset inside the line of column k (k = 0..8: CHROM..FORMAT,
     * k = 9: first sample column).  Owned by the reader; valid during the NEXT trk_vcf_read_batch
     * call too (a thread may read batch n + 1 while batch n is in use), gone with the one after. */
    const char* text;
    const int64_t* line_off;
    const int64_t* line_end;
    const int32_t* field_off;
    int16_t* gt_mapped;      /* [max_records, n_out, P] caller-owned or NULL: the genotypes once more, sample s in
                                column map[s] (trk_vcf_set_sample_map), unmapped columns no-calls             */
} trk_vcf_batch;

/* statSTR.py:520-542 (--samples): group membership belongs to the SAMPLE, so the columns can be laid out by sample class
 * while the record is parsed -- for free -- and every class becomes a column range the ungrouped count kernels read
 * (trk_batch.class_runs).  map[s] = output column of the file's sample s, or -1 (the sample is in no group: never
 * written); n_out = columns of a gt_mapped row (class ranges padded to 16-byte boundaries, the row to 128).  NULL map:
 * off.  The reader keeps filling trk_vcf_batch.gt in file order as well (the per-record paths read that one). */
int trk_vcf_set_sample_map(trk_vcf* v, const int32_t* map, int32_t n_out);

/* Decode up to max_records records.  Returns 0, or a non-zero code with trk_vcf_last_error(). */
int trk_vcf_read_batch(trk_vcf* v, int max_records, int max_ploidy, trk_vcf_batch* out);

/* Round 4: the sample columns left to the caller.  With trk_vcf_skip_samples(v, 1) trk_vcf_read_batch finds the lines,
 * the fixed columns and the FORMAT keys of every record and stops there: gt / phased / locus_ploidy / planes / gt_mapped
 * are NOT written.  trk_vcf_format_idx returns [n_records][stride] int8, stride = 1 + selected planes: the index of GT,
 * then of every selected plane's key, among the record's FORMAT keys (-1: absent) -- what trk_parse_samples
 * (include/trk.h) needs beside the text and the offsets to parse the sample columns on the device; valid as long as the
 * batch's line tables.  trk_vcf_parse_samples fills the arrays of the batch just read on the host after all (the
 * fallback for records the device flags; errors as trk_vcf_read_batch's). */
int trk_vcf_skip_samples(trk_vcf* v, int on);
/* The reader's two text buffers continue in the caller's memory, `cap_each` bytes each (pinned pages, so that the upload
 * of a batch's text is a plain DMA).  Between two batches only: the text of the batch read last moves (its pointers go
 * stale).  The memory stays the caller's: never freed or reallocated here; a batch that outgrows it moves the reader back
 * to memory of its own.  0: done (or already so); 1: the bytes held now do not fit. */
int trk_vcf_set_text_buffers(trk_vcf* v, void* a, void* b, size_t cap_each);
const int8_t* trk_vcf_format_idx(trk_vcf* v, int32_t* stride);
int trk_vcf_parse_samples(trk_vcf* v, trk_vcf_batch* b);

/* ---- BGZF members inflated by the caller (round 5: on the device, trk_inflate_blocks of include/trk.h) --------------
 * With a hook installed trk_vcf_read_batch no longer inflates: for every run of complete BGZF members it has read it
 * calls hook->inflate with the compressed bytes and the members' payloads; the hook inflates them wherever it likes (the
 * text of the run is bytes [abs_base, abs_base + total) of the file's text, counted from the byte `seed` was given
 * first) and gives back ONLY what the host side of a batch reads:
 *   - the newlines: *nl = offsets relative to abs_base, ascending, bit 63 set when the byte before is '\r' -- for a
 *     newline at offset 0 that byte is the LAST byte of the text handed over before (the run before's, or the seed's:
 *     a CRLF pair cut by a run's boundary); the array stays the hook's, valid until its next call;
 *   - the HEADS of the lines -- every byte from a line's start up to and including its ninth tab (CHROM ... FORMAT)
 *     -- copied to their places in `out` (out[i] = text byte abs_base + i); the sample columns are NOT written: in
 *     this mode trk_vcf_batch.text is valid in the heads only (trk_vcf_skip_samples must be on);
 *   - *line_state (in / out): the tabs seen so far (0 ... 9) in the line that is unfinished at the end of the text.
 * seed(text, n) is called once, when the hook is installed, with the text the reader has already inflated itself
 * (what trk_vcf_open read beyond the header).  Returns: 0, else the read fails with that code in the error text.
 * Whoever needs a batch's sample columns on the host (trk_vcf_parse_samples, the record writers' host paths) must put
 * them there first: trk_vcf_text_abs says where trk_vcf_batch.text[0] lies in the stream.  Not with trk_vcf_seek /
 * trk_vcf_shard (both return an error while a hook is installed). */
typedef struct {
    uint64_t payload_off;   /* of the member's raw DEFLATE stream in `comp`          */
    uint32_t payload_len;
    uint32_t isize;         /* bytes of text the member holds (<= 65536)              */
    uint64_t dst;           /* where that text goes, relative to abs_base             */
} trk_vcf_iblock;
typedef struct {
    void* user;
    int (*seed)(void* user, const char* text, size_t n_bytes);
    int (*inflate)(void* user, const unsigned char* comp, size_t comp_bytes, const trk_vcf_iblock* blocks, int n_blocks,
                   uint64_t abs_base, size_t total, char* out, int* line_state, const uint64_t** nl, size_t* n_nl);
    int32_t max_members;    /* > 0: at most this many members per call (a device inflates a member per wave: the run that
                               fills it once, no more, is the fastest -- 16 x the CUs on gfx950); 0: whatever was read */
    /* Optional, both or neither: `inflate` in two halves, so that the reader can keep TWO runs in flight -- it reads run
     * k + 1 from the file and submits it (the hook copies `comp` before it returns and starts inflating) BEFORE it
     * collects run k, whose results are what `inflate` returns.  Runs are collected in the order they were submitted;
     * abs_base counts on from run to run.  With these set, `inflate` is not called.                                  */
    int (*submit)(void* user, const unsigned char* comp, size_t comp_bytes, const trk_vcf_iblock* blocks, int n_blocks,
                  uint64_t abs_base, size_t total);
    int (*collect)(void* user, char* out, int* line_state, const uint64_t** nl, size_t* n_nl);
} trk_vcf_inflate_hook;
int trk_vcf_set_inflate_hook(trk_vcf* v, const trk_vcf_inflate_hook* hook);
uint64_t trk_vcf_text_abs(trk_vcf* v);

/* ---- record serialisation ---------------------------------------------------------------
 * what the reference gets from cyvcf2.Writer.write_record (htslib's
 * vcf_format) for the records dumpSTR rewrites (dumpSTR.py:684, 721-746, 1338).  The per-sample
 * columns of ONE record are serialised from the typed FORMAT arrays cyvcf2 hands out
 * (Variant.genotype.array(), Variant.format(key)):
 *   GT     int16 [S, P+1]  -1 -> '.', -2 (ploidy padding) skipped, last column = phased ('|' or '/')
 *   INT    int32 [S, k]    INT_MIN -> '.', INT_MIN+1 ends the vector; an empty vector is '.'
 *   FLOAT  float32 [S, k]  NaN -> '.', a vector of NaNs is a single '.'; others as printf("%g")
 *   BYTES / UCS4  fixed-width strings (numpy 'S<n>' / '<U<n>'), NUL padded; empty -> '.'
 *   CALLFILTER    dumpSTR's FORMAT/FILTER text built from the call-filter mask (dumpSTR.py:648-683): 'NOCALL' when
 *          bit 31 is set, 'PASS' for 0, else '<name k>_<%g of values[k][s]>' of every set bit k, comma joined
 * Output: for every sample '\t' + the columns joined by ':' (no newline).  Returns the number of
 * bytes written, or -(bytes needed) when cap is too small, or INT64_MIN for a bad argument.     */
enum { TRK_VCF_COL_GT = 0, TRK_VCF_COL_INT = 1, TRK_VCF_COL_FLOAT = 2, TRK_VCF_COL_BYTES = 3, TRK_VCF_COL_UCS4 = 4,
       TRK_VCF_COL_CALLFILTER = 5 };
typedef struct {
    const uint32_t* mask;           /* [S] trk_call_out.filter_mask row of the record                 */
    int32_t n_filters;
    int32_t reserved;
    const char* const* names;       /* [n_filters] filter names                                        */
    const double* const* values;    /* [n_filters] -> [S] offending values; NULL rows for unfired bits */
} trk_vcf_callfilter;
typedef struct {
    int32_t kind;
    int32_t ncol;      /* values per sample (GT: P + 1); strings: 1                       */
    int32_t itemsize;  /* strings: bytes per sample                                       */
    int32_t reserved;
    const void* data;  /* [S, ncol] C-contiguous; CALLFILTER: a trk_vcf_callfilter        */
} trk_vcf_column;
int64_t trk_vcf_format_samples(int32_t n_samples, int32_t n_columns, const trk_vcf_column* cols, char* out,
                               int64_t cap);

/* ---- typed decode of every FORMAT field of one record ----------------------------------
 * What cyvcf2's Variant.format(key) returns for the fields the batch reader was not asked to
 * select (dumpSTR touches ALL of them when it nulls a filtered call, dumpSTR.py:730-746):
 *   INT    int32  [S, k]  '.' -> INT_MIN, short vectors padded with INT_MIN + 1, k = longest vector
 *   FLOAT  float32 [S, k] '.' and padding -> NaN (text -> double -> float)
 *   UCS4   [S] strings of ncol code points, NUL padded (numpy '<U<ncol>'); ncol >= 1
 * `samples` is the tab separated sample columns of the record line (text[field_off[9] ..]);
 * a sample with fewer fields than FORMAT reads as '.' for the rest.  Two calls: pass 0 fills
 * `ncol` of every field whose kind is INT / FLOAT / UCS4 (other kinds are skipped), the caller
 * allocates `out`, pass 1 fills the arrays.  Returns 0; 1 when the number of sample columns is
 * not n_samples; 2 for a token that is not a number (the caller falls back to its own parser
 * for the error text); -1 for bad arguments.                                                  */
typedef struct {
    int32_t kind;
    int32_t ncol;
    void* out;
} trk_vcf_decode;
int trk_vcf_decode_formats(const char* samples, int64_t len, int32_t n_samples, int32_t n_fields,
                           trk_vcf_decode* fields, int32_t pass);

/* ---- batch harmonisation + statSTR rows -------------------------------------------------------
 * The host side of the per-locus hot path without one Python object per record ("one call per batch
 * of loci").  For the records of a batch as trk_vcf_read_batch left them:
 *
 * trk_vcf_harmonize  what the reference derives per record in _HarmonizeHipSTRRecord / _HarmonizeGangSTRRecord /
 *   _HarmonizeAdVNTRRecord and TRRecord.__init__ (tr_harmonizer.py:303-408, 693-773): upper-cased alleles, HipSTR's
 *   flank trim REF[START-POS : END-POS+1-len(REF)] (Python slice semantics), allele lengths len(allele) / len(motif)
 *   (HipSTR: PERIOD, GangSTR / adVNTR: len(RU)) -- and from them the class tables of trk_batch (include/trk.h:
 *   allele_off, len_class, str_class, len_class_value) plus the sorted distinct sequences (the keys of
 *   GetAlleleCounts, tr_harmonizer.py:1495-1499).  Records it does not cover (a missing mandatory INFO field,
 *   symbolic alleles) get status 1 and n_python counts them: the caller then runs the
 *   batch through the Python harmoniser, which raises the reference's errors.
 * trk_vcf_statstr_rows  the text of statSTR's output rows (statSTR.py:586-629) from the device results of the batch:
 *   chrom, POS, POS + len(ref allele), then per enabled statistic and sample group the columns in the reference's
 *   order and formats ('{:.<precision>}' floats, 'nan', 'key:%.3f' / 'key:%i' lists in sorted key order, keys being
 *   sequences or str(numpy.float64) lengths).  err_kind: 1 / 2 the reference raises ValueError / IndexError in the
 *   HWE test of locus err_locus, 3 a genotype index beyond the record's alleles.
 * All returned pointers are owned by the reader and stay valid until its next trk_vcf_read_batch / harmonize call. */
enum { TRK_VT_GANGSTR = 0, TRK_VT_HIPSTR = 1, TRK_VT_ADVNTR = 2, TRK_VT_EH = 3, TRK_VT_POPSTR = 4 };
typedef struct {
    int32_t n_records;
    int32_t n_python;              /* records with status != 0                                            */
    int64_t n_alleles_total;
    const int32_t* allele_off;     /* [n + 1]                                                             */
    const uint16_t* len_class;     /* [sumA]                                                              */
    const uint16_t* str_class;     /* [sumA]                                                              */
    const double* len_class_value; /* [sumA] length of length-class c of locus l at allele_off[l] + c     */
    const double* allele_len;      /* [sumA] length by allele INDEX (GetLengthGenotypes' LUT)             */
    const int64_t* pos;            /* [n] VCF POS                                                         */
    const int64_t* end;            /* [n] POS + len(ref allele)                                           */
    const uint8_t* passing;        /* [n] FILTER is '.' or 'PASS'                                          */
    const uint8_t* status;         /* [n] 0 harmonised here, 1 needs the Python harmoniser                */
    const char* keys;              /* distinct upper-cased sequences, locus by locus in sorted order      */
    const int64_t* key_off;        /* [sumA + 1] key c of locus l: keys[key_off[allele_off[l] + c] .. +1)  */
    const int32_t* n_str_classes;  /* [n]                                                                 */
    const int32_t* n_len_classes;  /* [n]                                                                 */
    const int32_t* hrun;           /* [n] longest homopolymer run of REF (utils.GetHomopolymerRun)        */
    const int32_t* period;         /* [n] INFO PERIOD as an integer, INT32_MIN where the record has none  */
    const int64_t* tr_pos;         /* [n] TRRecord.pos: INFO START for HipSTR (tr_harmonizer.py:407), else POS */
} trk_vcf_harmonized;
/* (the tables are the reader's: valid during the NEXT trk_vcf_harmonize call too, gone with the one after -- as a batch's text) */
int trk_vcf_harmonize(trk_vcf* v, const trk_vcf_batch* b, int vcftype, trk_vcf_harmonized* out);

enum { TRK_SS_THRESH = 1, TRK_SS_AFREQ = 2, TRK_SS_ACOUNT = 4, TRK_SS_NALLELES = 8, TRK_SS_HWEP = 16, TRK_SS_HET = 32,
       TRK_SS_ENTROPY = 64, TRK_SS_MEAN = 128, TRK_SS_MODE = 256, TRK_SS_VAR = 512, TRK_SS_NUMCALLED = 1024 };
typedef struct {
    int32_t n_groups, precision, use_length, flags;
    const int32_t* allele_count;   /* host copies of trk_stats_out: [G, sumA]                              */
    const int32_t* locus_int;      /* [G, n, 12]                                                           */
    const double* locus_f64;       /* [G, n, 12]                                                           */
} trk_vcf_statstr;
/* Returns the number of bytes written, -(bytes needed) when cap is too small, INT64_MIN for bad arguments.  `skip`
 * ([n], may be NULL): records left out (--only-passing).                                                       */
int64_t trk_vcf_statstr_rows(const trk_vcf_batch* b, const trk_vcf_harmonized* h, const trk_vcf_statstr* in,
                             const uint8_t* skip, char* out, int64_t cap, int32_t* err_locus, int32_t* err_kind);

/* ---- dumpSTR's output records, a batch at a time ------------------------------------------------
 * For every record of the batch (as trk_vcf_read_batch left it) whose head is not NULL: the caller's head text (the
 * nine leading columns of the OUTPUT line: FILTER and INFO rewritten, FORMAT with ':FILTER' appended -- dumpSTR.py:
 * 917-973, 1304-1336) followed by the per-sample columns re-serialised from the typed FORMAT arrays exactly as the
 * per-record path does (trk_vcf_decode_formats -> nulling of filtered calls, dumpSTR.py:721-746 -> trk_vcf_format_samples
 * with the FORMAT/FILTER column built from the call-filter mask, dumpSTR.py:648-683), and a newline.  Records are
 * independent: one per host-thread task, output in record order.
 *   mask8 / mask32     [n, S] trk_call_out.filter_mask8 / filter_mask of the batch (one of them)
 *   gt, phased, locus_ploidy  the batch's genotype tensor [n, S, ploidy], phase bytes [n, S], per-record ploidy
 *   filters[k]         name of call filter k and where the number behind '<name>_<value>' comes from: kind 0 the value
 *                      of column col_a of plane_a ([n, S, ncol_a], dtype 0 int32 / 1 float32) as a double, kind 1 that
 *                      value divided by plane_b's (HipSTR flank-indel / stutter ratios, filters.py:444-449), kind 2
 *                      columns col_a + col_a2 of plane_a (GangSTR QEXP[1] + QEXP[2] as a float32 sum, filters.py:668-
 *                      672; RC[1] + RC[3], filters.py:717-721), kind 3 GangSTR's bad confidence interval: plane_a the
 *                      REPCN columns, plane_b the pre-parsed REPCI (lo, hi per haplotype), the value is REPCN of the
 *                      first haplotype whose interval excludes it (filters.py:744-756), kind 4 PopSTR's
 *                      require-support: plane_a the AD columns (one per allele), col_a the threshold, the value is
 *                      the read support of the last haplotype whose allele has fewer reads (filters.py:858-867)
 *   format_keys/kinds  the header's FORMAT IDs and how each is decoded (TRK_VCF_COL_INT / _FLOAT / _UCS4); IDs not
 *                      listed decode as strings
 * Returns the bytes written; -(bytes needed) when cap is too small; INT64_MIN for bad arguments; INT64_MIN + 1 when
 * a record is outside what this path covers (*err_record: a FORMAT/FILTER field already present, no GT, a token that
 * is not a number ...): the caller then takes the per-record path for the batch.                                  */
typedef struct {
    const char* name;
    int32_t kind, reserved;
    const void* plane_a;
    int32_t dtype_a, ncol_a, col_a, col_a2;   /* col_a2: the second column of kind 2 */
    const void* plane_b;
    int32_t dtype_b, ncol_b, col_b, pad_b;
} trk_vcf_cf_value;
typedef struct {
    int32_t n_samples, ploidy, n_filters, n_format_keys, n_threads, reserved;
    const int16_t* gt;
    const uint8_t* phased;
    const uint8_t* locus_ploidy;
    const uint8_t* mask8;
    const uint32_t* mask32;
    const trk_vcf_cf_value* filters;
    const char* const* heads;        /* [n] NUL-terminated; NULL = the record is not written */
    const char* const* format_keys;
    const int32_t* format_kinds;
} trk_vcf_dumpstr;
int64_t trk_vcf_dumpstr_lines(const trk_vcf_batch* b, const trk_vcf_dumpstr* in, char* out, int64_t cap,
                              int32_t* err_record);

/* trk_vcf_dumpstr_records (round 4): the same records with NOTHING per record from the caller.
 *   heads      base.heads may be NULL (or hold NULL entries): the nine leading columns are then built here -- columns
 *              0-5 as read, FILTER = filter_text[l] (NULL: as read), INFO rewritten as vcfio.rewrite_info /
 *              _rewrite_info_general do (every declared Integer / Float value re-serialised the way htslib writes it
 *              back, a Flag as its bare key, the updates HRUN, HET, HWEP, AC, REFAC of dumpSTR.py:1304-1336 in place
 *              or appended in that order), FORMAT + ':FILTER'.  keep[l] == 0: the record is not written
 *              (--drop-filtered).  An INFO column this rewrite does not cover (a key twice, a number only Python's
 *              int() / float() take) sets need_head[l] = 1 and the call returns INT64_MIN + 2: the caller supplies
 *              those heads in base.heads and calls again.
 *   samples    fast_path != 0: the sample columns are first tried WITHOUT decoding (fast_samples in trk_vcf.cpp:
 *              tokens that '%g' / integer formatting would print back unchanged are copied as bytes, the others
 *              re-serialised one by one, filtered calls nulled as dumpSTR.py:721-746 does); a record the pass cannot
 *              prove equal to the decode -> null -> format path takes that path.  TRK_FMT_FAST=0 disables it. */
typedef struct {
    trk_vcf_dumpstr base;
    const uint8_t* keep;                 /* [n] or NULL (all records are written)                              */
    const char* const* filter_text;      /* [n] or NULL; NULL entries keep the column as read                   */
    const int32_t* hrun;                 /* [n] INFO HRUN                                                       */
    const uint8_t* have_stats;           /* [n] the locus has called samples left: HET / HWEP / AC / REFAC from
                                            the arrays below, otherwise -1 / -1 / zeros / 0                     */
    const double* het;                   /* [n]                                                                 */
    const double* hwep;                  /* [n]                                                                 */
    const int32_t* allele_count;         /* counts of the masked genotypes by allele index, [allele_off[n]]     */
    const int32_t* allele_off;           /* [n + 1]                                                             */
    int32_t n_info_keys, fast_path;
    const char* const* info_keys;        /* the header's INFO IDs ...                                           */
    const int32_t* info_kinds;           /* ... 0 String, 1 Integer, 2 Float, 3 Flag                            */
    uint8_t* need_head;                  /* [n] out (zeroed by the caller), or NULL                             */
    /* Round 4, optional (all four or none): the sample columns of the records as trk_format_samples (include/trk.h) wrote
     * them on the device, copied to the host -- record l's columns are dev_regions[dev_region_off[l] .. + dev_region_len[l]);
     * dev_flags[l] != 0: the device left the record to this writer.  The head is built here either way.            */
    const char* dev_regions;
    const int64_t* dev_region_off;
    const uint32_t* dev_region_len;
    const uint8_t* dev_flags;
    /* optional: the copy of dev_regions to the host may still be in flight when the call starts -- the writer builds the
     * heads first and calls dev_wait(dev_wait_arg) (e.g. trk_sync with its context; non-zero: the call fails with
     * INT64_MIN + 1, err_record -1) before it reads the first byte of dev_regions                                        */
    int (*dev_wait)(void* arg);
    void* dev_wait_arg;
    /* Round 5, optional -- WHOLE-RECORD EMIT: instead of columns that were copied to the host somewhere (dev_regions) and
     * are gathered behind their heads here, the device puts them where they belong in `out`.  With dev_emit set
     * (dev_regions may be NULL; dev_region_len / dev_flags as above) the writer builds the heads and the records the device
     * left to it, lays the batch out -- record l at out[at[l]], its head of head_len[l] bytes first -- and calls
     *   dev_emit(dev_emit_arg, rec_off, total, out)
     * once: rec_off[l] = at[l] + head_len[l] for a record whose columns the device holds, -1 for the others; the callee
     * writes the dev_region_len[l] bytes of those columns at out[rec_off[l]] and may write ANY byte of out[0, total)
     * besides (one copy of a device buffer laid out the same way is the intended use): the heads, the newlines and the
     * records written here are put in afterwards.  Non-zero: the call fails with INT64_MIN + 1, err_record -1.          */
    int (*dev_emit)(void* arg, const int64_t* rec_off, int64_t total, char* out);
    void* dev_emit_arg;
} trk_vcf_dumpstr2;
/* The FORMAT keys of every record of the batch as trk_format_samples wants them: kinds16 [n][16] (1 GT, 2 Integer,
 * 3 Float, 4 String, by in->format_keys / format_kinds; unlisted keys are strings), n_fields [n] -- 0 for a record the
 * device must not format (no GT, a FILTER field of its own, more than 16 keys, a vector-typed kind).                 */
int trk_vcf_format_kinds(const trk_vcf_batch* b, const trk_vcf_dumpstr* in, uint8_t* kinds16, uint8_t* n_fields);
int64_t trk_vcf_dumpstr_records(const trk_vcf_batch* b, const trk_vcf_dumpstr2* in, char* out, int64_t cap,
                                int32_t* err_record);
/* records written so far by this process: without decoding / through the decode path / with a caller-built head */
void trk_vcf_dumpstr_stats(int64_t* fast, int64_t* decoded, int64_t* caller_heads);

/* ---- BGZF output (round 6): dumpSTR --zip, reference dumpSTR.py:1241-1245 + 1347-1352 (`bgzip -f`, then `tabix`) ----
 * n bytes of text as consecutive BGZF members of at most 0xff00 bytes of text each (SAM specification section 4.1: gzip
 * members with the 'BC' extra field; every member is a DEFLATE stream of its own, CRC-32 and ISIZE behind it), compressed on
 * the caller-side worker pool (n_threads <= 0: the FMT_THREADS default) by libdeflate where the image has it, zlib else.
 * level 1 ... 9 (0: stored members).  No end-of-file member is written: the caller appends it when the file ends
 * (trk_bgzf_eof).  The members of [data, data + n) depend on the text and the level alone -- not on the thread count, not on
 * how a stream was cut into calls as long as every call but the last hands over a multiple of 0xff00 bytes.
 * Returns 0 and *out_bytes; 1: out_cap below trk_bgzf_bound(n); 2: the compressor failed. */
size_t trk_bgzf_bound(size_t n);
int trk_bgzf_compress(const void* data, size_t n, int level, int n_threads, void* out, size_t out_cap, size_t* out_bytes);
/* the 28-byte end-of-file member; returns its length */
size_t trk_bgzf_eof(void* out28);
/* the offsets of the newlines of text[0 .. n), ascending, into out[0 .. cap) (memchr on the worker pool: a writer that notes
 * where its records lie scans a 150 MB block in a couple of milliseconds); returns how many there are (more than cap: only
 * the first cap were written) */
int64_t trk_text_newlines(const void* text, size_t n, int64_t* out, size_t cap);
/* The record lines among the lines of text[0 .. n) whose newlines are nl[0 .. n_nl) (trk_text_newlines; a last line without
 * one ends at n), for a writer that builds the tabix index of what it writes: row k of out[cap][8] =
 *   { offset of the line, offset behind its newline, beg, end, offset of CHROM, its length, new, odd }
 * with [beg, end) the interval htslib's tabix gives a VCF line -- POS - 1 ... POS - 1 + len(REF), INFO/END when it lies beyond
 * beg --, new = 1 when CHROM differs from the record before (in this call), odd = 1 when the line's head is not what this
 * scanner reads (fewer than four columns, a POS or END that is not plain digits, a REF beyond ASCII): beg and end are then
 * unset and the caller reads that head itself.  Lines that are empty or begin with '#' are no records.  Returns the number of
 * records (more than cap: only the first cap rows were written). */
int64_t trk_text_record_places(const void* text, size_t n, const int64_t* nl, size_t n_nl, int64_t* out, size_t cap);
/* where the BGZF members of data[0 .. n) begin: out[k] = the offset of member k, k < cap, by the chain of their BSIZE fields
 * (what a writer that keeps virtual offsets walks after every block of members it made -- 9 000 steps per block on the device's
 * path: not a loop for an interpreter that holds its lock meanwhile).  Returns the number of members (more than cap: only
 * the first cap were written); -1: the chain leaves the buffer or meets something that is not a BGZF header. */
int64_t trk_bgzf_member_offsets(const void* data, size_t n, uint64_t* out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif
