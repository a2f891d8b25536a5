/*
 * trk.h -- C ABI of libtrk.so: the MI355X (gfx950) implementation of the
 * TRTools per-locus hot path (tr_harmonizer.TRRecord reductions -> statSTR
 * statistics / dumpSTR call filters, sample counters and locus filters).
 *
 * The reference (gymrek-lab/TRTools v6.1.0) is pure Python and has no FFI seam;
 * its "operator API" for this path is the Python surface
 *   trtools/utils/tr_harmonizer.py  TRRecord.Get{CalledSamples,CallRate,
 *       AlleleCounts,AlleleFreqs,GenotypeCounts,MaxAllele}      (:864-1575)
 *   trtools/utils/utils.py          Get{Heterozygosity,Entropy,Mean,Mode,
 *       Variance,HardyWeinbergBinomialTest}                      (:118-338)
 *   trtools/dumpSTR/filters.py      call-level Reason classes    (:327-867)
 *                                   locus-level Filter_* classes (:35-217)
 *   trtools/dumpSTR/dumpSTR.py      ApplyCallFilters :613-774, ApplyLocusFilters
 *                                   :917-973, INFO recompute :1304-1336.
 * Those methods are called once per VCF record; the entry points below are
 * their batched equivalents (one call per batch of L loci) and are what a
 * ctypes binding inside the reference would bind (INTEGRATION.md shows it).
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 (TRK_OK) or an
 *     error code, the message is available from trk_last_error(); nothing
 *     throws across the ABI.
 *   - all array pointers in trk_batch / trk_*_out are DEVICE pointers obtained
 *     from trk_dev_alloc (or any hipMalloc'ed memory of the same device),
 *     16-byte aligned, C-contiguous.  Host staging is explicit through
 *     trk_memcpy_h2d / trk_memcpy_d2h (PCIe cost is never hidden).
 *   - one in-flight call per trk_ctx; calls are asynchronous on the context's
 *     HIP stream, trk_sync() / trk_memcpy_d2h() synchronise.
 *   - genotype sentinels are cyvcf2's: -1 missing haplotype, -2 ploidy padding
 *     (tr_harmonizer.py:829-862).
 */
#ifndef TRK_H
#define TRK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trk_ctx trk_ctx;

enum {
    TRK_OK = 0,
    TRK_ERR_HIP = 1,     /* a HIP runtime call failed                            */
    TRK_ERR_ARG = 2,     /* invalid argument                                      */
    TRK_ERR_NOMEM = 3,   /* device allocation failed                              */
    TRK_ERR_RCCL = 4,    /* RCCL missing or a collective failed                   */
    TRK_ERR_DATA = 5     /* the data violates a reference invariant (see docs)    */
};

/* ---- context ------------------------------------------------------------ */

/* Create a context on HIP device `device` (one context per process per GPU). */
int trk_init(int device, trk_ctx** out);
void trk_free(trk_ctx* ctx);
/* Last error message of this context (ctx may be NULL: last init error).     */
const char* trk_last_error(trk_ctx* ctx);
/* 1 = HIP/gfx950.  There is no CPU backend in this library.                  */
int trk_backend(trk_ctx* ctx);
int trk_device_count(int* n);
int trk_device_info(trk_ctx* ctx, char* name, int name_len, int* n_cu,
                    uint64_t* hbm_bytes, char* arch, int arch_len);

/* ---- device memory / staging -------------------------------------------- */
int trk_dev_alloc(trk_ctx* ctx, size_t bytes, void** dptr);
int trk_dev_free(trk_ctx* ctx, void* dptr);
int trk_memcpy_h2d(trk_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int trk_memcpy_d2h(trk_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int trk_memcpy_d2d(trk_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes); /* async on the stream */
int trk_memset(trk_ctx* ctx, void* dst_dev, int value, size_t bytes);
int trk_sync(trk_ctx* ctx);
/* Pinned (page-locked) host staging memory: hipHostMalloc.  A copy from / to such a buffer runs at the full PCIe
 * rate and -- through the *_async forms -- overlaps with kernels of the other queue; pageable numpy buffers go
 * through the driver's bounce buffer and serialise.  The async forms enqueue on the selected queue and return;
 * trk_queue_sync(queue) / trk_sync() wait.  (compute.py keeps a ring of these per Engine: the reference streams
 * record -> row, statSTR.py:586-639; here batch n+1 uploads while batch n computes and batch n-1 is formatted.) */
int trk_host_alloc(trk_ctx* ctx, size_t bytes, void** hptr);
int trk_host_free(trk_ctx* ctx, void* hptr);
int trk_memcpy_h2d_async(trk_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int trk_memcpy_d2h_async(trk_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int trk_queue_sync(trk_ctx* ctx, int queue);
/* A context owns TRK_N_STREAMS in-order queues (HIP streams).  Every entry point enqueues on the selected
 * one (queue 0 after trk_init); trk_stream_wait makes `waiter` wait for what has been enqueued on `signal`
 * so far; trk_sync waits for all of them.  Each queue has its own finaliser scratch, so independent work
 * can overlap -- bench.py runs statSTR's finaliser (k_locus_finalize + k_hwe_test, latency bound) on
 * queue 1 beside dumpSTR's call-filter pass (HBM bound) on queue 0.  The associaTR entry points share one
 * workspace: use them from one queue at a time.                                                        */
#define TRK_N_STREAMS 4
int trk_stream_select(trk_ctx* ctx, int queue);
/* A queue for the CALLING thread alone: from now on this thread's copies and kernels (the entry points that take no
 * per-queue scratch: trk_memcpy_*, trk_parse_samples, trk_format_samples) go to `queue` whatever trk_stream_select says;
 * -1 returns the thread to the selected queue.  For a helper thread that feeds the device beside the caller's thread
 * (the command lines' reader uploads and parses batch n + 1 while batch n is counted, filtered and written): its 80 MB
 * uploads no longer sit in front of the caller's kernels.  Hand-over is the caller's business (the helper synchronises
 * its queue -- trk_queue_sync, or any blocking copy -- before the other thread touches what it made).                */
int trk_thread_queue(trk_ctx* ctx, int queue);
int trk_stream_wait(trk_ctx* ctx, int waiter, int signal);
/* Named ordering points for pipelines that run several batches deep: trk_event_record marks "everything enqueued so
 * far on the SELECTED queue"; trk_event_wait makes the selected queue wait for the mark of `slot` as last recorded
 * (never recorded: no wait).  Unlike trk_stream_wait the mark can be an old one -- a queue that recycles a double
 * buffer waits for the consumer of two batches ago, which has long finished, instead of the one just enqueued.    */
#define TRK_N_EVENTS 16
int trk_event_record(trk_ctx* ctx, int slot);
int trk_event_wait(trk_ctx* ctx, int slot);


/* Per-kernel profiling: when enabled every kernel launch is bracketed by HIP
 * events on the compute stream; trk_profile_get drains them (synchronises).  */
enum {
    TRK_K_LOCUS_COUNT = 0,   /* genotype-matrix reduction (allele histograms)  */
    TRK_K_LOCUS_FINALIZE = 1,
    TRK_K_CALL_FILTER = 2,
    TRK_K_LOCUS_FILTER = 3,
    TRK_K_SYNTH = 4,
    TRK_K_ASSOC_SCAN = 5,    /* associaTR: genotype x trait cross-products per locus */
    TRK_K_ASSOC_FINALIZE = 6,
    TRK_K_CF_REDUCE = 7,     /* k_cf_reduce: sums the call-filter pass's per-workgroup partial sample counters */
    TRK_K_COUNT = 8
};
int trk_profile_enable(trk_ctx* ctx, int on);
int trk_profile_get(trk_ctx* ctx, int kernel, int64_t* n_launches, double* total_ms);
int trk_profile_reset(trk_ctx* ctx);

/* ---- batch of loci -------------------------------------------------------
 * Replaces, for L records at once, what TRRecord.__init__ precomputes per
 * record (tr_harmonizer.py:693-773) plus cyvcf2's genotype.array():
 *   gt            allele INDICES, [L, S, P] int16 (phase column dropped)
 *   locus_ploidy  max ploidy of each record (vcfrecord.ploidy), <= P; columns
 *                 >= locus_ploidy[l] are ignored.  NULL -> all loci have P.
 *   allele_off    [L+1] prefix offsets into the per-allele tables
 *                 (A_l = allele_off[l+1]-allele_off[l] = 1 + #ALT)
 *   len_class     [sumA] rank of the allele's length among the locus's
 *                 distinct lengths, ascending      (GetLengthGenotypes :1239)
 *   str_class     [sumA] rank of the allele's (trimmed, upper-cased) sequence
 *                 among the locus's distinct sequences, in numpy '<U' order
 *                 (GetStringGenotypes :948-961)
 *   len_class_value [sumA] length (repeat units, float64) of length-class c of
 *                 locus l at allele_off[l]+c
 *   max_alleles   max_l A_l if the caller knows it (picks the LDS histogram size), else 0
 *   group_bits    [S] bit g set -> sample belongs to sample group g
 *                 (statSTR --samples, statSTR.py:520-542); NULL -> one group
 *                 holding every sample.  n_groups <= 8.
 *   row_stride    samples from the start of one row of gt to the start of the next; 0 = n_samples.  With gt
 *                 pointing at column c0 of a wider tensor the batch is a VIEW of the column range
 *                 [c0, c0 + n_samples) (trk_locus_stats' streaming count kernels only; every other entry wants 0).
 *   class_runs    sample groups without a per-call group lookup.  Group membership belongs to the SAMPLE, the same
 *                 for every locus: when the caller lays the sample columns out ordered by group-bit pattern
 *                 ("class"), every class is a contiguous column range, the ungrouped streaming kernel counts each
 *                 range (a view, row_stride) and the classes are added into their groups per locus afterwards --
 *                 any number of overlapping groups at the ungrouped kernel's rate.  HOST array
 *                 int32[4 * n_class_runs]: {first column, columns, real samples, group bits} per run; `first column`
 *                 and `columns` are multiples of four (16-byte rows), the columns beyond `real samples` are padding
 *                 whose genotypes MUST be -1.  group_bits (the same bits per column, 0 for padding) stays mandatory:
 *                 the kernels for batches outside the streaming path read it.  Engine.make_batch / compute.py
 *                 build this layout (trk_permute_columns does the gather on the device).
 */
typedef struct {
    int32_t n_loci;
    int32_t n_samples;
    int32_t ploidy;
    int32_t n_groups;
    int64_t n_alleles_total;
    int32_t max_alleles;   /* max A_l over the batch (sizes the LDS histogram); 0 = unknown */
    int32_t n_pad_samples; /* the last n_pad_samples samples of every row are padding: their genotypes MUST be -1
                              (no call) and they are left out of TRK_LI_N_SAMPLES.  Rows of a multiple of four
                              samples are 16-byte aligned, which the streaming kernels need: a cohort of 10001
                              samples is handed over as 10004 with n_pad_samples = 3 (compute.py does this; the
                              unpadded layout runs the per-call kernels, 4-6x slower) */
    const int16_t* gt;
    const uint8_t* locus_ploidy;
    const int32_t* allele_off;
    const uint16_t* len_class;
    const uint16_t* str_class;
    const double* len_class_value;
    const uint8_t* group_bits;
    int32_t row_stride;        /* 0 = n_samples */
    int32_t n_class_runs;      /* 0 = columns are not ordered by class */
    const int32_t* class_runs; /* HOST memory, int32[4 * n_class_runs] */
} trk_batch;

/* integer columns of trk_stats_out.locus_int ([G, L, TRK_LI_COLS] int32) */
enum {
    TRK_LI_N_CALLED = 0,     /* samples with no -1 haplotype (GetCalledSamples strict;
                                == sum(GetGenotypeCounts().values()), statSTR.py:426) */
    TRK_LI_N_LOWPLOIDY = 1,  /* of those, samples holding a -2 haplotype                */
    TRK_LI_N_HOM_LEN = 2,    /* num_hom of utils.py:327-333, alleles by length          */
    TRK_LI_N_HOM_STR = 3,    /* same, alleles by sequence                               */
    TRK_LI_N_ALLELES = 4,    /* sum of allele counts (called haplotypes)                */
    TRK_LI_N_BAD = 5,        /* calls holding an allele index >= A_l (reference: IndexError) */
    TRK_LI_HWE_STATUS_LEN = 6,
    TRK_LI_HWE_STATUS_STR = 7,
    TRK_LI_N_SAMPLES = 8,    /* samples in the group                                    */
    TRK_LI_NALLELES_LEN = 9, /* statSTR GetNAlleles, by length / by sequence            */
    TRK_LI_NALLELES_STR = 10,
    TRK_LI_COLS = 12
};
/* values of TRK_LI_HWE_STATUS_* */
enum {
    TRK_HWE_OK = 0,
    TRK_HWE_NAN = 1,          /* utils.py:323-332 returned nan                          */
    TRK_HWE_VALUE_ERROR = 2,  /* scipy binomtest would raise ValueError (n < 1)        */
    TRK_HWE_INDEX_ERROR = 3   /* haploid locus: IndexError at utils.py:331              */
};
/* float columns of trk_stats_out.locus_f64 ([G, L, TRK_LF_COLS] float64) */
enum {
    TRK_LF_THRESH = 0,   /* GetMaxAllele                    tr_harmonizer.py:1542 */
    TRK_LF_MEAN = 1,     /* utils.GetMean      (by length)  utils.py:215          */
    TRK_LF_MODE = 2,     /* utils.GetMode      (by length)  utils.py:238          */
    TRK_LF_VAR = 3,      /* utils.GetVariance  (by length)  utils.py:273          */
    TRK_LF_HET_LEN = 4,  /* utils.GetHeterozygosity         utils.py:142          */
    TRK_LF_HET_STR = 5,
    TRK_LF_ENTROPY_LEN = 6, /* utils.GetEntropy             utils.py:178          */
    TRK_LF_ENTROPY_STR = 7,
    TRK_LF_HWEP_LEN = 8, /* utils.GetHardyWeinbergBinomialTest utils.py:298       */
    TRK_LF_HWEP_STR = 9,
    TRK_LF_CALLRATE = 10,/* GetCallRate                    tr_harmonizer.py:921  */
    TRK_LF_COLS = 12
};

typedef struct {
    double nalleles_thresh;   /* statSTR --nalleles-thresh (statSTR.py:207)        */
    int32_t flags;            /* TRK_STATS_* */
    int32_t reserved;
} trk_stats_params;
enum {
    TRK_STATS_COUNT_ONLY = 1, /* skip the finaliser (allele_count + first 6 ints)  */
    TRK_STATS_TWIN = 2        /* out->allele_count is [2][G, sumA] and out->locus_int [2][G, L, TRK_LI_COLS]: the counts
                                 are stored twice, back to back, in the one pass over the tensor.  For the caller that lets
                                 trk_call_filters correct one copy in place (delta outputs, dumpSTR) and keeps the other
                                 for statSTR's own rows: no device-to-device copies between the two halves of a step.
                                 The finaliser (and locus_f64) see the first copy only.                              */
};

typedef struct {
    int32_t* allele_count;  /* [G, sumA] counts per allele INDEX
                               (GetAlleleCounts(index=True), :1420-1499)           */
    int32_t* locus_int;     /* [G, L, TRK_LI_COLS]                                 */
    double* locus_f64;      /* [G, L, TRK_LF_COLS]                                 */
} trk_stats_out;

/* (a3)(a5)-(a9) of SURVEY.md section 8: for every locus and sample group the
 * allele histogram and every statSTR statistic.
 * Ungrouped diploid batches of rows <= 2048 samples take two launches (the finaliser is
 * the count kernel's epilogue, the HWE tests a second kernel; same bits as the general
 * sequence count / finaliser / tests -- TRK_FUSED_STATS=0 in the environment selects the
 * latter); every other batch the general sequence.                                      */
int trk_locus_stats(trk_ctx* ctx, const trk_batch* in, const trk_stats_params* prm,
                    trk_stats_out* out);

/* The finaliser alone: float statistics + HWE test from allele_count / the first six
 * locus_int columns already present in `out` (after trk_locus_stats(COUNT_ONLY), possibly
 * corrected by trk_call_filters' delta outputs).                                        */
int trk_locus_finalize(trk_ctx* ctx, const trk_batch* in, const trk_stats_params* prm,
                       trk_stats_out* out);

/* ---- dumpSTR call-level filters ------------------------------------------ */
enum { TRK_DT_I32 = 0, TRK_DT_F32 = 1,
       TRK_DT_PLANAR = 0x100 /* or'ed in: data is [ncol, L, S] -- one contiguous [L, S] array per column.  Every
                                column then streams as 16-byte vectors like a single-column plane (GangSTR's
                                QEXP / RC / REPCN / REPCI: 4.3 ms instead of 7.5 ms for the nine-filter set at
                                50k x 5k, profiles/r01_notes.md)                                                */ };
typedef struct {
    const void* data;   /* device [L, S, ncol], or [ncol, L, S] with TRK_DT_PLANAR    */
    int32_t dtype;      /* TRK_DT_I32 / TRK_DT_F32 (| TRK_DT_PLANAR); int32 missing = INT_MIN, float32 missing = nan */
    int32_t ncol;
} trk_plane;
/* [n_cells, ncol] -> [ncol, n_cells] for 4-byte elements (device to device; a helper for callers whose FORMAT
 * planes are produced interleaved).                                                                           */
int trk_planarize(trk_ctx* ctx, const void* src, void* dst, int64_t n_cells, int32_t ncol);
/* Rows of row_words 4-byte words -> rows of row_words + pad_words words, the pad filled with `fill` (device to device,
 * src != dst).  The padding samples of trk_batch.n_pad_samples are appended this way after a dense upload: a diploid
 * genotype row or a FORMAT plane row is padded to a multiple of 32 samples so that EVERY row of the tensor starts on a
 * 128-byte boundary -- 10 000 samples are 40 000 bytes, every second row starts in the middle of a cache line and the
 * call-filter stream runs 3-4 % slower (profiles/r03_notes.md section 10).  fill: 0xffffffff for genotypes (two -1),
 * 0x80000000 for an int32 plane, a quiet nan (0x7fc00000) for a float32 plane.  The reference has no counterpart (its
 * rows are numpy arrays built per record, tr_harmonizer.py:1046-1090).                                              */
int trk_pad_rows(trk_ctx* ctx, const void* src, void* dst, int64_t n_rows, int32_t row_words, int32_t pad_words,
                 uint32_t fill);

/* Column gather of a genotype tensor (device to device): dst[l, j, :] = src[l, col[j], :] for j < n_dst, a column of
 * no-calls (-1) where col[j] < 0.  src [n_loci, n_src, ploidy] int16, dst [n_loci, n_dst, ploidy] int16, col DEVICE
 * int32[n_dst].  Lays a cohort out by sample class for trk_batch.class_runs. */
int trk_permute_columns(trk_ctx* ctx, const int16_t* src, int16_t* dst, const int32_t* col, int64_t n_loci,
                        int32_t n_src, int32_t n_dst, int32_t ploidy);


/* ---- the two output planes of a call-filter pass, placed -----------------------------------------
 * dumpSTR.py:613-774 (ApplyCallFilters) writes a masked genotype and a FILTER value per call: here two [n_loci,
 * n_samples] 4-byte planes written in lock step.  On MI355X that pair of write streams runs on one of two levels,
 * 10-18 % apart, decided by WHICH allocations the two planes are (profiles/r03_notes.md section 22, r04_notes.md
 * section 1: not by offsets, launch geometry or the inputs), stable while the allocations live.  trk_dev_alloc_pair
 * allocates the first plane, then candidates for the second one at a time, times the write-only half of the pass's
 * stream over (first, candidate) -- ~3 launches, 1-2 ms each at 4 GB planes -- and stops at the first pair that is
 * clearly fast (>= 6 % faster than another candidate, or >= TRK_PAIR_FAST_TBPS of write rate); the best pair is
 * returned, the other candidates freed.  When every neighbour is slow the search steps ahead: the spares are freed, a
 * spacer is allocated -- never touched --, one candidate is taken behind it, the spacer freed (TRK_PLACE_JUMP_GB: the
 * spacer sizes, default two jumps of 16 GB; transient = spacer + one plane, reported in peak_extra_bytes).  max_spare = FRESH planes that may exist beyond the two returned (0: plain
 * allocation + one probe; trk_call_filters' callers use 2): the transient never exceeds max_spare x bytes_each.
 * Both planes are plain device allocations (trk_dev_free); their contents are undefined. */
#define TRK_PAIR_MAX_PROBES 8
#define TRK_PAIR_FAST_TBPS 6.5
typedef struct {
    int32_t n_probed;              /* candidates timed                                                      */
    int32_t placed;                /* 1: the kept pair is on the fast level by one of the two criteria      */
    float probe_ms[TRK_PAIR_MAX_PROBES];  /* write-only probe of (a, candidate k)                           */
    float kept_ms;                 /* the kept pair's                                                        */
    int32_t have_a, have_b;        /* index in have[] of the plane returned as *a / *b, -1: a fresh allocation */
    int32_t n_jumps;               /* candidates taken behind a spacer (TRK_PLACE_JUMP_GB: sizes in GB, default "16,16") */
    double seconds;                /* host time the call took (allocations + probes)                        */
    uint64_t peak_extra_bytes;     /* freshly allocated beyond the two returned planes, at the peak         */
    int32_t reserved;              /* 1: the planes are the context's reserved pair (trk_reserve_pair)      */
    int32_t pad_;
} trk_pair_info;
/* RESERVING the pair.  Which level a pair of planes is on follows from where the two allocations lie in the device's
 * memory -- three classes of regions, a pair inside one class is slow, a pair across two is fast (the 16-plane matrix
 * in profiles/r05_class_probe.txt) -- and at the START of a process the class changes within the first few allocations:
 * plane 0 has a fast partner among planes 1 ... 4 in seven of eight fresh processes (five times it is plane 1), against
 * two of eight for the plane next to it once 12 GB of inputs have been allocated (same file).  trk_reserve_pair,
 * called right after trk_init and before any other device allocation, takes planes of bytes_each one at a time -- eight
 * at most --, times each with the ones before it and stops at the first fast pair; the best pair stays, the others go
 * back (transient: up to six planes, while the device is still empty).  The context owns the pair for its lifetime:
 * trk_dev_alloc_pair lends it out whenever it is fast, both planes are free and bytes_each fits
 * (trk_pair_info.reserved = 1) -- a sub-plane lies in its plane's region, so smaller shapes are served alike -- and
 * trk_dev_free on either pointer hands it back instead of freeing it.  When none of the eight planes makes a fast pair
 * (planes of at least 256 MB: below that no levels can be told apart and the first two are kept as they are) NOTHING stays
 * reserved: every plane goes back, the call returns TRK_OK with info->placed = 0 and trk_dev_alloc_pair searches as if
 * the call had not been made.  Cost: 2 x bytes_each of device memory for the life of the context when it succeeds, up
 * to 8 x bytes_each transiently and ~5 ms per probe at start-up (one probe in most processes). */
int trk_reserve_pair(trk_ctx* ctx, size_t bytes_each, trk_pair_info* info);
/* have[0 .. n_have): planes of bytes_each the caller already holds (an allocator's pooled buffers): have[0] becomes the
 * first plane, the others are the first candidates for the second; the ones not returned stay the caller's. */
int trk_dev_alloc_pair(trk_ctx* ctx, size_t bytes_each, int64_t n_loci, int64_t n_samples, int32_t max_spare,
                       void* const* have, int32_t n_have, void** a, void** b, trk_pair_info* info);

/* Filter opcodes: one per distinct arithmetic in dumpSTR/filters.py.           */
enum {
    TRK_F_LT = 1,          /* CallFilterMinValue :363-367  value < thr (in the plane's dtype) */
    TRK_F_GT = 2,          /* CallFilterMaxValue :405-409                                    */
    TRK_F_RATIO_GT = 3,    /* HipSTRCallFlankIndels/Stutter :444-449: a/b (float64) > thr     */
    TRK_F_CALLED_LT = 4,   /* GangSTRCallExpansionProb{Hom,Het} :597-639, HipSTRCallMinSuppReads
                              on a host pre-parsed plane: value < thr on called samples only  */
    TRK_F_CALLED_SUM_LT = 5,/* GangSTRCallExpansionProbTotal :665-674: (a+b in dtype) < thr   */
    TRK_F_CALLED_EQ = 6,   /* GangSTRCallSpanOnly :686-697: a == b on called samples          */
    TRK_F_CALLED_SUM_EQ = 7,/* GangSTRCallSpanBoundOnly :711-722: a + a2 == b                 */
    TRK_F_CALLED_OUTSIDE_CI = 8, /* GangSTRCallBadCI :739-757: plane a = REPCN [S,P],
                              plane b = REPCI pre-parsed to [S,2P] (lo0,hi0,lo1,hi1,..)       */
    TRK_F_AD_SUPPORT_LT = 9 /* PopSTRCallRequireSupport :858-867: AD[s, gt[s,j]] < thr       */
};
typedef struct {
    int32_t op;
    int32_t plane_a, col_a;
    int32_t plane_b, col_b;
    int32_t col_a2;
    double thr;
} trk_call_filter;

#define TRK_MAX_FILTERS 24
#define TRK_MAX_PLANES 16
#define TRK_MASK_NOCALL 0x80000000u

typedef struct {
    int16_t* gt_out;          /* [L,S,P] genotypes with filtered calls set to -1
                                 (dumpSTR.py:721-727); may be NULL.  May be the batch's own
                                 tensor (gt_out == trk_batch.gt, IN PLACE -- what the reference
                                 does with its record): every kernel reads a cell before it
                                 writes it, and the streaming kernels then store only the
                                 16-byte chunks that hold a filtered call (one big write stream
                                 instead of two: no pair of output planes to place)           */
    uint32_t* filter_mask;    /* [L,S] bit k = filter k fired, bit 31 = sample was a
                                 no-call (dumpSTR.py:651); 0 == 'PASS'; may be NULL */
    int64_t* sample_counters; /* [(1+nf), S] += : row 0 numcalls (:686-687),
                                 row 1+k sample_info[filter k] (:661)              */
    int64_t* sample_totaldp;  /* [S] += DP of PASS calls with DP > 0 (:707-709)    */
    int64_t* sample_dp_missing;/* [S] += PASS calls whose DP is missing (-> nan, :710) */
    int32_t* error;           /* [4] error[0] != 0: a PASS call had negative DP
                                 (ValueError :698-706); error[1]=locus, [2]=sample */
    /* Optional (both or neither): allele_count [sumA] and locus_int [L, TRK_LI_COLS] of the
     * UNFILTERED genotypes (group 0 of a trk_locus_stats(TRK_STATS_COUNT_ONLY) run on the same
     * batch).  The kernel subtracts what every filtered call contributed, so on return they are
     * the counts of gt_out -- the rebuilt record of dumpSTR.py:748-774 -- without a second pass
     * over the genotype tensor.  Follow with trk_locus_finalize.                            */
    int32_t* delta_allele_count;
    int32_t* delta_locus_int;
    double* sample_totaldp_f64; /* [S] += the depth of PASS calls when the DP/LC plane is Float (ExpansionHunter's
                                   LC); sample_totaldp stays 0 then.  Required only for a Float depth plane.   */
    uint8_t* filter_mask8;      /* optional [L,S]: the mask in one byte per call for at most 7 filters -- bit k = filter
                                   k fired, bit 7 = the sample was a no-call (TRK_MASK8_NOCALL); 0 == 'PASS'.  A caller
                                   that rebuilds its records from the mask (as this repository's dumpSTR does: a
                                   filtered call's genotype is '.') asks for this and neither gt_out nor filter_mask:
                                   12 B read + 1 B written per call instead of 12 + 8 (the per-sample counters and the
                                   delta outputs are unchanged).  NULL, or more than 7 filters: not written.        */
} trk_call_out;
#define TRK_MASK8_NOCALL 0x80u

/* (a11)-(a18): evaluate `n_filters` call-level filters on every call of the
 * batch.  `dp_plane` = index of the DP (or LC) plane used for totaldp, -1 if
 * the records carry neither (totaldp := nan, dumpSTR.py:712-713).               */
int trk_call_filters(trk_ctx* ctx, const trk_batch* in, const trk_plane* planes,
                     int n_planes, const trk_call_filter* filters, int n_filters,
                     int dp_plane, trk_call_out* out);

/* ---- dumpSTR locus-level filters ----------------------------------------- */
enum {
    TRK_LOCF_CALLRATE = 0,  /* filters.py:59-61   */
    TRK_LOCF_HWE = 1,       /* filters.py:98-103  */
    TRK_LOCF_HETLOW = 2,    /* filters.py:140-144 */
    TRK_LOCF_HETHIGH = 3,   /* filters.py:181-185 */
    TRK_LOCF_EXTERN0 = 4,   /* bits 4..27: host-evaluated filters (HRUN :190-217,
                               BED regions :219-300) passed in extern_bits        */
    TRK_LOCF_NO_CALLS = 31  /* dumpSTR.py:957-965 */
};
typedef struct {
    double min_callrate, min_hwep, min_het, max_het; /* nan = filter disabled      */
    int32_t use_length;        /* dumpSTR --use-length                             */
    int32_t n_extern;          /* number of extern filter bits in use              */
    const uint32_t* extern_bits; /* device [L] or NULL                            */
} trk_locus_filter_spec;
enum {
    TRK_LC_TOTALCALLS = 0, TRK_LC_PASS = 1, TRK_LC_NO_CALLS = 2,
    TRK_LC_FILTER0 = 3,      /* + bit index of the filter (0..27)                  */
    TRK_LC_HWE_ERRORS = 31,  /* loci where the reference would raise in the HWE filter */
    TRK_LC_COLS = 32
};
typedef struct {
    uint32_t* locus_bits;    /* [L] bit per fired filter; 0 == PASS                */
    int64_t* loc_counters;   /* [TRK_LC_COLS] += (loc_info of dumpSTR.py:1264-1268) */
} trk_locus_out;
/* (a19): locus filter decisions + loc_info counters from group 0 of `stats`.   */
int trk_locus_filters(trk_ctx* ctx, int32_t n_loci, const trk_stats_out* stats,
                      const trk_locus_filter_spec* spec, trk_locus_out* out);

/* ---- associaTR linear-regression scan --------
 * For every locus of a batch, what one iteration of load_trs
 * (associaTR/load_and_filter_genotypes.py:157-259) plus the regression block of
 * perform_gwas_helper (associaTR/associaTR.py:246-291) compute:
 *   curr_samples   = sample_in & GetCalledSamples()                       (:167-169)
 *   allele counts of those samples' genotypes (GetAlleleFreqs(curr_samples), :177)
 *   the locus filter: no called samples / one (rounded) length allele / non-major
 *   allele count < cutoff (:229-239), then 'n covars >= n samples' (associaTR.py:257)
 *   summed length genotype per sample, standardised over curr_samples (:266-273)
 *   OLS of the outcome on [genotype, 1, covariates] over curr_samples: p-value,
 *   coefficient, standard error of the genotype term and the centred R^2 (:277-289,
 *   statsmodels OLS.fit(); here by normal equations in float64: cross-products in one
 *   pass over the genotype tensor, a Cholesky factorisation per locus, Student-t tail).
 * The outcome / covariates are the caller's standardised columns (associaTR.py:198-202),
 * row-major by VECTOR: vec[0] = outcome, vec[1..] = covariates, each [S] (entries of
 * samples with sample_in == 0 are ignored).  The intercept is implicit.
 * trk_assoc_scan takes up to TRK_ASSOC_MAX_VEC_WIDE rows (associaTR.py:138-204 has no bound: any number of
 * --same-file-covars / .npy columns), in ONE pass over the genotype tensor for diploid batches whose rows are whole
 * 16-byte chunks and up to 62 rows (up to four 16-row tiles of [vectors..., 1] per 16 loci on the matrix pipe); other
 * batches with more than TRK_ASSOC_MAX_VEC rows, and every design of 63 rows and more, are scanned pair of 15-row groups
 * by pair -- g(g-1)/2 passes for g = ceil(M / 15) groups -- and the whole design solved per locus by one wavefront
 * (two rows of the normal matrix per lane from 63 rows on; the bound is that tile's size in LDS).
 * trk_assoc_scan_dosage takes up to TRK_ASSOC_MAX_VEC rows in one pass and up to TRK_ASSOC_MAX_VEC_WIDE pair of groups by pair.
 */
#define TRK_ASSOC_MAX_VEC 31
#define TRK_ASSOC_MAX_VEC_WIDE 126
typedef struct {
    int32_t n_vec;              /* M >= 1: outcome + (M-1) covariates                        */
    int32_t flags;              /* 0                                                         */
    const double* vec;          /* device [M, S]                                             */
    const uint8_t* sample_in;   /* device [S], NULL = every sample                           */
    const double* allele_len;   /* device [sumA] length in repeat units per allele INDEX
                                   (the LUT of GetLengthGenotypes, tr_harmonizer.py:1239)    */
    const uint16_t* rlen_class; /* device [sumA]: for length class c of locus l (entry
                                   allele_off[l]+c, classes as trk_batch.len_class) the rank of
                                   its ROUNDED length among the locus's distinct rounded lengths
                                   (clean_len_alleles, load_and_filter_genotypes.py:37-45)   */
    double non_major_cutoff;    /* --non-major-cutoff                                        */
} trk_assoc_params;

/* trk_assoc_out.locus_int columns ([L, TRK_AI_COLS] int32) */
enum {
    TRK_AI_N_TESTED = 0,   /* samples in curr_samples                                        */
    TRK_AI_STATUS = 1,     /* TRK_AS_*                                                       */
    TRK_AI_N_RALLELES = 2, /* distinct rounded length alleles among curr_samples             */
    TRK_AI_RANK = 3,       /* rank of the design (df_resid = n - rank)                       */
    TRK_AI_N_BAD = 4,      /* calls with an allele index >= A_l                               */
    TRK_AI_N_HAPS = 5,     /* called haplotypes among curr_samples (denominator of the AFs)  */
    TRK_AI_COLS = 8
};
enum {
    TRK_AS_OK = 0,
    TRK_AS_NO_CALLED = 1,      /* 'No called samples'                                        */
    TRK_AS_ONE_ALLELE = 2,     /* 'Only one called allele'                                   */
    TRK_AS_NON_MAJOR = 3,      /* 'non-major allele count<cutoff'                            */
    TRK_AS_N_COVARS = 4,       /* 'n covars >= n samples'                                    */
    TRK_AS_ZERO_VARIANCE = 5,  /* summed genotype constant over curr_samples (reference: 0/0) */
    TRK_AS_COLLINEAR = 6       /* genotype in the span of the covariates                     */
};
/* trk_assoc_out.locus_f64 columns ([L, TRK_AF_COLS] float64); nan where not tested */
enum {
    TRK_AF_PVAL = 0,
    TRK_AF_COEF = 1,      /* coefficient of the STANDARDISED genotype (params[0])           */
    TRK_AF_SE = 2,        /* its standard error (bse[0])                                    */
    TRK_AF_RSQUARED = 3,
    TRK_AF_GT_STD = 4,    /* np.std of the summed genotypes (associaTR.py:272)              */
    TRK_AF_GT_MEAN = 5,
    TRK_AF_TVALUE = 6,
    TRK_AF_DF_RESID = 7,
    TRK_AF_NONMAJOR = 8,  /* np.sum(af)*n_samples*2 of load_and_filter_genotypes.py:236     */
    TRK_AF_COLS = 10
};
typedef struct {
    int32_t* locus_int;     /* device [L, TRK_AI_COLS]                                       */
    double* locus_f64;      /* device [L, TRK_AF_COLS]                                       */
    int32_t* allele_count;  /* device [sumA] counts per allele INDEX over curr_samples       */
} trk_assoc_out;
int trk_assoc_scan(trk_ctx* ctx, const trk_batch* in, const trk_assoc_params* prm, trk_assoc_out* out);

/* associaTR --beagle-dosages (load_and_filter_genotypes.py:179-215, associaTR.py:266-273): the
 * regressor is the expected summed length from the Beagle AP1/AP2 allele probabilities,
 *     g_s = sum over rounded-length classes u (ascending) of len_u * (d_u,1 + d_u,2),
 *     d_u,p = sum over the alleles of class u (index order, reference allele first) of AP_p,
 *     AP_p(ref) = max(0, 1 - sum_i AP_p[i])   (float32, numpy's summation order),
 * over curr_samples = sample_in & called (by GT).  Same regression outputs as trk_assoc_scan
 * (locus filters that depend on the allele frequencies are left to the caller: status is one of
 * OK / N_COVARS / ZERO_VARIANCE / COLLINEAR); in addition per class the sums behind
 * allele_frequency and the per-allele dosage r^2, and per locus those behind the length r^2.   */
typedef struct {
    const float* ap1;            /* device [L, S, n_alt_cols] float32 (cyvcf2 format('AP1'))           */
    const float* ap2;
    int32_t n_alt_cols;          /* >= max_l (A_l - 1); columns beyond A_l - 1 are not read             */
    int32_t reserved;
    const int32_t* perm;         /* device [sumA] allele indices of each locus ordered by (class, index) */
    const uint16_t* dclass;      /* device [sumA] by allele INDEX: rank of round(length, precision) among
                                    the locus's distinct rounded lengths (np.unique(len_alleles))        */
    const double* dclass_value;  /* device [sumA] rounded length of class u at allele_off[l] + u         */
    const uint16_t* best_class;  /* device [sumA] by allele INDEX: the class whose value equals
                                    np.around(length, precision) (best-guess calls, :199-202), 0xffff none */
} trk_assoc_dosage;
/* class_sums [sumA, TRK_ADC_COLS] (row allele_off[l] + u): sum d, sum d^2, sum x, sum x*d over the
 * 2n haplotype entries (x = best-guess call equals the class); locus_sums [L, TRK_ADL_COLS]: sum x,
 * sum x^2, sum y, sum y^2, sum x*y, 2n, min x, max x with x = best-guess length, y = expected length per
 * haplotype (min == max: numpy's corrcoef of a constant vector is decided by rounding -- the caller handles it) */
#define TRK_ADC_COLS 4
#define TRK_ADL_COLS 8
int trk_assoc_scan_dosage(trk_ctx* ctx, const trk_batch* in, const trk_assoc_params* prm,
                          const trk_assoc_dosage* dos, trk_assoc_out* out, double* class_sums, double* locus_sums);

/* ---- per-sample dosages ----------------------------------------
 * TRRecord.GetDosages (tr_harmonizer.py:1098-1208) for every (locus, sample) of a batch:
 *   BESTGUESS       sum of the called alleles' lengths ('-1' / '-2' count 0)
 *   BESTGUESS_NORM  the same with any '-1' / '-2' -> nan, then (d - 2 min) / (max - min) clipped to [0, 2]
 *   BEAGLEAP        sum over both haplotypes of clip(AP_p . alt_lengths, 0, max alt) + clip(1 - sum AP_p, 0, 1) * ref
 *   BEAGLEAP_NORM   normalised the same way
 * float32 out [L, S] (the reference returns float32).  locus_err [L]: bit 0 an AP row sums to more
 * than 1.1, bit 1 a negative AP value, bit 2 a normalised dosage >= 2.1 or <= -0.1 -- the conditions
 * under which the reference raises ValueError (strict) or returns nan for the whole record.       */
enum { TRK_DOS_BESTGUESS = 0, TRK_DOS_BEAGLEAP = 1, TRK_DOS_BESTGUESS_NORM = 2, TRK_DOS_BEAGLEAP_NORM = 3 };
int trk_dosages(trk_ctx* ctx, const trk_batch* in, const double* allele_len, int dosage_type, const float* ap1,
                const float* ap2, int n_alt_cols, float* out, int32_t* locus_err);

/* ---- the sample columns of a batch of VCF records, parsed on the device (round 4; SURVEY.md section 8(f1) widened) ----
 * Replaces the per-sample half of the native reader's parse (trk_vcf_read_batch: cyvcf2's genotype.array() and
 * format(key) of the reference's record loop, tr_harmonizer.py:1420-1499 / dumpSTR.py:613-700) for callers that bring
 * the batch's TEXT to the device instead of its decoded arrays: the genotype tensor and up to four scalar FORMAT planes
 * come into being in HBM.  The host still finds the lines, the nine fixed columns and the FORMAT keys (trk_vcf.h).
 * Grammar per token: alleles '.' or at most four digits separated by '/' or '|'; Integer planes -?d{1,9} or '.';
 * Float planes -?d*(.d*)? of at most fifteen digits or '.', value = the correctly rounded float64 of the decimal cast to
 * float32 (what strtod + a cast give).  Anything else -- exponents, vectors, inf / nan, signs on alleles -- sets
 * TRK_PARSE_HOST in the record's flag: the rows of that record are undefined and the caller parses it with the host
 * code.  Subfields a token does not hold are missing values (INT32_MIN / NaN), alleles beyond a call's own are -2.     */
#define TRK_PARSE_MAX_PLANES 4
enum { TRK_PARSE_INT = 0, TRK_PARSE_FLOAT = 1 };
enum {
    TRK_PARSE_HOST = 1,    /* a token outside the device grammar: parse this record on the host              */
    TRK_PARSE_COLUMNS = 2, /* not n_samples sample columns (the reader's error 3)                            */
    TRK_PARSE_PLOIDY = 4   /* a call with more alleles than `ploidy` (the reader's error 2)                  */
};
typedef struct {
    const uint8_t* text;     /* device, 16-byte aligned; readable up to the 16-byte chunk that holds the last line_end */
    int64_t n_bytes;
    int32_t n_records, n_samples, ploidy;
    int32_t n_planes;        /* <= TRK_PARSE_MAX_PLANES */
    const int64_t* smp_off;  /* device [n_records]: offset in text of the record's first sample token (field_off[9])      */
    const int64_t* line_end; /* device [n_records]: offset of the record's newline ('\n' or '\r': it must be there)       */
    const int8_t* gt_idx;    /* device [n_records]: index of GT among the record's FORMAT keys, -1: none                 */
    const int8_t* plane_idx[TRK_PARSE_MAX_PLANES];   /* device [n_records] each: index of plane k's key, -1: absent      */
    int32_t plane_kind[TRK_PARSE_MAX_PLANES];        /* TRK_PARSE_INT / TRK_PARSE_FLOAT                                   */
} trk_parse_in;
typedef struct {
    int16_t* gt;           /* device [n_records, n_samples, ploidy]                       */
    uint8_t* phased;       /* optional device [n_records, n_samples]: some '|' in the call */
    void* planes[TRK_PARSE_MAX_PLANES];   /* device [n_records, n_samples] int32 / float32 */
    uint8_t* locus_ploidy; /* device [n_records]: the most alleles a call of the record holds (>= 1) */
    uint8_t* flags;        /* device [n_records]: 0 or TRK_PARSE_* bits                    */
} trk_parse_out;
int trk_parse_samples(trk_ctx* ctx, const trk_parse_in* in, trk_parse_out* out);

/* ---- BGZF members inflated on the device (round 5; SURVEY.md section 8(f1): the reader row's "BGZF block inflate") ----
 * Replaces the inflate step of the native reader (trk_vcf.cpp's pool of libdeflate / zlib workers -- what htslib does
 * for the reference, /root/reference/trtools/utils/utils.py:19-67) for callers that bring a batch's COMPRESSED bytes to
 * the device: a bgzip'ed file is a chain of independent gzip members of at most 64 KiB of text (RFC 1952 / 1951, no
 * preset dictionary); the host finds the members (sizes are in their headers) and names each one's raw DEFLATE payload;
 * one wave inflates one member (stored, fixed and dynamic blocks) straight into the text buffer the parse kernels
 * read.  The text equals zlib's byte for byte; a member the kernel cannot finish gets a flag and is the host's
 * (nothing is written beyond out_off + out_len of any member).  comp must be readable for 8 bytes beyond the last
 * payload (the member's own CRC32 / ISIZE trailer is there in a BGZF file). */
enum {
    TRK_INFLATE_STREAM = 1,   /* not a valid DEFLATE stream (or it reads beyond the payload)                */
    TRK_INFLATE_OVERRUN = 2,  /* the stream holds more, or less, text than out_len                           */
    TRK_INFLATE_INPUT = 4     /* offsets / lengths outside the buffers                                       */
};
typedef struct {
    const uint8_t* comp;      /* device: the compressed bytes                                                */
    int64_t n_comp_bytes;
    int32_t n_blocks;
    int32_t pad_;
    const int64_t* in_off;    /* device [n_blocks]: offset in comp of the member's DEFLATE payload           */
    const int32_t* in_len;    /* device [n_blocks]: payload bytes (BSIZE + 1 - header - 8)                   */
    const int64_t* out_off;   /* device [n_blocks]: offset in text of the member's first byte                */
    const int32_t* out_len;   /* device [n_blocks]: the member's ISIZE (<= 65536)                            */
} trk_inflate_in;
typedef struct {
    uint8_t* text;            /* device                                                                      */
    uint8_t* flags;           /* device [n_blocks]: 0 or TRK_INFLATE_* bits                                  */
} trk_inflate_out;
int trk_inflate_blocks(trk_ctx* ctx, const trk_inflate_in* in, const trk_inflate_out* out);

/* The native reader's inflate hook (include/trk_vcf.h: trk_vcf_set_inflate_hook) served by this context's device: every
 * run of BGZF members the reader reads goes to the device compressed, is inflated there into a segment of text that
 * STAYS in HBM, is indexed there (newlines; the ninth tab of every line) and only the newlines and the heads of the lines
 * go back to the host.  trk_inflate_hook hands out the three values of a trk_vcf_inflate_hook {user, seed, inflate};
 * trk_inflate_text copies bytes [abs_from, abs_from + n_bytes) of the file's text (trk_vcf_text_abs + an offset into
 * trk_vcf_batch.text) from the segments to dst (device) -- the text trk_parse_samples / trk_format_samples read -- and
 * lets go of the segments that end at or before release_before.  A member the kernel flags is inflated by zlib inside
 * the hook.  trk_inflate_stats: {members, of which flagged, bytes of text, compressed bytes, hook calls}.  The hook runs
 * on the queue of the thread that reads (trk_thread_queue); one reader per context at a time. */
int trk_inflate_hook(trk_ctx* ctx, void** user, void** seed_fn, void** inflate_fn);
/* the hook's `inflate` in two halves (trk_vcf_inflate_hook.submit / .collect): the reader keeps two runs in flight, the
 * kernels run on a queue of the hook's own, the file read and the upload of run k + 1 go on behind the kernel of run k */
int trk_inflate_hook_async(trk_ctx* ctx, void** submit_fn, void** collect_fn);
int trk_inflate_text(trk_ctx* ctx, uint64_t abs_from, int64_t n_bytes, void* dst, uint64_t release_before);
int trk_inflate_stats(trk_ctx* ctx, uint64_t out[5]);

/* ---- BGZF members DEFLATED on the device (round 6; the mirror of trk_inflate_blocks) -------------------------------------
 * What it replaces: `bgzip -f` over dumpSTR's finished output (the reference shells out, dumpSTR.py:1241-1245, 1347-1352).
 * n bytes of text in HOST memory (pinned or not) -> consecutive BGZF members of TRK_DEFLATE_MEMBER bytes of text each in host_out: the
 * text goes up, one wave per member makes the member's DEFLATE stream (greedy LZ77, sixteen candidates compared side by side per step, one
 * dynamic-Huffman block; a member that would not get smaller is stored), the members are laid out back to back on the device
 * and come down in one copy; the CRC-32 of every member is computed on the host meanwhile (the text is there) and put in.
 * Any stream that inflates to the text is a right answer: the bytes differ from trk_bgzf_compress's (libdeflate / zlib), the
 * text they hold does not.  No end-of-file member (trk_bgzf_eof).  out_cap >= trk_deflate_bound(n).  Returns TRK_OK and
 * *out_bytes; TRK_ERR_ARG: out_cap too small.  Calls on one context take turns (a lock); the call has a queue of its own, so a
 * writer thread may make it beside the caller's kernels. */
#define TRK_DEFLATE_MEMBER 16384      /* bytes of text per member trk_deflate_bgzf makes (bgzip's: 0xff00; any size <= 64 KB is BGZF) */
size_t trk_deflate_bound(size_t n);
int trk_deflate_bgzf(trk_ctx* ctx, const void* host_text, size_t n, void* host_out, size_t out_cap, size_t* out_bytes);

/* ---- dumpSTR's sample columns written on the device (round 4; the host form is trk_vcf_dumpstr_records' span writer) ----
 * Per sample: a tab, then the token as it stands (+ ':.' per FORMAT key it lacks) + ':PASS' / ':NOCALL', or for a filtered
 * call the nulled token + ':' + '<filter>_<value>,...' (dumpSTR.py:648-683, 715-746).  A kept token is copied only when
 * decode -> format would print it back unchanged (canonical numbers, one separator in GT, ASCII strings, no vectors); a
 * record with anything else -- or a value whose '%g' needs an exponent or sits on a rounding tie -- gets TRK_PARSE_HOST in
 * its flag and is left to the host writer.  Pass 1 leaves rec_len / flags; the caller lays the records out (out_off) and
 * pass 2 writes the bytes of every unflagged record.  text / smp_off / line_end as for trk_parse_samples.            */
#define TRK_FORMAT_MAX_FIELDS 16
#define TRK_FORMAT_MAX_FILTERS 7
enum { TRK_FORMAT_GT = 1, TRK_FORMAT_INT = 2, TRK_FORMAT_FLOAT = 3, TRK_FORMAT_STRING = 4 };
typedef struct {
    const uint8_t* text;
    int64_t n_bytes;
    int32_t n_records, n_samples, mask_stride, plane_stride;
    const int64_t* smp_off;
    const int64_t* line_end;
    const uint8_t* field_kind;   /* device [n_records][TRK_FORMAT_MAX_FIELDS]: TRK_FORMAT_* of the record's FORMAT keys   */
    const uint8_t* n_fields;     /* device [n_records]: FORMAT keys of the record; 0: not for the device                 */
    const uint8_t* ploidy;       /* device [n_records]: alleles of a nulled call ('./.')                                 */
    const uint8_t* mask8;        /* device [n_records, mask_stride]: trk_call_out.filter_mask8                           */
    int32_t n_filters, reserved; /* <= TRK_FORMAT_MAX_FILTERS                                                            */
    char filter_name[TRK_FORMAT_MAX_FILTERS][32];
    const void* filter_plane[TRK_FORMAT_MAX_FILTERS];   /* device [n_records, plane_stride]: the value behind '<name>_<value>' */
    int32_t filter_dtype[TRK_FORMAT_MAX_FILTERS];       /* 0 int32, 1 float32                                            */
} trk_format_in;
typedef struct {
    uint32_t* rec_len;       /* device [n_records] (pass 1): bytes of the record's sample columns, 0 when flagged        */
    uint8_t* flags;          /* device [n_records] (pass 1 writes, pass 2 reads)                                         */
    uint8_t* out;            /* device (pass 2)                                                                          */
    const int64_t* out_off;  /* device [n_records] (pass 2): where every record's columns go in `out`                    */
} trk_format_out;
int trk_format_samples(trk_ctx* ctx, const trk_format_in* in, trk_format_out* out, int pass);

/* ---- qcSTR's reductions (trtools/qcSTR/qcSTR.py:529-561, 619-621) ----------------
 * One pass over the genotype tensor and (optionally) the FORMAT quality plane of a batch:
 *   a sample's entry at a locus is a CALL unless every haplotype index of the record is -1 (qcSTR.py:533-535 --
 *   note: not dumpSTR's rule, a half-missing call counts here);
 *   sample_calls[s] = calls of sample s over the batch's loci (qcSTR.py:536), locus_calls[l] = calls at locus l
 *   (summed per chromosome by the caller, qcSTR.py:537);
 *   quality: the score of a no-call becomes nan (qcSTR.py:540); then
 *     ignore_no_call == 0: nan -> 0 and every selected sample enters (543): sample_qual_sum[s] += q (548),
 *                          locus mean = locus_qual_sum[l] / locus_qual_n[l] with n = selected samples (554);
 *     ignore_no_call != 0: only the non-nan entries enter sums and counts (551, 556).
 *   sample_in [S] (qcSTR --samples, qcSTR.py:463-470): entries of samples outside the set count nowhere.
 * Sums are float64 in a fixed order (the reference adds float32 scores into a float64 total per sample and takes a
 * float32 pairwise mean per locus: agreement is to float32 rounding, not bit for bit).  The counts are exact.   */
typedef struct {
    const uint8_t* sample_in; /* [S] or NULL                                                    */
    const float* quality;     /* [L*S] float32 (nan = missing) or NULL: call counts only        */
    int32_t ignore_no_call;   /* qcSTR --quality-ignore-no-call                                 */
    int32_t pad;
} trk_qc_params;
typedef struct {
    int64_t* sample_calls;    /* [S]                                                            */
    int64_t* locus_calls;     /* [L]                                                            */
    double* sample_qual_sum;  /* [S] or NULL (required with a quality plane)                    */
    int64_t* sample_qual_n;   /* [S] or NULL: entries summed per sample                         */
    double* locus_qual_sum;   /* [L] or NULL (required with a quality plane)                    */
    int64_t* locus_qual_n;    /* [L] or NULL: entries summed per locus                          */
} trk_qc_out;
int trk_qc_reduce(trk_ctx* ctx, const trk_batch* in, const trk_qc_params* prm, trk_qc_out* out);

/* Two-sided Student-t tail 2*sf(|t|, df) == scipy.stats.t.sf(|t|, df)*2 (the third-party call
 * behind statsmodels' pvalues); host double, same code as the device finaliser.              */
double trk_student_t_two_sided(double t, double df);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) --------------------- */
/* 128-byte opaque id created by rank 0 and distributed by the launcher.       */
int trk_comm_unique_id(uint8_t id[128]);
int trk_comm_init(trk_ctx* ctx, int rank, int n_ranks, const uint8_t id[128]);
/* in-place sum over ranks of int64 device counters (sample/locus counters).   */
int trk_allreduce_sum_i64(trk_ctx* ctx, int64_t* dev, size_t count);
/* gather `bytes_per_rank` bytes from every rank into recv (rank-major).       */
int trk_allgather(trk_ctx* ctx, const void* send_dev, void* recv_dev, size_t bytes_per_rank);
/* The whole exchange of one dumpSTR pass as ONE grouped RCCL launch (ncclGroupStart/End): the in-place sum of
 * `n_sums` packed int64 counters (sample_info rows, totaldp, loc_info: dumpSTR.py:1251-1268 -- the caller lays them
 * out back to back in one buffer) and the rank-major gather of every rank's per-locus filter bits.  Either half may
 * be empty (NULL / 0).                                                                                            */
int trk_exchange(trk_ctx* ctx, int64_t* sums_dev, size_t n_sums, const void* send_dev, void* recv_dev,
                 size_t bytes_per_rank);

/* ---- scalar helpers (host, double) --------------------------------------- */
/* Two-sided exact binomial test p-value == scipy.stats.binomtest(k, n, p).pvalue
 * (third-party call at utils.py:334-338); same code as the device finaliser.  */
double trk_binomtest_two_sided(int64_t k, int64_t n, double p);


/* Measurement and test equipment of the same library (timers, the bare stream probe, the synthetic call-set
 * generators, device-side test entries) is declared in trk_test.h: not part of the drop-in surface. */

#ifdef __cplusplus
}
#endif
#endif /* TRK_H */
