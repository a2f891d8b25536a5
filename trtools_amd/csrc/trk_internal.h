// trk_internal.h -- declarations shared by the translation units of libtrk.so
#ifndef TRK_INTERNAL_H
#define TRK_INTERNAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/trk.h"
#include "../../include/trk_test.h"

#define TRK_MAX_PLOIDY 8

// an option of include/trk_test.h (nullptr: unset); defined in trk_vcf.cpp, which the sanitizer builds recompile
extern "C" __attribute__((visibility("hidden"))) const char* trk_opt(const char* name);
// CRC-32 of every `member`-byte piece of text[0 .. n) on the caller-side worker pool (trk_vcf.cpp: libdeflate's where the
// image has it, zlib's else): crc[m] for m < ceil(n / member)
extern "C" __attribute__((visibility("hidden"))) void trk_member_crc32(const void* text, size_t n, size_t member, uint32_t* crc);

namespace trk {
// class_ws: device scratch of n_class_runs x (sumA + L x TRK_LI_COLS) int32 for the class passes of a batch whose
// columns are ordered by sample class (trk_batch.class_runs); nullptr: the per-call group kernels
hipError_t launch_locus_count(const trk_batch& b, int max_alleles, int32_t* allele_count, int32_t* locus_int,
                              int n_cu, hipStream_t stream, bool twin, int32_t* class_ws);
// wgs_per_cu: resident workgroups (of four members) per CU: 4 = all the kernel's LDS allows, the best rate of the kernel
// alone; 3 leaves 59 KB of a CU's LDS to whatever else runs (the hook: the parse and count kernels of the batch before)
hipError_t launch_inflate(const trk_inflate_in& in, const trk_inflate_out& out, int n_cu, hipStream_t stream, int wgs_per_cu = 4);
struct LineIndexWs {     // device results / scratch of launch_line_index (trk_inflate.hip)
    uint32_t* counts;    // [tiles of 16 KB + 1]
    uint32_t* n_nl;      // newlines found
    uint64_t* nl;        // [nl_cap] their offsets, bit 63: after a '\r'
    uint32_t nl_cap;
    uint64_t* head_off;  // [nl_cap + 1] per line (the last: the unfinished one)
    uint32_t* head_len;
    uint32_t* pack_off;
    uint32_t* head_total;
    int32_t* state;      // [2] tabs seen in the unfinished last line; the text's last byte
    uint8_t* packed;     // the heads back to back
    uint32_t packed_cap;
};
size_t deflate_slot_bytes();
size_t deflate_tok_bytes(int n_cu, int n_members);
hipError_t launch_deflate(const uint8_t* text, int64_t n, uint8_t* slots, uint32_t* sizes, uint32_t* tok, uint64_t* off, uint8_t* out,
                          int n_cu, hipStream_t stream);
hipError_t launch_line_count(const uint8_t* text, int64_t n, const LineIndexWs& ws, hipStream_t stream);   // step 1: *ws.n_nl
hipError_t launch_line_index(const uint8_t* text, int64_t n, int tabs_in, const LineIndexWs& ws, hipStream_t stream);
hipError_t launch_permute_columns(const int16_t* src, int16_t* dst, const int32_t* col, int64_t n_loci, int n_src,
                                  int n_dst, int ploidy, int n_cu, hipStream_t stream);
bool launch_locus_stats_fused(const trk_batch& b, int32_t* allele_count, int32_t* locus_int, double* locus_f64,
                              void* worklist, double nalleles_thresh, hipStream_t stream, hipError_t* err, int stage);
hipError_t launch_locus_finalize(const trk_batch& b, const int32_t* allele_count, int32_t* locus_int,
                                 double* locus_f64, int32_t* scratch, void* worklist, double nalleles_thresh,
                                 hipStream_t stream);
size_t finalize_worklist_bytes(int64_t n_group_loci);
// one deferred HWE test: homogeneous work items, so the lanes of a wave differ only in loop trip counts, and loci
// whose two allele partitions coincide are tested once.  Written by the finalisers (trk_kernels.hip), read by the
// test kernels (trk_hwe.hip).
struct HweItem {
    int32_t slot;   // g * L + l
    int32_t modes;  // bit 0: write HWEP_LEN, bit 1: write HWEP_STR
    int32_t k, n;
    double p;
};
// the tests of a compact work list (header word 0 items; `overflow`: room for one index per item) and of fixed slots
// (modes == 0: none) -- trk_hwe.hip
hipError_t launch_hwe_tests(unsigned int* count, const HweItem* items, double* locus_f64, unsigned int* overflow,
                            int64_t n_group_loci, hipStream_t stream);
hipError_t launch_hwe_slots(const HweItem* items, unsigned int n_slots, double* locus_f64, hipStream_t stream);
// device scratch owned by the context, handed out (and grown) on request; nullptr when it cannot be had
struct Scratch {
    void* user;
    void* (*get)(void* user, size_t bytes);
    void (*next_kernel)(void* user);   // called between the streaming kernel and its reduction kernel (profiling bracket)
};
hipError_t launch_call_filter(const trk_batch& b, const trk_plane* planes, int n_planes,
                              const trk_call_filter* filters, int n_filters, int dp_plane, const trk_call_out& out,
                              int n_cu, hipStream_t stream, const Scratch& scratch);
hipError_t launch_locus_filter(int L, const int32_t* locus_int, const double* locus_f64,
                               const trk_locus_filter_spec& spec, uint32_t* bits, int64_t* counters,
                               hipStream_t stream);
hipError_t launch_synth(const trk_synth_spec& sp, int16_t* gt, int32_t* dp, float* q, int32_t* dstutter,
                        int32_t* dflank, int n_cu, hipStream_t stream);
hipError_t launch_synth_gangstr(const trk_synth_spec& sp, const int16_t* gt, const int32_t* dp,
                                const int32_t* allele_repcn, float* qexp, int32_t* repcn, int32_t* rc,
                                int32_t* repci, int n_cu, hipStream_t stream);
// associaTR scan (trk_assoc.hip): prepare -> scan -> finalize on the same stream and workspace
hipError_t launch_binomtest_batch(const int64_t* k, const int64_t* n, const double* p, int64_t count, double* out,
                                  int lanes, hipStream_t stream);
size_t assoc_workspace_bytes(const trk_batch& b, int n_vec);
hipError_t launch_assoc_prepare(const trk_batch& b, const trk_assoc_params& prm, const trk_assoc_out& out,
                                void* workspace, hipStream_t stream);
hipError_t launch_assoc_scan(const trk_batch& b, const trk_assoc_params& prm, const trk_assoc_out& out,
                             void* workspace, int n_cu, hipStream_t stream);
hipError_t launch_assoc_finalize(const trk_batch& b, const trk_assoc_params& prm, const trk_assoc_out& out,
                                 void* workspace, hipStream_t stream);
hipError_t launch_assoc_dosage(const trk_batch& b, const trk_assoc_params& prm, const trk_assoc_dosage& dos,
                               const trk_assoc_out& out, double* class_sums, double* locus_sums, void* workspace,
                               hipStream_t stream);
hipError_t launch_dosages(const trk_batch& b, const double* allele_len, int type, const float* ap1, const float* ap2,
                          int n_alt_cols, float* out, int32_t* locus_err, hipStream_t stream);
hipError_t launch_pad_rows(const void* src, void* dst, int64_t n_rows, int row_words, int pad_words, uint32_t fill,
                           int n_cu, hipStream_t stream);
hipError_t launch_planarize(const void* src, void* dst, int64_t n_cells, int ncol, hipStream_t stream);
hipError_t launch_stream_probe(const void* const* in, int n_in, void* const* out, int n_out, int64_t n_loci,
                               int64_t n_samples, int n_cu, hipStream_t stream);
// qcSTR's reductions (trk_qc.hip)
size_t qc_workspace_bytes(const trk_batch& b, const float* quality, int n_cu);
hipError_t launch_qc_reduce(const trk_batch& b, const trk_qc_params& prm, const trk_qc_out& out, void* workspace,
                            int n_cu, hipStream_t stream);

hipError_t launch_parse_samples(const trk_parse_in& in, const trk_parse_out& out, hipStream_t stream);
hipError_t launch_format_samples(const trk_format_in& in, const trk_format_out& out, int pass, hipStream_t stream);
}  // namespace trk
#endif
