/* trk_test.h -- measurement and test equipment exported by libtrk.so next to the product ABI (include/trk.h):
 * what bench.py, tools/ and tests/ use and a TRTools integration never calls.  No counterpart in the reference.
 */
#ifndef TRK_TEST_H
#define TRK_TEST_H
#include "trk.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- timing on the context's stream (HIP events) ------------------------ */
/* Slots 0..TRK_N_TIMERS-1.  start/stop enqueue events on the compute stream;
 * elapsed synchronises on the stop event and returns milliseconds.           */
#define TRK_N_TIMERS 16
int trk_timer_start(trk_ctx* ctx, int slot);
int trk_timer_stop(trk_ctx* ctx, int slot);
int trk_timer_elapsed_ms(trk_ctx* ctx, int slot, float* ms);

/* Measurement aids (bench.py; no counterpart in the reference).
 * trk_stream_probe: the call-filter pass's stream shape with no arithmetic -- three [n_loci, n_samples] 4-byte planes
 * read, two written, 16 bytes per lane, the pass's tiling and grid -- `reps` launches timed with HIP events on the
 * selected queue; *avg_ms = average launch time.  out0 / out1 are overwritten.  What THIS box's memory system gives
 * a 12 B-in / 8 B-out stream (boxes of one pool differ by 15 %: profiles/r03_notes.md).
 * trk_device_clocks: the device's reported peak engine / memory clocks (kHz) and memory bus width (bits). */
int trk_stream_probe(trk_ctx* ctx, const void* in0, const void* in1, const void* in2, void* out0, void* out1,
                     int64_t n_loci, int64_t n_samples, int32_t reps, float* avg_ms);
int trk_device_clocks(trk_ctx* ctx, int32_t* sclk_khz, int32_t* mclk_khz, int32_t* mem_bus_bits);

/* ---- host / device test entries of the exact binomial test ------------------- */
double trk_binom_pmf(int64_t k, int64_t n, double p);
/* The same test for `count` triples ON THE DEVICE: lanes = 2, the lane-pair routine statSTR's / dumpSTR's deferred
 * HWE tests use (k_hwe_test); lanes = 1, the serial routine in one lane (the two agree bit for bit).  k, n, p, out
 * are HOST arrays (copied in and out; a test / diagnostic entry, synchronous).  Triples outside n >= 1,
 * 0 <= k <= n, 0 <= p <= 1 give nan.                                                                          */
int trk_binomtest_batch(trk_ctx* ctx, const int64_t* k, const int64_t* n, const double* p, int64_t count, double* out,
                        int32_t lanes);

/* ---- synthetic many-sample VCF batches (bench / tests) ------------------- */
typedef struct {
    uint64_t seed;
    int32_t n_loci, n_samples;           /* diploid                               */
    const int32_t* allele_off;            /* device [L+1]                          */
    const uint32_t* allele_cdf24;         /* device [sumA] cumulative allele probabilities, 24-bit */
    const uint32_t* miss_thr16;           /* device [L] P(no-call) * 65536         */
    const uint32_t* inbreed_thr16;        /* device [L] P(2nd allele := 1st) * 65536 */
    int32_t locus_base;                   /* global index of locus 0 (sharding)    */
    int32_t reserved;
} trk_synth_spec;
/* Fill gt [L,S,2] and (optional, may be NULL) FORMAT planes DP int32 [L,S],
 * Q float32 [L,S], DSTUTTER / DFLANKINDEL int32 [L,S] with a counter-based
 * generator that trtools_amd/synth.py reproduces bit-for-bit in numpy.         */
int trk_synth_fill(trk_ctx* ctx, const trk_synth_spec* spec, int16_t* gt, int32_t* dp,
                   float* q, int32_t* dstutter, int32_t* dflankindel);

/* GangSTR-shaped FORMAT planes for an already generated (gt, dp) pair: QEXP float32 [L,S,3],
 * REPCN int32 [L,S,2], RC int32 [L,S,4] (enclosing, spanning, FRR, bounding; sums to DP) and
 * REPCI int32 [L,S,4] (lo0,hi0,lo1,hi1).  allele_repcn: device [sumA] integer repeat count
 * of every allele.  numpy twin: trtools_amd/synth.py::gangstr_planes_numpy.               */
int trk_synth_fill_gangstr(trk_ctx* ctx, const trk_synth_spec* spec, const int16_t* gt, const int32_t* dp,
                           const int32_t* allele_repcn, float* qexp, int32_t* repcn, int32_t* rc,
                           int32_t* repci);

/* ---- options: forced code paths of the parity tests, A/B switches of tools/ -------------------
 * The product reads a handful of documented settings from the environment (README.md "Environment") and nothing
 * else.  Every other switch of the library -- "take the per-call kernel although the streaming one applies", "launch
 * geometry x", "print the reader's timing" -- is an OPTION by name (the names are the ones the tools have always used:
 * TRK_CF_GENERIC, TRK_FUSED_STATS, TRK_VCF_PARSE_GENERIC ...): set here, process-wide, value NULL = unset; read by the
 * library at every launch.  Only the lab build of the library (`make lab`: -DTRK_LAB, trtools_amd/libtrk_lab.so, what
 * tools/ load through TRK_LIBTRK) also takes options from the environment. */
int trk_test_set_option(const char* name, const char* value);
const char* trk_test_get_option(const char* name);

#ifdef __cplusplus
}
#endif
#endif /* TRK_TEST_H */
