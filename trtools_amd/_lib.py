"""ctypes binding of libtrk.so (include/trk.h).  No torch, no fallback: if the
HIP library is missing or no MI355X is visible this module raises."""
from . import _knobs
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libtrk.so')

TRK_LI_COLS = 12
TRK_LF_COLS = 12
TRK_LC_COLS = 32
TRK_MAX_FILTERS = 24
TRK_MAX_PLANES = 16
TRK_MASK_NOCALL = 0x80000000
TRK_MASK8_NOCALL = 0x80

# locus_int columns
LI_N_CALLED, LI_N_LOWPLOIDY, LI_N_HOM_LEN, LI_N_HOM_STR, LI_N_ALLELES, LI_N_BAD, \
    LI_HWE_STATUS_LEN, LI_HWE_STATUS_STR, LI_N_SAMPLES, LI_NALLELES_LEN, LI_NALLELES_STR = range(11)
HWE_OK, HWE_NAN, HWE_VALUE_ERROR, HWE_INDEX_ERROR = range(4)
# locus_f64 columns
LF_THRESH, LF_MEAN, LF_MODE, LF_VAR, LF_HET_LEN, LF_HET_STR, LF_ENTROPY_LEN, LF_ENTROPY_STR, \
    LF_HWEP_LEN, LF_HWEP_STR, LF_CALLRATE = range(11)
# kernels (profiling)
K_LOCUS_COUNT, K_LOCUS_FINALIZE, K_CALL_FILTER, K_LOCUS_FILTER, K_SYNTH = range(5)
KERNEL_NAMES = ['k_locus_count', 'k_locus_finalize', 'k_call_filter', 'k_locus_filter', 'k_synth',
                'k_assoc_scan', 'k_assoc_finalize', 'k_cf_reduce']
# filter ops
F_LT, F_GT, F_RATIO_GT, F_CALLED_LT, F_CALLED_SUM_LT, F_CALLED_EQ, F_CALLED_SUM_EQ, \
    F_CALLED_OUTSIDE_CI, F_AD_SUPPORT_LT = range(1, 10)
DT_I32, DT_F32 = 0, 1
DT_PLANAR = 0x100
# locus filter bits / counters
LOCF_CALLRATE, LOCF_HWE, LOCF_HETLOW, LOCF_HETHIGH, LOCF_EXTERN0 = 0, 1, 2, 3, 4
LOCF_NO_CALLS = 31
LC_TOTALCALLS, LC_PASS, LC_NO_CALLS, LC_FILTER0 = 0, 1, 2, 3
LC_HWE_ERRORS = 31
STATS_COUNT_ONLY = 1
STATS_TWIN = 2
TRK_N_STREAMS = 4


class Batch(C.Structure):
    _fields_ = [('n_loci', C.c_int32), ('n_samples', C.c_int32), ('ploidy', C.c_int32),
                ('n_groups', C.c_int32), ('n_alleles_total', C.c_int64),
                ('max_alleles', C.c_int32), ('n_pad_samples', C.c_int32),
                ('gt', C.c_void_p), ('locus_ploidy', C.c_void_p), ('allele_off', C.c_void_p),
                ('len_class', C.c_void_p), ('str_class', C.c_void_p),
                ('len_class_value', C.c_void_p), ('group_bits', C.c_void_p),
                ('row_stride', C.c_int32), ('n_class_runs', C.c_int32), ('class_runs', C.c_void_p)]


class StatsParams(C.Structure):
    _fields_ = [('nalleles_thresh', C.c_double), ('flags', C.c_int32), ('reserved', C.c_int32)]


class StatsOut(C.Structure):
    _fields_ = [('allele_count', C.c_void_p), ('locus_int', C.c_void_p), ('locus_f64', C.c_void_p)]


class Plane(C.Structure):
    _fields_ = [('data', C.c_void_p), ('dtype', C.c_int32), ('ncol', C.c_int32)]


class CallFilter(C.Structure):
    _fields_ = [('op', C.c_int32), ('plane_a', C.c_int32), ('col_a', C.c_int32),
                ('plane_b', C.c_int32), ('col_b', C.c_int32), ('col_a2', C.c_int32),
                ('thr', C.c_double)]


class CallOut(C.Structure):
    _fields_ = [('gt_out', C.c_void_p), ('filter_mask', C.c_void_p),
                ('sample_counters', C.c_void_p), ('sample_totaldp', C.c_void_p),
                ('sample_dp_missing', C.c_void_p), ('error', C.c_void_p),
                ('delta_allele_count', C.c_void_p), ('delta_locus_int', C.c_void_p),
                ('sample_totaldp_f64', C.c_void_p), ('filter_mask8', C.c_void_p)]


class LocusFilterSpec(C.Structure):
    _fields_ = [('min_callrate', C.c_double), ('min_hwep', C.c_double), ('min_het', C.c_double),
                ('max_het', C.c_double), ('use_length', C.c_int32), ('n_extern', C.c_int32),
                ('extern_bits', C.c_void_p)]


class LocusOut(C.Structure):
    _fields_ = [('locus_bits', C.c_void_p), ('loc_counters', C.c_void_p)]


class AssocParams(C.Structure):
    _fields_ = [('n_vec', C.c_int32), ('flags', C.c_int32), ('vec', C.c_void_p), ('sample_in', C.c_void_p),
                ('allele_len', C.c_void_p), ('rlen_class', C.c_void_p), ('non_major_cutoff', C.c_double)]


class AssocDosage(C.Structure):
    _fields_ = [('ap1', C.c_void_p), ('ap2', C.c_void_p), ('n_alt_cols', C.c_int32), ('reserved', C.c_int32),
                ('perm', C.c_void_p), ('dclass', C.c_void_p), ('dclass_value', C.c_void_p), ('best_class', C.c_void_p)]


ADC_COLS, ADL_COLS = 4, 8
DOS_TYPES = {'bestguess': 0, 'beagleap': 1, 'bestguess_norm': 2, 'beagleap_norm': 3}


class QcParams(C.Structure):
    _fields_ = [('sample_in', C.c_void_p), ('quality', C.c_void_p), ('ignore_no_call', C.c_int32), ('pad', C.c_int32)]


class QcOut(C.Structure):
    _fields_ = [('sample_calls', C.c_void_p), ('locus_calls', C.c_void_p), ('sample_qual_sum', C.c_void_p),
                ('sample_qual_n', C.c_void_p), ('locus_qual_sum', C.c_void_p), ('locus_qual_n', C.c_void_p)]


class AssocOut(C.Structure):
    _fields_ = [('locus_int', C.c_void_p), ('locus_f64', C.c_void_p), ('allele_count', C.c_void_p)]


# trk_inflate_blocks (include/trk.h): BGZF members inflated on the device
INFLATE_STREAM, INFLATE_OVERRUN, INFLATE_INPUT = 1, 2, 4


class InflateIn(C.Structure):
    _fields_ = [('comp', C.c_void_p), ('n_comp_bytes', C.c_int64), ('n_blocks', C.c_int32), ('pad_', C.c_int32),
                ('in_off', C.c_void_p), ('in_len', C.c_void_p), ('out_off', C.c_void_p), ('out_len', C.c_void_p)]


class InflateOut(C.Structure):
    _fields_ = [('text', C.c_void_p), ('flags', C.c_void_p)]


# trk_parse_samples (include/trk.h): the sample columns of a batch of records parsed on the device
PARSE_MAX_PLANES = 4
PARSE_INT, PARSE_FLOAT = 0, 1
PARSE_HOST, PARSE_COLUMNS, PARSE_PLOIDY = 1, 2, 4


class ParseIn(C.Structure):
    _fields_ = [('text', C.c_void_p), ('n_bytes', C.c_int64), ('n_records', C.c_int32), ('n_samples', C.c_int32),
                ('ploidy', C.c_int32), ('n_planes', C.c_int32), ('smp_off', C.c_void_p), ('line_end', C.c_void_p),
                ('gt_idx', C.c_void_p), ('plane_idx', C.c_void_p * PARSE_MAX_PLANES), ('plane_kind', C.c_int32 * PARSE_MAX_PLANES)]


FORMAT_MAX_FIELDS, FORMAT_MAX_FILTERS = 16, 7


class FormatIn(C.Structure):
    _fields_ = [('text', C.c_void_p), ('n_bytes', C.c_int64), ('n_records', C.c_int32), ('n_samples', C.c_int32),
                ('mask_stride', C.c_int32), ('plane_stride', C.c_int32), ('smp_off', C.c_void_p), ('line_end', C.c_void_p),
                ('field_kind', C.c_void_p), ('n_fields', C.c_void_p), ('ploidy', C.c_void_p), ('mask8', C.c_void_p),
                ('n_filters', C.c_int32), ('reserved', C.c_int32), ('filter_name', (C.c_char * 32) * FORMAT_MAX_FILTERS),
                ('filter_plane', C.c_void_p * FORMAT_MAX_FILTERS), ('filter_dtype', C.c_int32 * FORMAT_MAX_FILTERS)]


class FormatOut(C.Structure):
    _fields_ = [('rec_len', C.c_void_p), ('flags', C.c_void_p), ('out', C.c_void_p), ('out_off', C.c_void_p)]


class ParseOut(C.Structure):
    _fields_ = [('gt', C.c_void_p), ('phased', C.c_void_p), ('planes', C.c_void_p * PARSE_MAX_PLANES),
                ('locus_ploidy', C.c_void_p), ('flags', C.c_void_p)]


# associaTR scan: trk_assoc_out columns / status codes (include/trk.h)
ASSOC_MAX_VEC = 31        # one pass over the genotype tensor
ASSOC_MAX_VEC_WIDE = 126  # trk_assoc_scan: one MFMA pass up to 62 rows, pairs of 15-row groups beyond (TRK_ASSOC_MAX_VEC_WIDE)
AI_N_TESTED, AI_STATUS, AI_N_RALLELES, AI_RANK, AI_N_BAD, AI_N_HAPS, AI_COLS = 0, 1, 2, 3, 4, 5, 8
(AF_PVAL, AF_COEF, AF_SE, AF_RSQUARED, AF_GT_STD, AF_GT_MEAN, AF_TVALUE, AF_DF_RESID, AF_NONMAJOR) = range(9)
AF_COLS = 10
(AS_OK, AS_NO_CALLED, AS_ONE_ALLELE, AS_NON_MAJOR, AS_N_COVARS, AS_ZERO_VARIANCE, AS_COLLINEAR) = range(7)


PAIR_MAX_PROBES = 8


class PairInfo(C.Structure):
    _fields_ = [('n_probed', C.c_int32), ('placed', C.c_int32), ('probe_ms', C.c_float * PAIR_MAX_PROBES),
                ('kept_ms', C.c_float), ('have_a', C.c_int32), ('have_b', C.c_int32), ('n_jumps', C.c_int32),
                ('seconds', C.c_double),
                ('peak_extra_bytes', C.c_uint64), ('reserved', C.c_int32), ('pad_', C.c_int32)]


class SynthSpec(C.Structure):
    _fields_ = [('seed', C.c_uint64), ('n_loci', C.c_int32), ('n_samples', C.c_int32),
                ('allele_off', C.c_void_p), ('allele_cdf24', C.c_void_p), ('miss_thr16', C.c_void_p),
                ('inbreed_thr16', C.c_void_p), ('locus_base', C.c_int32), ('reserved', C.c_int32)]


# every symbol include/trk.h and include/trk_test.h declare (tests check that the library exports them)
EXPORTS = [
    'trk_init', 'trk_free', 'trk_last_error', 'trk_backend', 'trk_device_count', 'trk_device_info',
    'trk_dev_alloc', 'trk_dev_alloc_pair', 'trk_reserve_pair', 'trk_dev_free', 'trk_memcpy_h2d', 'trk_memcpy_d2h', 'trk_memcpy_d2d', 'trk_memset', 'trk_sync',
    'trk_timer_start', 'trk_timer_stop', 'trk_timer_elapsed_ms',
    'trk_profile_enable', 'trk_profile_get', 'trk_profile_reset',
    'trk_locus_stats', 'trk_locus_finalize', 'trk_call_filters', 'trk_locus_filters',
    'trk_comm_unique_id', 'trk_comm_init', 'trk_allreduce_sum_i64', 'trk_allgather',
    'trk_binomtest_two_sided', 'trk_binom_pmf', 'trk_binomtest_batch', 'trk_synth_fill', 'trk_synth_fill_gangstr', 'trk_test_set_option', 'trk_test_get_option',
    'trk_assoc_scan', 'trk_assoc_scan_dosage', 'trk_student_t_two_sided', 'trk_dosages', 'trk_qc_reduce', 'trk_parse_samples', 'trk_format_samples', 'trk_inflate_blocks', 'trk_inflate_hook', 'trk_inflate_hook_async', 'trk_inflate_text', 'trk_inflate_stats', 'trk_deflate_bgzf', 'trk_deflate_bound', 'trk_planarize', 'trk_pad_rows', 'trk_permute_columns', 'trk_stream_probe', 'trk_device_clocks', 'trk_stream_select', 'trk_stream_wait',
    'trk_host_alloc', 'trk_host_free', 'trk_memcpy_h2d_async', 'trk_memcpy_d2h_async', 'trk_queue_sync', 'trk_thread_queue', 'trk_exchange', 'trk_event_record', 'trk_event_wait',
]

_lib = None


class TrkError(RuntimeError):
    pass


# the sources libtrk.so is built from, in the order csrc/Makefile hashes them
_SOURCES = ['csrc/trk_api.hip', 'csrc/trk_assoc.hip', 'csrc/trk_binom.h', 'csrc/trk_deflate.hip', 'csrc/trk_hwe.hip', 'csrc/trk_inflate.hip', 'csrc/trk_internal.h', 'csrc/trk_kernels.hip',
            'csrc/trk_parse.hip', 'csrc/trk_qc.hip', 'csrc/trk_student.h', 'csrc/trk_vcf.cpp', '../include/trk.h', '../include/trk_test.h', '../include/trk_vcf.h']


def source_digest():
    import hashlib
    h = hashlib.sha256()
    for rel in _SOURCES:
        with open(os.path.join(_HERE, rel), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()


def _ensure_current():
    """A prebuilt libtrk.so must come from the sources next to it: the build leaves their SHA-256 in
    libtrk.so.srchash.  On a mismatch the library is rebuilt (hipcc cross-compiles anywhere); if that is not
    possible this raises rather than run a stale binary.  The check is skipped only in a lab process
    (TRK_LAB=1 TRK_SKIP_STALE_CHECK=1: tools/ running one build against another); a binary-only install -- no
    sources next to the library -- has nothing to compare and loads as it is."""
    if _knobs.lab('TRK_SKIP_STALE_CHECK'):
        return
    try:
        want = source_digest()
    except OSError:
        return                      # a binary-only install: nothing to compare against
    try:
        have = open(LIB_PATH + '.srchash').read().strip()
    except OSError:
        have = None
    if have == want:
        return
    import fcntl
    import shutil
    import subprocess
    if shutil.which('hipcc') is None or shutil.which('make') is None:
        raise TrkError("libtrk.so is older than its sources (digest %s, sources %s) and cannot be rebuilt here "
                       "(no hipcc/make)" % (have, want))
    with open(os.path.join(_HERE, 'csrc', '.build.lock'), 'w') as lock:     # one builder among parallel ranks
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            have = open(LIB_PATH + '.srchash').read().strip()
        except OSError:
            have = None
        if have != want:
            r = subprocess.run(['make', '-C', os.path.join(_HERE, 'csrc'), '-j4'], stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT)
            if r.returncode != 0:
                raise TrkError("libtrk.so is stale and the rebuild failed:\n%s" % r.stdout.decode()[-2000:])


def set_option(name, value):
    """An option of include/trk_test.h (forced code paths of the parity tests, A/B switches of tools/): ``value`` None
    unsets it.  Process-wide; the product itself sets none."""
    load().trk_test_set_option(name.encode(), None if value is None else str(value).encode())


class options:
    """``with options(TRK_CF_GENERIC=1): ...`` -- library options set for the block, restored after it."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        lib = load()
        self.old = {k: lib.trk_test_get_option(k.encode()) for k in self.kv}
        for k, v in self.kv.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, None if v is None else v.decode())
        return False


def load():
    """dlopen libtrk.so and declare signatures.  Raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TrkError("libtrk.so is not built (%s). Run `make -C trtools_amd/csrc` or "
                       "`python -c 'import __graft_entry__ as g; g.build()'`. "
                       "There is no CPU fallback." % LIB_PATH)
    alt = _knobs.lab('TRK_LIBTRK')          # A/B runs of two builds on one box (tools/ab_bench.sh)
    if alt:
        lib = C.CDLL(alt)
    else:
        _ensure_current()
        lib = C.CDLL(LIB_PATH)
    vp, i32, i64, u64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
    P = C.POINTER
    lib.trk_init.argtypes = [C.c_int, P(vp)]
    lib.trk_free.argtypes = [vp]
    lib.trk_free.restype = None
    lib.trk_last_error.argtypes = [vp]
    lib.trk_last_error.restype = C.c_char_p
    lib.trk_backend.argtypes = [vp]
    lib.trk_device_count.argtypes = [P(C.c_int)]
    lib.trk_device_info.argtypes = [vp, C.c_char_p, C.c_int, P(C.c_int), P(u64), C.c_char_p, C.c_int]
    lib.trk_dev_alloc.argtypes = [vp, C.c_size_t, P(vp)]
    lib.trk_dev_free.argtypes = [vp, vp]
    lib.trk_dev_alloc_pair.argtypes = [vp, C.c_size_t, i64, i64, i32, P(vp), i32, P(vp), P(vp), P(PairInfo)]
    lib.trk_reserve_pair.argtypes = [vp, C.c_size_t, P(PairInfo)]
    lib.trk_test_set_option.argtypes = [C.c_char_p, C.c_char_p]
    lib.trk_test_get_option.argtypes = [C.c_char_p]
    lib.trk_test_get_option.restype = C.c_char_p
    lib.trk_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.trk_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    lib.trk_memcpy_d2d.argtypes = [vp, vp, vp, C.c_size_t]
    lib.trk_memset.argtypes = [vp, vp, C.c_int, C.c_size_t]
    lib.trk_sync.argtypes = [vp]
    lib.trk_timer_start.argtypes = [vp, C.c_int]
    lib.trk_timer_stop.argtypes = [vp, C.c_int]
    lib.trk_timer_elapsed_ms.argtypes = [vp, C.c_int, P(C.c_float)]
    lib.trk_profile_enable.argtypes = [vp, C.c_int]
    lib.trk_profile_get.argtypes = [vp, C.c_int, P(i64), P(dbl)]
    lib.trk_profile_reset.argtypes = [vp]
    lib.trk_locus_stats.argtypes = [vp, P(Batch), P(StatsParams), P(StatsOut)]
    lib.trk_locus_finalize.argtypes = [vp, P(Batch), P(StatsParams), P(StatsOut)]
    lib.trk_call_filters.argtypes = [vp, P(Batch), P(Plane), C.c_int, P(CallFilter), C.c_int, C.c_int,
                                     P(CallOut)]
    lib.trk_locus_filters.argtypes = [vp, i32, P(StatsOut), P(LocusFilterSpec), P(LocusOut)]
    lib.trk_comm_unique_id.argtypes = [P(C.c_uint8)]
    lib.trk_comm_init.argtypes = [vp, C.c_int, C.c_int, P(C.c_uint8)]
    lib.trk_allreduce_sum_i64.argtypes = [vp, vp, C.c_size_t]
    lib.trk_allgather.argtypes = [vp, vp, vp, C.c_size_t]
    lib.trk_exchange.argtypes = [vp, vp, C.c_size_t, vp, vp, C.c_size_t]
    lib.trk_host_alloc.argtypes = [vp, C.c_size_t, P(vp)]
    lib.trk_host_free.argtypes = [vp, vp]
    lib.trk_memcpy_h2d_async.argtypes = [vp, vp, vp, C.c_size_t]
    lib.trk_memcpy_d2h_async.argtypes = [vp, vp, vp, C.c_size_t]
    lib.trk_thread_queue.argtypes = [vp, C.c_int]
    lib.trk_queue_sync.argtypes = [vp, C.c_int]
    lib.trk_event_record.argtypes = [vp, C.c_int]
    lib.trk_event_wait.argtypes = [vp, C.c_int]
    lib.trk_binomtest_two_sided.argtypes = [i64, i64, dbl]
    lib.trk_binomtest_two_sided.restype = dbl
    lib.trk_binomtest_batch.argtypes = [vp, vp, vp, vp, i64, vp, C.c_int32]
    lib.trk_binom_pmf.argtypes = [i64, i64, dbl]
    lib.trk_binom_pmf.restype = dbl
    lib.trk_assoc_scan.argtypes = [vp, P(Batch), P(AssocParams), P(AssocOut)]
    lib.trk_assoc_scan_dosage.argtypes = [vp, P(Batch), P(AssocParams), P(AssocDosage), P(AssocOut), vp, vp]
    lib.trk_planarize.argtypes = [vp, vp, vp, i64, C.c_int32]
    lib.trk_pad_rows.argtypes = [vp, vp, vp, i64, C.c_int32, C.c_int32, C.c_uint32]
    lib.trk_permute_columns.argtypes = [vp, vp, vp, vp, i64, C.c_int32, C.c_int32, C.c_int32]
    lib.trk_stream_probe.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, C.c_int32, P(C.c_float)]
    lib.trk_device_clocks.argtypes = [vp, P(C.c_int32), P(C.c_int32), P(C.c_int32)]
    lib.trk_stream_select.argtypes = [vp, C.c_int]
    lib.trk_stream_wait.argtypes = [vp, C.c_int, C.c_int]
    lib.trk_dosages.argtypes = [vp, P(Batch), vp, C.c_int, vp, vp, C.c_int, vp, vp]
    lib.trk_qc_reduce.argtypes = [vp, P(Batch), P(QcParams), P(QcOut)]
    lib.trk_parse_samples.argtypes = [vp, P(ParseIn), P(ParseOut)]
    lib.trk_inflate_blocks.argtypes = [vp, P(InflateIn), P(InflateOut)]
    lib.trk_inflate_hook.argtypes = [vp, P(vp), P(vp), P(vp)]
    lib.trk_inflate_text.argtypes = [vp, u64, i64, vp, u64]
    lib.trk_inflate_stats.argtypes = [vp, P(u64)]
    lib.trk_inflate_hook_async.argtypes = [vp, P(vp), P(vp)]
    lib.trk_deflate_bound.argtypes = [C.c_size_t]
    lib.trk_deflate_bound.restype = C.c_size_t
    lib.trk_deflate_bgzf.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, P(C.c_size_t)]
    lib.trk_format_samples.argtypes = [vp, P(FormatIn), P(FormatOut), C.c_int]
    lib.trk_student_t_two_sided.argtypes = [dbl, dbl]
    lib.trk_student_t_two_sided.restype = dbl
    lib.trk_synth_fill.argtypes = [vp, P(SynthSpec), vp, vp, vp, vp, vp]
    lib.trk_synth_fill_gangstr.argtypes = [vp, P(SynthSpec), vp, vp, vp, vp, vp, vp, vp]
    _lib = lib
    return lib
