"""Cohorts whose size is not a multiple of four samples: the compute seam pads every row with no-call samples
(trk_batch.n_pad_samples) so that the streaming kernels apply.  Results must equal the unpadded run (per-call
kernels) bit for bit, for statSTR statistics (with and without sample groups), the dumpSTR pass and the association
scan."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def comp():
    from trtools_amd.compute import DeviceCompute
    c = DeviceCompute()
    yield c
    c.eng.close()


def _host_batch(S, n_loci=40, seed=0, groups=False):
    from trtools_amd.batch import HostBatch
    rng = np.random.default_rng(seed)
    lens, strs, gts = [], [], []
    for l in range(n_loci):
        A = 1 + int(rng.integers(0, 6))
        ss = ['AC' * (3 + i) for i in range(A)]
        if A > 2 and l % 5 == 0:
            ss[2] = 'AC' * 3 + 'GG'          # same length as another allele, different sequence
        strs.append(ss)
        lens.append([len(s) / 2 for s in ss])
        g = rng.integers(0, A, size=(S, 2)).astype(np.int16)
        g[rng.random(S) < 0.08] = -1
        g[rng.random(S) < 0.03, 1] = -1
        gts.append(g)
    gt = np.stack(gts)
    gt[3] = -1
    gb = None
    if groups:
        gb = rng.integers(0, 4, size=S).astype(np.uint8)
    return HostBatch(gt, np.full(n_loci, 2, dtype=np.uint8), lens, strs, group_bits=gb, n_groups=2), rng


def _both(fn):
    old = os.environ.get('TRK_PAD_SAMPLES')
    try:
        os.environ['TRK_PAD_SAMPLES'] = '1'
        a = fn()
        os.environ['TRK_PAD_SAMPLES'] = '0'
        b = fn()
    finally:
        if old is None:
            del os.environ['TRK_PAD_SAMPLES']
        else:
            os.environ['TRK_PAD_SAMPLES'] = old
    return a, b


@pytest.mark.parametrize('S', [1001, 1002, 1003, 7])
@pytest.mark.parametrize('groups', [False, True])
def test_statistics(comp, S, groups):
    hb, _ = _host_batch(S, seed=S, groups=groups)
    a, b = _both(lambda: comp.locus_stats(hb))
    assert np.array_equal(a.allele_count, b.allele_count)
    assert np.array_equal(a.locus_int, b.locus_int)
    assert np.array_equal(a.locus_f64, b.locus_f64, equal_nan=True)
    from trtools_amd import _lib as L
    if not groups:
        assert np.all(a.locus_int[0][:, L.LI_N_SAMPLES] == S)


@pytest.mark.parametrize('S', [1001, 1003, 2])
def test_dumpstr_pass(comp, S):
    from trtools_amd import _lib as L
    hb, rng = _host_batch(S, seed=10 + S)
    dp = rng.integers(0, 60, size=(hb.n_loci, S)).astype(np.int32)
    dp[rng.random((hb.n_loci, S)) < 0.02] = np.iinfo(np.int32).min
    q = rng.random((hb.n_loci, S)).astype(np.float32)
    qexp = rng.random((hb.n_loci, S, 3)).astype(np.float32)
    filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_LT, plane_a=1, thr=0.3),
               dict(op=L.F_CALLED_SUM_LT, plane_a=2, col_a=1, col_a2=2, thr=0.5)]
    spec = dict(min_callrate=0.8, min_hwep=0.01, min_het=0.05, max_het=0.9)
    (ca, sa, ba, la), (cb, sb, bb, lb) = _both(lambda: comp.dumpstr_batch(hb, [dp, q, qexp], filters, 0, spec))
    for name in ('gt_out', 'mask', 'sample_counters', 'totaldp', 'dp_missing'):
        x, y = getattr(ca, name), getattr(cb, name)
        assert x.shape == y.shape and np.array_equal(x, y), name
    assert ca.gt_out.shape[1] == S and ca.sample_counters.shape[1] == S
    assert np.array_equal(sa.allele_count, sb.allele_count)
    assert np.array_equal(sa.locus_int, sb.locus_int)
    assert np.array_equal(sa.locus_f64, sb.locus_f64, equal_nan=True)
    assert np.array_equal(ba, bb) and np.array_equal(la, lb)


@pytest.mark.parametrize('S,subset', [(1001, False), (1003, True)])
def test_association_scan(comp, S, subset):
    hb, rng = _host_batch(S, seed=20 + S)
    vec = rng.standard_normal((3, S))
    sin = (rng.random(S) < 0.8) if subset else None
    if sin is not None:
        vec[:, ~sin] = 0.0

    def run():
        if hasattr(comp, '_assoc_vec'):
            del comp._assoc_vec
        return comp.assoc_batch(hb, vec.copy(), sin, 0.0)
    a, b = _both(run)
    assert np.array_equal(a.locus_int, b.locus_int)
    assert np.array_equal(a.allele_count, b.allele_count)
    assert np.allclose(a.locus_f64, b.locus_f64, rtol=1e-11, atol=1e-13, equal_nan=True)


def test_pad_rows_entry(comp):
    """trk_pad_rows: the padding samples are appended on the device (no host copy of the tensor); genotypes get -1,
    int32 planes the missing value, float32 planes nan, multi-column planes whole padding samples."""
    eng = comp.eng
    rng = np.random.default_rng(5)
    gt = rng.integers(-1, 5, size=(37, 1000, 2)).astype(np.int16)
    dp = rng.integers(0, 99, size=(37, 1000)).astype(np.int32)
    q3 = rng.random((37, 1000, 3)).astype(np.float32)
    for arr, fill in ((gt, -1), (dp, np.iinfo(np.int32).min), (q3, np.nan)):
        d = eng.pad_samples(eng.upload(arr), 24)
        got = d.get()
        d.free()
        assert got.shape == (37, 1024) + arr.shape[2:]
        assert np.array_equal(got[:, :1000], arr)
        pad = got[:, 1000:]
        assert np.all(np.isnan(pad)) if fill is np.nan else np.all(pad == fill)
    d = eng.upload(dp)
    assert eng.pad_samples(d, 0) is d
    d.free()


@pytest.mark.parametrize('S', [1000, 2500])
def test_rows_on_cache_line_boundaries(comp, S):
    """From 512 samples on the statSTR / dumpSTR passes pad every row to a multiple of 32 samples (128 bytes;
    TRK_ROW_ALIGN): same results as the dense layout, bit for bit, and nothing of the padding comes back."""
    from trtools_amd import _lib as L
    hb, rng = _host_batch(S, seed=77 + S)
    assert comp._n_pad(hb, rows=True) == (-S) % 32 and comp._n_pad(hb) == 0
    dp = rng.integers(0, 60, size=(hb.n_loci, S)).astype(np.int32)
    q = rng.random((hb.n_loci, S)).astype(np.float32)
    filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=50), dict(op=L.F_LT, plane_a=1, thr=0.3)]
    spec = dict(min_callrate=0.8, min_hwep=0.01, min_het=0.05, max_het=0.9)

    def both(fn):
        old = os.environ.get('TRK_ROW_ALIGN')
        try:
            os.environ['TRK_ROW_ALIGN'] = '32'
            a = fn()
            os.environ['TRK_ROW_ALIGN'] = '4'
            b = fn()
        finally:
            os.environ.pop('TRK_ROW_ALIGN', None)
            if old is not None:
                os.environ['TRK_ROW_ALIGN'] = old
        return a, b
    (ca, sa, ba, la), (cb, sb, bb, lb) = both(lambda: comp.dumpstr_batch(hb, [dp, q], filters, 0, spec))
    for name in ('gt_out', 'mask', 'sample_counters', 'totaldp', 'dp_missing'):
        x, y = getattr(ca, name), getattr(cb, name)
        assert x.shape == y.shape and np.array_equal(x, y), name
    assert ca.gt_out.shape[1] == S and ca.sample_counters.shape[1] == S
    assert np.array_equal(sa.locus_int, sb.locus_int) and np.array_equal(sa.locus_f64, sb.locus_f64, equal_nan=True)
    assert np.array_equal(ba, bb) and np.array_equal(la, lb)
    assert np.all(sa.locus_int[0][:, L.LI_N_SAMPLES] == S)
    a, b = both(lambda: comp.locus_stats(hb))
    assert np.array_equal(a.locus_int, b.locus_int) and np.array_equal(a.locus_f64, b.locus_f64, equal_nan=True)
    assert np.array_equal(a.allele_count, b.allele_count)
