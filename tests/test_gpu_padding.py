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
