"""The small-batch statSTR pass in two launches (k_locus_count_v3<4,4,true>: the finaliser as the count kernel's
epilogue; k_hwe_test_slots) against the five-launch chain (count, counter reset, k_locus_finalize, k_hwe_test,
k_hwe_test_serial): every output bit for bit, and the chain itself against the oracle.

Reference: the per-record statistics of statSTR.py:575-639 through tr_harmonizer.py:1420-1560 and utils.py:139-338."""
import os

import numpy as np
import pytest

from helpers import lab_env

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from trtools_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _fetch(res):
    return res.allele_count.get().copy(), res.locus_int.get().copy(), res.locus_f64.get().copy()


def _both(eng, b, thresh):
    with lab_env(TRK_FUSED_STATS='0'):
        chain = _fetch(eng.locus_stats(b, nalleles_thresh=thresh))
    fused = _fetch(eng.locus_stats(b, nalleles_thresh=thresh))
    return chain, fused


def _same(chain, fused):
    assert np.array_equal(chain[0], fused[0])
    assert np.array_equal(chain[1], fused[1]), np.argwhere(chain[1] != fused[1])[:5]
    a, c = chain[2].view(np.uint64), fused[2].view(np.uint64)
    nan_both = np.isnan(chain[2]) & np.isnan(fused[2])
    bad = (a != c) & ~nan_both
    assert not bad.any(), (np.argwhere(bad)[:5], chain[2][bad][:5], fused[2][bad][:5])


@pytest.mark.parametrize("S", [4, 60, 64, 252, 1000, 2048])
@pytest.mark.parametrize("max_alt", [0, 3, 14, 40])
def test_fused_pass_equals_the_chain(eng, S, max_alt):
    from test_gpu_stats import _random_batch, check_against_oracle
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    rng = np.random.default_rng(1000 * max_alt + S)
    n_loci = 45          # the last wave of four loci holds one
    gt, lens, strs, _, (off, lc, sc, cv) = _random_batch(rng, n_loci, S, 2, max_alt)
    for l in range(n_loci):
        r = rng.random(S)
        if l % 3 == 0:
            gt[l][r < 0.05, 1] = -2
            gt[l][(r >= 0.10) & (r < 0.12)] = -2
        gt[l][(r >= 0.05) & (r < 0.08)] = (-1, -2)
        gt[l][(r >= 0.12) & (r < 0.16)] = -1
        gt[l][(r >= 0.16) & (r < 0.19), 1] = -1
    gt[3] = -1                 # nothing called: ValueError status
    gt[4] = -2
    gt[7] = 0                  # one allele, everybody homozygous
    if S >= 60:
        gt[8][:, 0] = 0        # every call heterozygous or 0/0
    b = eng.make_batch(gt, off, lc, sc, cv)
    chain, fused = _both(eng, b, 0.02)
    _same(chain, fused)
    check_against_oracle(orc, L, *fused, off, gt, lens, strs, [None], 0.02)


def test_fused_pass_on_a_synthetic_cohort_and_padding(eng):
    """configs[1]'s shape at a tenth of its loci, rows padded with no-call samples (trk_batch.n_pad_samples)."""
    from trtools_amd.synth import SynthBatch
    sb = SynthBatch(eng, 1000, 1000, seed=77, planes=())
    chain, fused = _both(eng, sb.batch, 0.01)
    _same(chain, fused)
    sb.pad_rows(32)
    chain_p, fused_p = _both(eng, sb.batch, 0.01)
    _same(chain_p, fused_p)
    _same(chain, fused_p)      # the padding leaves every statistic alone


def test_out_of_range_indices_and_negative_threshold(eng):
    from test_gpu_stats import _random_batch
    rng = np.random.default_rng(5)
    gt, lens, strs, _, (off, lc, sc, cv) = _random_batch(rng, 17, 128, 2, 6)
    gt[2][5] = (30, 0)
    gt[9][7] = (1, 99)
    b = eng.make_batch(gt, off, lc, sc, cv)
    for thresh in (0.0, -1.0, 0.5, 1.0):
        _same(*_both(eng, b, thresh))


def test_large_batches_and_the_limit_knob(eng):
    """The fused pass serves every batch of short rows; TRK_FUSED_STATS=<loci> caps it (0: never): same bits."""
    from trtools_amd.synth import SynthBatch
    sb = SynthBatch(eng, 40000, 256, seed=3, planes=())
    chain, fused = _both(eng, sb.batch, 0.01)
    _same(chain, fused)
    with lab_env(TRK_FUSED_STATS='1000'):
        capped = _fetch(eng.locus_stats(sb.batch))
    _same(chain, capped)


@pytest.mark.parametrize("ploidy,n_groups,S", [(2, 0, 5000), (2, 3, 1000), (1, 0, 257), (3, 2, 300), (2, 8, 64), (4, 0, 90)])
def test_cooperative_finaliser_equals_the_one_thread_finaliser(eng, ploidy, n_groups, S):
    """k_locus_finalize_coop (sixteen lanes per (group, locus), what batches of up to 4096 rows and <= 64 alleles per
    locus take after the general count kernels) against k_locus_finalize (TRK_FIN_COOP=0): every output bit for bit --
    ploidy 1-4 with a ploidy table, sample groups, the wave-per-locus count kernel's long rows."""
    from test_gpu_stats import _random_batch
    rng = np.random.default_rng(31 * ploidy + 7 * n_groups + S)
    n_loci = 37
    gt, lens, strs, lp, (off, lc, sc, cv) = _random_batch(rng, n_loci, S, ploidy, 11, with_low=(ploidy > 2))
    gt[2] = -1
    gb = None
    if n_groups:
        gb = np.zeros(S, dtype=np.uint8)
        for g in range(n_groups):
            gb |= (rng.random(S) < 0.4).astype(np.uint8) << g
    b = eng.make_batch(gt, off, lc, sc, cv, locus_ploidy=lp if ploidy != 2 else None, group_bits=gb,
                       n_groups=max(n_groups, 1))
    with lab_env(TRK_FUSED_STATS='0'):
        with lab_env(TRK_FIN_COOP='0'):
            one = _fetch(eng.locus_stats(b, nalleles_thresh=0.03))
        coop = _fetch(eng.locus_stats(b, nalleles_thresh=0.03))
    _same(one, coop)
