"""GPU parity of the lane-pair exact binomial test (k_hwe_test's routine, trk_binomtest_batch) -- the third-party call at
the end of the reference's HWE statistic, scipy.stats.binomtest(num_hom, n, exp_hom_frac).pvalue (utils.py:334-338):
bit for bit against the serial routine in one lane on the device, and against scipy itself."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from trtools_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def triples(seed, count):
    """HWE-shaped triples (k near n p, every distance from it), tiny / huge n, p at and near 0 and 1, k at 0 / n / the
    mean itself and one off it, means that are integers."""
    rng = np.random.default_rng(seed)
    n = np.where(rng.random(count) < 0.2, rng.integers(1, 30, count), rng.integers(30, 20000, count)).astype(np.int64)
    p = rng.random(count)
    shape = rng.integers(0, 10, count)
    p = np.where(shape == 0, rng.random(count) * 1e-3, p)
    p = np.where(shape == 1, 1.0 - rng.random(count) * 1e-3, p)
    p = np.where(shape == 2, rng.integers(0, 11, count) / 10.0, p)          # 0, 1 and means that are whole numbers
    sd = np.sqrt(n * p * (1 - p)) + 1.0
    k = np.rint(n * p + rng.normal(size=count) * sd * rng.choice([0.05, 1.0, 4.0, 12.0], size=count)).astype(np.int64)
    k = np.where(shape == 3, np.floor(n * p).astype(np.int64) + rng.integers(-1, 3, count), k)   # at / beside the mean
    k = np.where(shape == 4, rng.integers(0, 2, count) * n, k)              # 0 or n
    k = np.where(shape == 5, rng.integers(0, 20000, count) % (n + 1), k)    # anywhere
    far = np.rint(n * p + rng.choice([-1, 1], count) * rng.uniform(20, 120, count) * sd).astype(np.int64)
    k = np.where(shape == 6, far, k)                                        # 20-120 sd out: Newton steps before the walk
    return np.clip(k, 0, n), n, p


def test_pair_routine_equals_the_serial_one_bit_for_bit(eng):
    for seed in (1, 2, 3):
        k, n, p = triples(seed, 60000)
        dev = eng.binomtest_batch(k, n, p)
        one = eng.binomtest_batch(k, n, p, lanes=1)
        assert np.array_equal(dev, one), np.flatnonzero(dev != one)[:10]
        assert np.all((dev >= 0) & (dev <= 1))
        # the host build of the same source (libm instead of the device's log / exp / reciprocal)
        host = np.array([eng.binomtest(a, b, c) for a, b, c in zip(k[:5000], n[:5000], p[:5000])])
        assert np.all(np.abs(dev[:5000] - host) <= 1e-11 * host + 1e-300)


def test_pair_routine_against_scipy(eng):
    from scipy.stats import binomtest
    k, n, p = triples(11, 4000)
    dev = eng.binomtest_batch(k, n, p)
    for i in range(len(k)):
        want = binomtest(int(k[i]), int(n[i]), float(p[i])).pvalue
        if want < 1e-250:       # scipy's own cdf underflows to 0 there and its p-value falls BELOW pmf(k)
            assert dev[i] < 1e-240
            continue
        assert abs(dev[i] - want) <= 1e-9 * want + 1e-300, (k[i], n[i], p[i], dev[i], want)


def test_invalid_triples_and_empty(eng):
    out = eng.binomtest_batch([3, -1, 5, 2], [0, 5, 4, 5], [0.5, 0.5, 0.5, 1.5])
    assert np.all(np.isnan(out))
    assert eng.binomtest_batch([], [], []).shape == (0,)


def test_p_values_down_to_the_underflow_boundary_against_scipy(eng):
    """What a statSTR `hwep` cell holds for very unlikely loci.  Along one family of triples
    (n = 20 000, p = 0.5, k moving away from the mean) the exact p-value falls from 1e-5 to below the smallest
    float64:
      * down to 1e-280 the device agrees with scipy RELATIVE to the value (1e-9), not merely to an absolute 1e-9;
      * where pmf(k) itself underflows (below ~1e-308: |z| beyond ~37.6) both tails are sums of zeros: the device
        returns exactly 0.0 -- scipy returns 0.0 there too, or a denormal-range number out of the incomplete-beta
        tail (< 1e-300).  statSTR's '{:.p}'-formatted cell reads '0.0' on the device side where the reference may
        print e.g. '1e-310' -- inside north_star's 1e-9, and the boundary is this test's to document;
      * in between (1e-308 ... 1e-280: denormal intermediate sums) both sides are only required to be that small."""
    from scipy.stats import binomtest
    n, p = 20000, 0.5
    ks = np.unique(np.concatenate([np.arange(10320, 13000, 37), np.arange(12600, 12760)])).astype(np.int64)
    dev = eng.binomtest_batch(ks, np.full(ks.size, n, dtype=np.int64), np.full(ks.size, p))
    seen_rel = seen_zero = 0
    last = 1.0
    for k, d in zip(ks, dev):
        want = binomtest(int(k), n, p).pvalue
        assert d <= last * (1 + 1e-12)          # monotone in the distance from the mean
        last = max(d, 0.0)
        if want >= 1e-280:
            assert abs(d - want) <= 1e-9 * want, (k, d, want)
            seen_rel += 1
        elif want >= 1e-300:
            assert d < 1e-270, (k, d, want)
        else:
            assert d < 1e-290, (k, d, want)
            seen_zero += d == 0.0
    assert seen_rel > 40 and seen_zero > 20
    # the statistic's formatting at the boundary: what the table shows
    assert '{:.4}'.format(float(dev[-1])) == '0.0'
