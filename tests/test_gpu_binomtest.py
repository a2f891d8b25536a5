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
