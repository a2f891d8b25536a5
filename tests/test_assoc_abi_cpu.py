"""Host-callable scalar of the associaTR path: trk_student_t_two_sided == 2 * scipy.stats.t.sf
(the third-party call behind statsmodels' p-values, associaTR.py:283).  No GPU needed."""
import os
import sys

import numpy as np
import scipy.stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_student_t_tail_matches_scipy():
    from trtools_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(7)
    worst = 0.0
    for v in [1, 2, 3, 5, 8, 10, 17, 30, 39, 47, 48, 100, 333, 1000, 5000, 9988, 9999, 1e5, 1e6, 1e7]:
        ts = np.concatenate([rng.random(60) * 3, rng.random(30) * 40, [1e-8, 1e-3, 0.5, 1, 1.7, 1.74, 2, 5, 10, 37, 60, 200]])
        for t in ts:
            want = 2 * scipy.stats.t.sf(t, v)
            got = lib.trk_student_t_two_sided(float(t), float(v))
            assert got == lib.trk_student_t_two_sided(-float(t), float(v))
            if want > 1e-300:
                worst = max(worst, abs(got / want - 1))
    assert worst < 1e-11, worst
    assert lib.trk_student_t_two_sided(0.0, 10.0) == 1.0
    assert np.isnan(lib.trk_student_t_two_sided(float('nan'), 10.0))
    assert lib.trk_student_t_two_sided(float('inf'), 10.0) == 0.0


def test_assoc_symbols_and_tables():
    from trtools_amd import _lib, synth
    lib = _lib.load()
    assert hasattr(lib, 'trk_assoc_scan')
    alen, rcls = synth.pack_assoc_tables([[10.0, 10.004, 10.006, 12.0, 10.0]], 2)
    assert list(alen) == [10.0, 10.004, 10.006, 12.0, 10.0]
    # length classes ascending: 10.0, 10.004, 10.006, 12.0 -> rounded 10.0, 10.0, 10.01, 12.0
    assert list(rcls[:4]) == [0, 0, 1, 2]
