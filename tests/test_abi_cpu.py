"""CPU-side checks of the C-ABI library: it loads, exports every symbol that
include/trk.h declares, its host scalar helpers match scipy, and the product
refuses to run without a GPU (no CPU fallback).  No device compute here."""
import os
import re

import pytest

from helpers import load_golden, unjf, close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from trtools_amd import _lib as L
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L, L.load()


def test_library_exports_every_declared_symbol():
    L, lib = _lib()
    hdr = open(os.path.join(ROOT, 'include', 'trk.h')).read() + open(os.path.join(ROOT, 'include', 'trk_test.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = sorted(set(re.findall(r'\b(trk_[a-z0-9_]+)\s*\(', hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "libtrk.so does not export %s" % name
    assert sorted(L.EXPORTS) == declared


def test_shipped_library_is_built_from_the_sources_next_to_it():
    """The build leaves the SHA-256 of every source file next to libtrk.so; load() refuses (or rebuilds) a binary
    whose digest differs, so a stale prebuilt library cannot be what the GPU tests exercise."""
    L, _ = _lib()
    assert open(L.LIB_PATH + '.srchash').read().strip() == L.source_digest()
    mk = open(os.path.join(ROOT, 'trtools_amd', 'csrc', 'Makefile')).read()
    listed = re.search(r'^SRCS = (.*?)\n\n', mk, flags=re.S | re.M).group(1).replace('\\\n', ' ').split()
    assert [os.path.normpath(os.path.join('csrc', f)) for f in listed] == [os.path.normpath(f) for f in L._SOURCES]


def test_binomtest_host_entry_matches_scipy_vectors():
    L, lib = _lib()
    for k, n, p, pv in load_golden('binomtest_vectors.json')['cases']:
        got = lib.trk_binomtest_two_sided(k, n, p)
        assert close(got, unjf(pv), 1e-9, 1e-300), (k, n, p, got, pv)


def test_struct_sizes_match_header():
    """ctypes mirrors of the ABI structs (64-bit Linux layout)."""
    import ctypes as C
    L, _ = _lib()
    assert C.sizeof(L.Batch) == 4 * 4 + 8 + 8 + 7 * 8 + 2 * 4 + 8
    assert C.sizeof(L.CallFilter) == 6 * 4 + 8
    assert C.sizeof(L.Plane) == 16
    assert C.sizeof(L.LocusFilterSpec) == 4 * 8 + 8 + 8
    assert C.sizeof(L.SynthSpec) == 8 + 8 + 4 * 8 + 8


def test_engine_fails_loudly_without_gpu():
    import ctypes as C
    L, lib = _lib()
    n = C.c_int()
    lib.trk_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    from trtools_amd.engine import Engine
    with pytest.raises(L.TrkError):
        Engine(0)


def test_binomtest_host_entry_random_cases_match_scipy():
    """The exact two-sided test behind the HWE column (utils.py:334-338) against scipy itself on a sweep of n, p and k
    (edges, the mean and its neighbours, random draws around the mean): the guided search for the far-side crossing
    (trk_binom.h binom_boundary) must land where scipy's bisection does."""
    import numpy as np
    import scipy.stats as st
    from trtools_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(3)
    cases = []
    for n in [1, 2, 3, 5, 10, 37, 100, 999, 2000, 10000]:
        for p in [1e-6, 0.003, 0.05, 0.25, 0.5, 0.5000001, 0.77, 0.97, 0.999999]:
            ks = set([0, 1, n // 2, n - 1, n, int(n * p), int(n * p) + 1, max(0, int(n * p) - 1)] +
                     [int(x) for x in rng.integers(0, n + 1, size=6)])
            cases += [(k, n, p) for k in ks if 0 <= k <= n]
    for _ in range(800):
        n = int(rng.integers(1, 20001))
        p = float(rng.random())
        k = int(np.clip(rng.normal(n * p, 3 * np.sqrt(n * p * (1 - p)) + 1), 0, n))
        cases.append((k, n, p))
    for k, n, p in cases:
        got = lib.trk_binomtest_two_sided(k, n, p)
        want = st.binomtest(k, n, p).pvalue
        assert abs(got - want) <= 1e-9 * max(want, 1e-300) or abs(got - want) < 1e-300, (k, n, p, got, want)


def test_library_exports_every_symbol_of_the_reader_header():
    """include/trk_vcf.h (native reader, batch harmoniser, row / record writers): every declared entry point is exported."""
    _, lib = _lib()
    hdr = open(os.path.join(ROOT, 'include', 'trk_vcf.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = sorted(set(re.findall(r'\b(trk_vcf_[a-z0-9_]+)\s*\(', hdr)))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), "libtrk.so does not export %s" % name
