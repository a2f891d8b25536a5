"""statSTR on further argument sets (the reference's own tests and every caller's fixture VCF with every
statistic on): return codes and tables byte for byte against what the REAL reference wrote here
(tools/gen_golden_statstr_more.py -> tests/golden/statstr_more)."""
import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from statstr_more_cases import CASES, OUT      # noqa: E402

WANT = json.load(open(os.path.join(OUT, 'results.json')))['rc']


# argument sets that may leave the batch pipeline, and why; every other successful run stays on it
PER_RECORD_OK = {}      # (round 4: LongTR's symbolic <DEL> alleles are harmonised natively: nothing leaves the pipeline)


def run_and_check(outdir):
    import gen_golden_statstr_more as gm
    from trtools_amd.statSTR import statSTR
    paths = []

    def main(args):
        try:
            return statSTR.main(args)
        finally:
            paths.append(dict(statSTR.LAST_RUN))
    rcs = gm.run_cases(main, outdir)
    for (name, *_), pth in zip(CASES, paths):
        if WANT[name] == 0 and name not in PER_RECORD_OK:
            assert pth.get('path') == 'batch', (name, pth)
    bad = {n: (rcs[n], WANT[n]) for n in WANT if rcs[n] != WANT[n]}
    assert not bad, bad
    n = 0
    for name, *_ in CASES:
        if WANT[name] != 0:
            continue
        want = gzip.open(os.path.join(OUT, name + '.tab.gz'), 'rt').read()
        got = open(os.path.join(outdir, name + '.tab')).read()
        if got != want:
            gl, wl = got.split('\n'), want.split('\n')
            first = next(i for i in range(max(len(gl), len(wl))) if i >= len(gl) or i >= len(wl) or gl[i] != wl[i])
            raise AssertionError((name, first, gl[first][:300] if first < len(gl) else None,
                                  wl[first][:300] if first < len(wl) else None))
        n += 1
    assert n == 23


def test_more_reference_cases_host_layer_cpu(tmp_path):
    from trtools_amd import runtime
    from oracle_compute import OracleCompute
    old = runtime.set_compute(OracleCompute())
    try:
        run_and_check(str(tmp_path))
    finally:
        runtime.set_compute(old)


@pytest.mark.gpu
def test_more_reference_cases_gpu(tmp_path):
    from trtools_amd import runtime
    runtime.set_compute(None)
    run_and_check(str(tmp_path))
