"""dumpSTR on further argument sets of the reference's own test-suite (EH input, region filters, a second
round on dumpSTR output, Beagle-imputed inputs, pre-existing fields, broken input): return codes, logs byte
for byte and output VCFs (reference comparator rules) against what the REAL reference produced here
(tools/gen_golden_dumpstr_more.py -> tests/golden/dumpstr_more)."""
import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from dumpstr_more_cases import CASES, OUT      # noqa: E402
from vcf_compare import compare_vcfs           # noqa: E402

WANT = json.load(open(os.path.join(OUT, 'results.json')))['rc']


PATHS = {}
# the argument sets that may leave the batch pipeline, and why; every other successful run must go through it for
# every batch (the fallbacks used to be silent)
PER_RECORD_OK = {}      # (round 4: a FORMAT/FILTER field that is already there is replaced in place by the native writer)


def run_all(outdir):
    import gen_golden_dumpstr_more as gm
    from trtools_amd.dumpSTR import dumpSTR

    def main(args):
        try:
            return dumpSTR.main(args)
        finally:
            PATHS[sys.argv[2]] = dict(dumpSTR.LAST_RUN)
    PATHS.clear()
    return gm.run_cases(main, outdir)


def check(outdir, rcs):
    bad = {n: (rcs[n], WANT[n]) for n in WANT if rcs[n] != WANT[n]}
    assert not bad, bad
    n_vcf = 0
    for name, _, kw in CASES:
        if WANT[name] != 0:
            continue
        for ext in ('.samplog.tab', '.loclog.tab'):
            assert open(os.path.join(outdir, name + ext)).read() == open(os.path.join(OUT, name + ext)).read(), (name, ext)
        gold = os.path.join(outdir, name + '.gold.vcf')
        with gzip.open(os.path.join(OUT, name + '.vcf.gz'), 'rb') as fin, open(gold, 'wb') as fout:
            fout.write(fin.read())
        assert compare_vcfs(os.path.join(outdir, name + '.vcf'), gold) == [], name
        n_vcf += 1
    assert n_vcf == 16
    for name in WANT:
        if WANT[name] == 0 and name not in PER_RECORD_OK:
            assert PATHS[name].get('path') == 'batch' and PATHS[name]['batches'] >= 1, (name, PATHS[name])


def test_more_reference_cases_host_layer_cpu(tmp_path):
    from trtools_amd import runtime
    from oracle_compute import OracleCompute
    old = runtime.set_compute(OracleCompute())
    try:
        check(str(tmp_path), run_all(str(tmp_path)))
    finally:
        runtime.set_compute(old)


@pytest.mark.gpu
def test_more_reference_cases_gpu(tmp_path):
    from trtools_amd import runtime
    runtime.set_compute(None)
    check(str(tmp_path), run_all(str(tmp_path)))
