"""dumpSTR on small synthetic HipSTR / GangSTR VCFs: all three outputs (.vcf, .samplog.tab,
.loclog.tab) against the files the IMPORTED reference produced from the same inputs
(tools/gen_golden_dumpstr.py).  The .vcf comparison pins the per-call FILTER text
('<name>_<%g value>', NOCALL, PASS), genotype/FORMAT nulling, locus FILTER and recomputed INFO."""
import os
import sys

import pytest

from helpers import GOLDEN

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
G = os.path.join(GOLDEN, 'dumpstr_synth')


def _cases():
    import gen_golden_dumpstr as gg
    return gg


def _run(tmp_path, compute, name):
    from trtools_amd import runtime
    from trtools_amd.dumpSTR import dumpSTR
    gg = _cases()
    caller, kw = gg.CASES[name]
    out = str(tmp_path / name)
    old = runtime.set_compute(compute)
    argv = sys.argv
    sys.argv = ['dumpSTR', '--synthetic-golden', name]
    try:
        assert dumpSTR.main(gg.make_args(out, os.path.join(G, 'synth_%s.vcf' % caller), caller, **kw)) == 0
    finally:
        sys.argv = argv
        runtime.set_compute(old)
    for ext in ('.samplog.tab', '.loclog.tab', '.vcf'):
        got = open(out + ext).read().split('\n')
        want = open(os.path.join(G, name + ext)).read().split('\n')
        if ext == '.vcf':
            # INFO HET / HWEP are floats printed with %g: compare numerically (<= 1e-6 rel), rest exact
            assert len(got) == len(want)
            for i, (a, b) in enumerate(zip(got, want)):
                if a == b:
                    continue
                fa, fb = a.split('\t'), b.split('\t')
                assert fa[:7] == fb[:7] and fa[8:] == fb[8:], (name, i)
                ia = dict(x.split('=', 1) if '=' in x else (x, '') for x in fa[7].split(';'))
                ib = dict(x.split('=', 1) if '=' in x else (x, '') for x in fb[7].split(';'))
                assert ia.keys() == ib.keys()
                for k in ia:
                    if ia[k] != ib[k]:
                        assert k in ('HET', 'HWEP') and abs(float(ia[k]) - float(ib[k])) <= 1e-6 * abs(float(ib[k])), (name, i, k)
        else:
            assert got == want, (name, ext)


@pytest.mark.parametrize("name", ['hipstr_all', 'hipstr_uselength_drop', 'gangstr_all'])
def test_host_layer_cpu(tmp_path, name):
    from oracle_compute import OracleCompute
    _run(tmp_path, OracleCompute(), name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ['hipstr_all', 'hipstr_uselength_drop', 'gangstr_all'])
def test_device_gpu(tmp_path, name):
    from trtools_amd.compute import DeviceCompute
    _run(tmp_path, DeviceCompute(), name)
