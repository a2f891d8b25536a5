"""statSTR mirror (trtools_amd.statSTR) end to end against the reference's golden
tables (sample_stats/many_samples_all*.tab; arguments of the reference's
statSTR/tests/test_statSTR.py:253-309).

CPU run: host layer with the oracle-backed compute stand-in (tests/oracle_compute.py).
GPU run: the real libtrk path; the table must be byte-identical to the golden file."""
import argparse
import os

import pytest

from helpers import GOLDEN

DATA = os.path.join(GOLDEN, 'data')


def _args(out, **kw):
    ns = argparse.Namespace(
        vcf=os.path.join(DATA, 'many_samples.vcf.gz'), out=out, vcftype='hipstr', samples=None,
        sample_prefixes=None, plot_afreq=False, region=None, thresh=True, afreq=True, acount=True, hwep=True,
        het=True, entropy=True, mean=True, mode=True, var=True, numcalled=True, use_length=False, precision=4,
        nalleles=True, nalleles_thresh=0.1, only_passing=False)
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


def _run(tmp_path, compute, golden, **kw):
    from trtools_amd import runtime
    from trtools_amd.statSTR import statSTR
    old = runtime.set_compute(compute)
    try:
        out = str(tmp_path / 'out')
        assert statSTR.main(_args(out, **kw)) == 0
    finally:
        runtime.set_compute(old)
    got = open(out + '.tab').read().split('\n')
    want = open(os.path.join(DATA, golden)).read().split('\n')
    diffs = [(i, g, w) for i, (g, w) in enumerate(zip(got, want)) if g != w]
    msg = ''
    if diffs:
        i, g, w = diffs[0]
        gc, wc = g.split('\t'), w.split('\t')
        cols = [(j, a, b) for j, (a, b) in enumerate(zip(gc, wc)) if a != b]
        msg = "%d differing lines; first at line %d, columns %s" % (len(diffs), i, cols[:4])
    assert not diffs and len(got) == len(want), msg


def _strat():
    return dict(samples=os.path.join(DATA, 'many_samples_subsample1.txt') + ',' +
                os.path.join(DATA, 'many_samples_subsample2.txt'))


def test_host_layer_all_stats_cpu(tmp_path):
    from oracle_compute import OracleCompute
    _run(tmp_path, OracleCompute(), 'many_samples_all.tab')


def test_host_layer_stratified_cpu(tmp_path):
    from oracle_compute import OracleCompute
    _run(tmp_path, OracleCompute(), 'many_samples_all_strat.tab', **_strat())


@pytest.mark.gpu
def test_golden_all_stats_gpu(tmp_path):
    from trtools_amd.compute import DeviceCompute
    _run(tmp_path, DeviceCompute(), 'many_samples_all.tab')


@pytest.mark.gpu
def test_golden_stratified_gpu(tmp_path):
    """The reference's stratified table three ways: sample columns parsed on the device and counted by the grouped kernel
    in file order (round 6, the default), and the host parse with / without the class-ordered columns."""
    from helpers import lab_env
    from trtools_amd.compute import DeviceCompute
    from trtools_amd.statSTR import statSTR
    _run(tmp_path, DeviceCompute(), 'many_samples_all_strat.tab', **_strat())
    assert statSTR.LAST_RUN['device_parse'] and statSTR.LAST_RUN['path'] == 'batch'
    for sort in ('1', '0'):
        with lab_env(TRK_GROUPS_DEVICE_PARSE='0', TRK_CLASS_SORT=sort):
            _run(tmp_path, DeviceCompute(), 'many_samples_all_strat.tab', **_strat())
            assert not statSTR.LAST_RUN['device_parse']


def test_bad_inputs_return_1(tmp_path):
    from trtools_amd.statSTR import statSTR
    assert statSTR.main(_args(str(tmp_path / 'o'), vcf=str(tmp_path / 'missing.vcf'))) == 1
    assert statSTR.main(_args(str(tmp_path / 'nodir' / 'o'))) == 1
    # a region query needs a bgzipped + indexed file (statSTR.py:511-514)
    import shutil
    noidx = str(tmp_path / 'noindex.vcf.gz')
    shutil.copy(os.path.join(DATA, 'many_samples.vcf.gz'), noidx)
    assert statSTR.main(_args(str(tmp_path / 'o'), vcf=noidx, region='chr1:1-10')) == 1
