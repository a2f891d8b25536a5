"""associaTR command line (trtools_amd.associaTR.associaTR.main) against the tables the REAL reference
wrote for the same arguments (tests/golden/associatr, tools/gen_golden_associatr.py) and against the
reference's plink2 fixtures under its own acceptance rule.

CPU: the host layer (covariate joins, harmonisation, packing, text) with the oracle-backed compute
stand-in.  GPU: the same with the device (trk_assoc_scan through the C ABI)."""
import contextlib
import io
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import assoc_cases                                            # noqa: E402
from assoc_compare import compare_tables, compare_to_plink    # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden', 'associatr')
GT_CASES = sorted(assoc_cases.CASES)     # GT-based and --beagle-dosages runs alike


def run_cli(out, kw, len_precision, p_precision):
    from trtools_amd.associaTR import associaTR as at
    old = (at.load_and_filter_genotypes.allele_len_precision, at.pval_precision)
    at.load_and_filter_genotypes.allele_len_precision, at.pval_precision = len_precision, p_precision
    try:
        with contextlib.redirect_stdout(io.StringIO()) as log:
            at.main(assoc_cases.make_args(out, **kw))
    finally:
        at.load_and_filter_genotypes.allele_len_precision, at.pval_precision = old
    return log.getvalue()


def check_case(name, tmp_path):
    kw, plink, skip = assoc_cases.CASES[name]
    out = str(tmp_path / 'a.tsv')
    log = run_cli(out, kw, 10, 15)
    assert 'samples in the VCF' in log and 'Done.' in log
    from trtools_amd.associaTR import associaTR as at
    # (round 6) GT-based runs go through the batch pipeline -- native reader, native batch harmoniser, one scan per batch,
    # rows without an object per record; --beagle-dosages keeps the per-record loop
    assert at.LAST_RUN['path'] == ('per-record' if kw.get('beagle_dosages') else 'batch'), at.LAST_RUN
    compare_tables(out, os.path.join(GOLD, name + '.precise.tsv'), rtol=1e-9)
    if plink:
        assert compare_to_plink(out, os.path.join(assoc_cases.DATA, plink), 'test_pheno', skip_filtered=skip) > 100
    run_cli(out, kw, 2, 2)
    compare_tables(out, os.path.join(GOLD, name + '.tsv'), rtol=1e-9, p_rtol=0.0)
    assert not os.path.exists(out + '.temp')


@pytest.fixture
def oracle_compute():
    from trtools_amd import runtime
    from oracle_compute import OracleCompute
    old = runtime.set_compute(OracleCompute())
    yield
    runtime.set_compute(old)


@pytest.mark.parametrize('name', GT_CASES)
def test_cli_host_layer_with_oracle_compute(name, tmp_path, oracle_compute):
    check_case(name, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize('name', GT_CASES)
def test_cli_on_device(name, tmp_path):
    from trtools_amd import runtime
    runtime.set_compute(None)
    check_case(name, tmp_path)


@pytest.mark.parametrize('name', ['hipstr_covars', 'multiallelic', 'region', 'sample_subset'])
def test_per_record_loop_writes_the_batch_pipelines_table(name, tmp_path, oracle_compute):
    """The two ways through perform_gwas -- record objects one by one (TRK_ASSOC_BATCH=0) and the batch pipeline -- write the
    same bytes."""
    from helpers import lab_env
    from trtools_amd.associaTR import associaTR as at
    kw, _, _ = assoc_cases.CASES[name]
    a, b = str(tmp_path / 'a.tsv'), str(tmp_path / 'b.tsv')
    run_cli(a, kw, 10, 15)
    assert at.LAST_RUN['path'] == 'batch'
    with lab_env(TRK_ASSOC_BATCH='0'):
        run_cli(b, kw, 10, 15)
    assert at.LAST_RUN['path'] == 'per-record'
    assert open(a).read() == open(b).read() and open(a).read().count('\n') > 1


def test_refused_options(tmp_path, oracle_compute):
    from trtools_amd.associaTR import associaTR as at
    with pytest.raises(NotImplementedError):
        at.main(assoc_cases.make_args(str(tmp_path / 'x.tsv'), same_samples=True, plotting_phenotype='p.npy'))


def test_region_is_a_slice_of_the_full_run(tmp_path, oracle_compute):
    """associaTR/tests/test_associaTR.py:113-133."""
    full, part, none = (str(tmp_path / n) for n in ('f.tsv', 'p.tsv', 'n.tsv'))
    run_cli(full, dict(same_samples=True), 2, 2)
    run_cli(part, dict(same_samples=True, region='1:993134-3781638'), 2, 2)
    run_cli(none, dict(same_samples=True, region='2:993134-3781638'), 2, 2)
    lines = open(full).readlines()
    got = open(part).readlines()
    assert got[0] == lines[0] and got[1:] == lines[77:77 + len(got) - 1] and len(got) - 1 == 366 - 77 + 1
    assert len(open(none).readlines()) == 1


def _wide_traits(tmp_path, n_cols, seed=9):
    """Seeded trait file for the 50-sample HipSTR fixture: sample index, outcome, n_cols - 1 covariates."""
    import numpy as np
    rng = np.random.default_rng(seed)
    t = rng.normal(size=(50, n_cols))
    f = str(tmp_path / ('traits_%d.npy' % n_cols))
    np.save(f, t)
    return f


def test_wide_design_limits_of_the_host_layer(tmp_path, oracle_compute):
    """More than 31 trait columns (no bound in the reference, associaTR.py:138-204): accepted up to 126 for the
    GT-based scan and for --beagle-dosages, refused above."""
    out = str(tmp_path / 'w.tsv')
    run_cli(out, dict(same_samples=True, tr_vcf=assoc_cases.HIPSTR, traits=[_wide_traits(tmp_path, 40)]), 2, 2)
    rows = open(out).readlines()
    assert len(rows) > 10 and sum('n covars >= n samples' not in r for r in rows[1:]) > 5
    with pytest.raises(ValueError):
        run_cli(out, dict(same_samples=True, tr_vcf=assoc_cases.HIPSTR, traits=[_wide_traits(tmp_path, 127)]), 2, 2)
    with pytest.raises(ValueError):
        run_cli(out, dict(same_samples=True, beagle_dosages=True, traits=[_wide_traits(tmp_path, 127)]), 2, 2)


@pytest.mark.gpu
@pytest.mark.parametrize('n_cols', [32, 36, 40])
def test_wide_design_cli_on_device_equals_oracle_compute(n_cols, tmp_path):
    """The whole CLI with 32-40 trait columns on the device against the same run through the oracle-backed seam."""
    from trtools_amd import runtime
    from oracle_compute import OracleCompute
    kw = dict(same_samples=True, tr_vcf=assoc_cases.HIPSTR, traits=[_wide_traits(tmp_path, n_cols)])
    dev, ora = str(tmp_path / 'd.tsv'), str(tmp_path / 'o.tsv')
    runtime.set_compute(None)
    run_cli(dev, kw, 10, 15)
    old = runtime.set_compute(OracleCompute())
    try:
        run_cli(ora, kw, 10, 15)
    finally:
        runtime.set_compute(old)
    assert compare_tables(dev, ora, rtol=1e-7) > 50
