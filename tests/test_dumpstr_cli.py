"""dumpSTR mirror (trtools_amd.dumpSTR) end to end against the reference's golden
``.samplog.tab`` / ``.loclog.tab`` files, compared byte for byte like the reference's
own tests do (dumpSTR/tests/test_dumpSTR.py:720-918, testsupport/utils.py:202-219).

CPU: host layer + oracle-backed compute stand-in on the smaller inputs.
GPU: all eight golden runs through libtrk."""
import argparse
import os

import pytest

from helpers import GOLDEN

D = os.path.join(GOLDEN, 'data', 'dumpSTR')

ARG_NAMES = """min_locus_callrate min_locus_hwep min_locus_het max_locus_het filter_regions filter_regions_names
hipstr_min_call_DP hipstr_max_call_DP hipstr_min_call_Q hipstr_max_call_flank_indel hipstr_max_call_stutter
hipstr_min_supp_reads longtr_min_call_DP longtr_max_call_DP longtr_min_call_Q longtr_max_call_flank_indel
longtr_min_supp_reads gangstr_expansion_prob_het gangstr_expansion_prob_hom gangstr_expansion_prob_total
gangstr_filter_badCI gangstr_min_call_DP gangstr_max_call_DP gangstr_min_call_Q advntr_min_call_DP
advntr_max_call_DP advntr_min_spanning advntr_min_flanking advntr_min_ML eh_min_ADFL eh_min_ADIR eh_min_ADSP
eh_min_call_LC eh_max_call_LC popstr_min_call_DP popstr_max_call_DP popstr_require_support num_records""".split()

LOCUS = dict(min_locus_callrate=0.5, min_locus_hwep=0.5, min_locus_het=0.05, max_locus_het=0.45,
             filter_regions_names='foo_region', filter_regions=os.path.join(D, 'sample_region.bed.gz'),
             vcftype='hipstr')
CASES = {
    'locus_filters': ('trio_chr21_hipstr.sorted.vcf.gz', 'locus_filters', LOCUS),
    'drop_filtered': ('trio_chr21_hipstr.sorted.vcf.gz', 'locus_filters', dict(LOCUS, drop_filtered=True)),
    'advntr_filters': ('NA12878_chr21_advntr.sorted.vcf.gz', 'advntr_filters',
                       dict(advntr_min_call_DP=50, advntr_max_call_DP=2000, advntr_min_spanning=1,
                            advntr_min_flanking=20, advntr_min_ML=0.95)),
    'hipstr_filters': ('trio_chr21_hipstr.sorted.vcf.gz', 'hipstr_filters',
                       dict(filter_hrun=True, use_length=True, max_locus_het=0.45, min_locus_het=0.05,
                            min_locus_hwep=0.5, hipstr_max_call_flank_indel=0.05, hipstr_max_call_stutter=0.3,
                            hipstr_min_supp_reads=10, hipstr_min_call_DP=30, hipstr_max_call_DP=200,
                            hipstr_min_call_Q=0.9, vcftype='hipstr')),
    'longtr_filters': ('longtr_testfile.vcf.gz', 'longtr_filters',
                       dict(filter_hrun=True, use_length=True, max_locus_het=0.45, min_locus_het=0.05,
                            min_locus_hwep=0.5, longtr_max_call_flank_indel=0.05, longtr_min_supp_reads=10,
                            longtr_min_call_DP=30, longtr_max_call_DP=200, longtr_min_call_Q=0.9,
                            vcftype='longtr')),
    'gangstr_filters_most': ('trio_chr21_gangstr.sorted.vcf.gz', 'gangstr_filters_most',
                             dict(gangstr_min_call_DP=10, gangstr_max_call_DP=100, gangstr_min_call_Q=0.9,
                                  gangstr_filter_span_only=True, gangstr_filter_spanbound_only=True,
                                  gangstr_filter_badCI=True)),
    'gangstr_filters_expansion': ('test_gangstr.vcf.gz', 'gangstr_filters_expansion',
                                  dict(gangstr_expansion_prob_het=0.001, gangstr_expansion_prob_hom=0.0005,
                                       gangstr_expansion_prob_total=0.001)),
    'popstr_filters': ('NA12878_chr21_popstr.sorted.vcf.gz', 'popstr_filters',
                       dict(popstr_min_call_DP=30, popstr_max_call_DP=200, popstr_require_support=15,
                            use_length=True)),
}


def make_args(out, vcf, **kw):
    ns = argparse.Namespace(vcf=vcf, out=out, zip=False, vcftype='auto', use_length=False, filter_hrun=False,
                            drop_filtered=False, gangstr_filter_span_only=False,
                            gangstr_filter_spanbound_only=False, die_on_warning=False, verbose=False)
    for n in ARG_NAMES:
        setattr(ns, n, None)
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


def run_case(tmp_path, compute, name):
    from trtools_amd import runtime
    from trtools_amd.dumpSTR import dumpSTR
    vcf, gold, kw = CASES[name]
    out = str(tmp_path / 'test')
    old = runtime.set_compute(compute)
    try:
        assert dumpSTR.main(make_args(out, os.path.join(D, vcf), **kw)) == 0
    finally:
        runtime.set_compute(old)
    for ext in ('.samplog.tab', '.loclog.tab'):
        got = open(out + ext).read()
        want = open(os.path.join(D, gold + ext)).read()
        assert got == want, "%s%s differs:\n--- got\n%s\n--- want\n%s" % (name, ext, got[:1500], want[:1500])
    return out


@pytest.mark.parametrize("name", ['longtr_filters', 'advntr_filters', 'gangstr_filters_expansion'])
def test_host_layer_cpu(tmp_path, name):
    from oracle_compute import OracleCompute
    run_case(tmp_path, OracleCompute(), name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_logs_gpu(tmp_path, name):
    from trtools_amd.compute import DeviceCompute
    global _DC
    try:
        _DC
    except NameError:
        _DC = DeviceCompute()
    run_case(tmp_path, _DC, name)


def test_filter_argument_checks(tmp_path):
    """CheckFilters messages / return codes (dumpSTR.py:396-521)."""
    from trtools_amd.dumpSTR import dumpSTR
    vcf = os.path.join(D, 'longtr_testfile.vcf.gz')
    out = str(tmp_path / 'o')
    assert dumpSTR.main(make_args(out, vcf, vcftype='longtr', min_locus_hwep=2)) == 1
    assert dumpSTR.main(make_args(out, vcf, vcftype='longtr', min_locus_het=0.5, max_locus_het=0.2)) == 1
    assert dumpSTR.main(make_args(out, vcf, vcftype='longtr', hipstr_min_call_DP=10)) == 1   # wrong caller
    assert dumpSTR.main(make_args(out, vcf, vcftype='longtr', longtr_min_call_DP=-1)) == 1
    assert dumpSTR.main(make_args(out, vcf, vcftype='longtr', longtr_min_call_DP=10, longtr_max_call_DP=5)) == 1
    assert dumpSTR.main(make_args(str(tmp_path / 'x.'), vcf, vcftype='longtr')) == 1
    assert dumpSTR.main(make_args(out, str(tmp_path / 'nope.vcf'))) == 1


# ---- output VCF against the reference's golden VCFs (written by the real cyvcf2/htslib) ----
# same tolerance rules as the reference's assert_same_vcf (tests/vcf_compare.py); the three
# golden VCFs that are missing from the reference checkout (hipstr_filters, gangstr_filters_most,
# locus_filters) cannot be compared.
VCF_GOLD = {
    'longtr_filters': ('longtr_filters.vcf', {'GLDIFF'}),
    'advntr_filters': ('advntr_filters.vcf.gz', set()),
    'drop_filtered': ('drop_filtered.vcf.gz', {'GLDIFF'}),
    'gangstr_filters_expansion': ('gangstr_filters_expansion.vcf.gz', set()),
    'popstr_filters': ('popstr_filters.vcf.gz', set()),
}


def _check_vcf(out, name):
    from vcf_compare import compare_vcfs
    gold, fmt_ignore = VCF_GOLD[name]
    problems = compare_vcfs(out + '.vcf', os.path.join(D, gold), info_ignore={'AC', 'REFAC', 'HET', 'HWEP'},
                            format_ignore=fmt_ignore)
    assert not problems, "\n".join(problems[:20])


def test_output_vcf_matches_reference_golden_cpu(tmp_path):
    from oracle_compute import OracleCompute
    _check_vcf(run_case(tmp_path, OracleCompute(), 'longtr_filters'), 'longtr_filters')


def test_popstr_require_support_runs_through_the_batch_pipeline_cpu(tmp_path):
    """--popstr-require-support (filters.py:835-867; AD is Number=R) no longer sends a PopSTR run to the per-record
    loop: AD is decoded into a fixed number of columns, the native writer computes the reported read support
    (trk_vcf_cf_value kind 4).  Logs and VCF equal the reference's goldens; with a narrower AD plane than the
    records' allele lists the batches fall back to the per-record loop and the outputs stay the same."""
    from oracle_compute import OracleCompute
    from trtools_amd.dumpSTR import dumpSTR
    _check_vcf(run_case(tmp_path, OracleCompute(), 'popstr_filters'), 'popstr_filters')
    assert dumpSTR.LAST_RUN['path'] == 'batch' and dumpSTR.LAST_RUN['fallback_batches'] == 0
    old = dumpSTR._Run.AD_COLUMNS
    dumpSTR._Run.AD_COLUMNS = 2
    try:
        sub = tmp_path / 'narrow'
        sub.mkdir()
        _check_vcf(run_case(sub, OracleCompute(), 'popstr_filters'), 'popstr_filters')
        assert dumpSTR.LAST_RUN['fallback_batches'] > 0
    finally:
        dumpSTR._Run.AD_COLUMNS = old


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(VCF_GOLD))
def test_output_vcf_matches_reference_golden_gpu(tmp_path, name):
    from trtools_amd.compute import DeviceCompute
    _check_vcf(run_case(tmp_path, DeviceCompute(), name), name)
