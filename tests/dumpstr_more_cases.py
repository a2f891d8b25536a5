"""Further dumpSTR argument sets taken from the reference's own test-suite
(trtools/dumpSTR/tests/test_dumpSTR.py: EH input, region filters, a second dumpSTR round, Beagle-imputed
inputs, pre-existing fields, zipped output, broken input) -- shared by tools/gen_golden_dumpstr_more.py,
which records what the REAL reference does with them, and tests/test_dumpstr_more.py."""
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(REPO, 'tests', 'golden', 'data')
D = os.path.join(DATA, 'dumpSTR')
REG = os.path.join(DATA, 'regions')
BGL = os.path.join(DATA, 'beagle')
OUT = os.path.join(REPO, 'tests', 'golden', 'dumpstr_more')


def r(name):
    return os.path.join(REG, name)


# name -> (vcf, overrides).  'vcf' may be '@<case>' = the output VCF of an earlier case.
CASES = [
    ('eh_file', os.path.join(D, 'NA12878_chr21_eh.sorted.vcf.gz'), dict(use_length=True, num_records=10)),
    ('regions_one', os.path.join(D, 'test_gangstr.vcf.gz'), dict(num_records=10, filter_regions=r('test_regions1.bed.gz'))),
    ('regions_named', os.path.join(D, 'test_gangstr.vcf.gz'),
     dict(num_records=10, filter_regions=r('test_regions1.bed.gz'), filter_regions_names='test')),
    ('regions_two', os.path.join(D, 'test_gangstr.vcf.gz'),
     dict(filter_regions=r('test_regions1.bed.gz') + ',' + r('test_regions2.bed.gz'), filter_regions_names='test1,test2')),
    ('regions_name_mismatch', os.path.join(D, 'test_gangstr.vcf.gz'),
     dict(num_records=10, filter_regions=r('test_regions1.bed.gz') + ',' + r('test_regions2.bed.gz'),
          filter_regions_names='test1')),
    ('regions_nonexistent', os.path.join(D, 'test_gangstr.vcf.gz'),
     dict(num_records=10, filter_regions=r('test_nonexistent.bed'), filter_regions_names='test1')),
    ('regions_no_tabix', os.path.join(D, 'test_gangstr.vcf.gz'),
     dict(num_records=10, filter_regions=r('test_regions3.bed.gz'), filter_regions_names='test1')),
    ('regions_nochr_bed', os.path.join(D, 'test_gangstr.vcf.gz'),
     dict(num_records=10, filter_regions=r('test_regions4.bed.gz'), filter_regions_names='test1')),
    ('regions_nochr_vcf', os.path.join(D, 'test_gangstr_nochr.vcf.gz'),
     dict(num_records=10, filter_regions=r('test_regions4.bed.gz'), filter_regions_names='test1')),
    # test_TwoDumpSTRRounds without --zip (the reference shells out to `tabix`, absent from this image)
    ('round_one', os.path.join(D, 'test_gangstr.vcf.gz'), dict(num_records=10, min_locus_callrate=0)),
    ('round_two', '@round_one', dict(num_records=10, min_locus_callrate=0)),
    ('gangstr_trio_all', os.path.join(D, 'trio_chr21_gangstr.sorted.vcf.gz'),
     dict(vcftype='gangstr', num_records=10, gangstr_min_call_DP=10, gangstr_max_call_DP=20, gangstr_min_call_Q=0.99,
          gangstr_filter_span_only=True, gangstr_filter_spanbound_only=True, gangstr_filter_badCI=True)),
    ('hipstr_hrun_regions', os.path.join(D, 'trio_chr21_hipstr.sorted.vcf.gz'),
     dict(vcftype='hipstr', num_records=400, filter_hrun=True, min_locus_callrate=0.7, min_locus_het=0.1,
          filter_regions=os.path.join(D, 'sample_region.bed.gz'), hipstr_min_call_DP=15)),
    ('broken_vcf', os.path.join(D, 'test_broken.vcf.gz'), dict(num_records=10, die_on_warning=True, verbose=True)),
    ('bad_pre_ac_refac', os.path.join(D, 'bad_preexisting_filter_ac_refac.vcf'), dict(num_records=10)),
    ('bad_pre_het_hwep', os.path.join(D, 'bad_preexisting_het_hwep.vcf'), dict(num_records=10)),
    ('bad_pre_hrun', os.path.join(D, 'bad_preexisting_hrun.vcf'), dict(num_records=10)),
    ('worrisome_pre_filter', os.path.join(D, 'worrisome_preexisting_filter.vcf'), dict(num_records=10)),
]
for _caller in ('advntr', 'eh', 'gangstr', 'hipstr'):
    CASES.append(('beagle_allowed_' + _caller, os.path.join(BGL, _caller + '_imputed.vcf.gz'),
                  dict(min_locus_hwep=0.1, min_locus_het=0.1, max_locus_het=0.9, filter_regions=r('test_regions1.bed.gz'))))
    CASES.append(('beagle_callrate_' + _caller, os.path.join(BGL, _caller + '_imputed.vcf.gz'), dict(min_locus_callrate=0.1)))
CASES.append(('beagle_hrun_hipstr', os.path.join(BGL, 'hipstr_imputed.vcf.gz'), dict(filter_hrun=True)))
for _k, _v in (('hipstr_min_call_DP', 5), ('hipstr_max_call_DP', 1000), ('hipstr_min_call_Q', 0.2),
               ('hipstr_max_call_flank_indel', 0.2), ('hipstr_max_call_stutter', 0.2), ('hipstr_min_supp_reads', 2)):
    CASES.append(('beagle_call_' + _k, os.path.join(BGL, 'hipstr_imputed.vcf.gz'), {_k: _v}))
