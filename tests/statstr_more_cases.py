"""Further statSTR argument sets (the reference's own tests, trtools/statSTR/tests/test_statSTR.py:47-110, and every
caller's fixture VCF with every statistic switched on) -- shared by tools/gen_golden_statstr_more.py, which records
what the REAL reference writes for them, and tests/test_statstr_more.py."""
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(REPO, 'tests', 'golden', 'data')
S = os.path.join(DATA, 'statSTR')
D = os.path.join(DATA, 'dumpSTR')
OUT = os.path.join(REPO, 'tests', 'golden', 'statstr_more')
ALL = dict(thresh=True, afreq=True, acount=True, hwep=True, het=True, entropy=True, mean=True, mode=True, var=True,
           numcalled=True, nalleles=True)
NONE = {k: False for k in ALL}

GROUPS = os.path.join(DATA, 'statSTR', 'groups')     # ten sample lists of many_samples.vcf.gz (five samples each)
TEN = ','.join(os.path.join(GROUPS, 'g%d.txt' % i) for i in range(10))

CASES = [
    # more sample groups than one kernel pass takes (8): two passes over the batch
    ('ten_groups', os.path.join(DATA, 'many_samples.vcf.gz'), 'hipstr',
     dict(ALL, samples=TEN, sample_prefixes=','.join('grp%d' % i for i in range(10)), region='1:3000000-6000000')),
    ('ten_groups_uselength', os.path.join(DATA, 'many_samples.vcf.gz'), 'hipstr',
     dict(ALL, use_length=True, samples=TEN, sample_prefixes=','.join('grp%d' % i for i in range(10)), region='1:1-3000000')),
    ('ceu_only_passing', os.path.join(S, 'CEU_test.vcf.gz'), 'auto', dict(NONE, only_passing=True)),
    ('ceu_all', os.path.join(S, 'CEU_test.vcf.gz'), 'auto', dict(ALL)),
    ('few_all', os.path.join(S, 'few_samples_few_loci.vcf.gz'), 'auto', dict(ALL)),
    ('few_uselength', os.path.join(S, 'few_samples_few_loci.vcf.gz'), 'auto', dict(ALL, use_length=True)),
    ('few_region', os.path.join(S, 'few_samples_few_loci.vcf.gz'), 'auto', dict(ALL, region='chr1:3045469-3045470')),
    ('few_samples', os.path.join(S, 'few_samples_few_loci.vcf.gz'), 'auto',
     dict(ALL, samples=os.path.join(S, 'fewer_samples.txt'))),
    ('few_missing_samples', os.path.join(S, 'few_samples_few_loci.vcf.gz'), 'auto',
     dict(ALL, samples=os.path.join(S, 'missing_samples.txt'))),
    ('region_needs_tabix', os.path.join(S, 'test_ExpansionHunter.vcf'), 'auto', dict(NONE, thresh=True, region='chr1:3045469-3045470')),
    ('multiple_chroms', os.path.join(S, 'many_samples_multiple_chroms.vcf.gz'), 'auto', dict(ALL)),
    ('multiple_chroms_region', os.path.join(S, 'many_samples_multiple_chroms.vcf.gz'), 'auto', dict(ALL, region='2')),
    ('eh_small', os.path.join(S, 'test_ExpansionHunter.vcf'), 'auto', dict(ALL)),
    ('advntr_small', os.path.join(S, 'test_advntr.vcf'), 'auto', dict(ALL)),
    ('popstr_small', os.path.join(S, 'test_popstr.vcf'), 'auto', dict(ALL)),
    ('longtr_small', os.path.join(S, 'test_longtr.vcf'), 'longtr', dict(ALL)),
    ('longtr_small_uselength', os.path.join(S, 'test_longtr.vcf'), 'longtr', dict(ALL, use_length=True)),
    ('trio_hipstr', os.path.join(D, 'trio_chr21_hipstr.sorted.vcf.gz'), 'hipstr', dict(ALL, precision=6)),
    ('trio_hipstr_uselength', os.path.join(D, 'trio_chr21_hipstr.sorted.vcf.gz'), 'hipstr', dict(ALL, use_length=True)),
    ('trio_gangstr', os.path.join(D, 'trio_chr21_gangstr.sorted.vcf.gz'), 'gangstr', dict(ALL)),
    ('na12878_advntr', os.path.join(D, 'NA12878_chr21_advntr.sorted.vcf.gz'), 'auto', dict(ALL)),
    ('na12878_popstr', os.path.join(D, 'NA12878_chr21_popstr.sorted.vcf.gz'), 'auto', dict(ALL)),
    ('na12878_eh', os.path.join(D, 'NA12878_chr21_eh.sorted.vcf.gz'), 'auto', dict(ALL)),
    ('longtr_testfile', os.path.join(D, 'longtr_testfile.vcf.gz'), 'longtr', dict(ALL)),
    ('test_gangstr', os.path.join(D, 'test_gangstr.vcf.gz'), 'auto', dict(ALL, nalleles_thresh=0.05)),
]
