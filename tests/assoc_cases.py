"""Argument sets of the associaTR golden runs (shared by tools/gen_golden_associatr.py and the tests).
They mirror the reference's own test-suite (associaTR/tests/test_associaTR.py:17-160) plus two runs on
the multi-allelic HipSTR fixture with seeded traits."""
import argparse
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(REPO, 'tests', 'golden', 'data', 'associaTR')
OUT = os.path.join(REPO, 'tests', 'golden', 'associatr')
BI = os.path.join(DATA, 'many_samples_biallelic_dosages.vcf.gz')
MULTI = os.path.join(DATA, 'many_samples_multiallelic_dosages.vcf.gz')
HIPSTR = os.path.join(REPO, 'tests', 'golden', 'data', 'many_samples.vcf.gz')


def T(name):
    return os.path.join(DATA, name)


# name -> (argument overrides, plink fixture or None, skip_filtered)
CASES = {
    'one_trait_file': (dict(same_samples=True), 'single.plink2.trait_0.glm.linear', False),
    'two_trait_files': (dict(same_samples=True, traits=[T('traits_0.npy'), T('traits_1.npy')]),
                        'combined.plink2.trait_0.glm.linear', False),
    'sample_merge': (dict(traits=[T('traits_0_40_samples.npy')]), 'single_40.plink2.trait_0.glm.linear', False),
    'two_files_sample_merge': (dict(traits=[T('traits_0_40_samples.npy'), T('traits_1_45_samples.npy')]),
                               'combined_35.plink2.trait_0.glm.linear', False),
    'sample_subset': (dict(same_samples=True, sample_list=T('samples_6_to_45.txt')),
                      'single_40.plink2.trait_0.glm.linear', False),
    'merge_and_subset': (dict(traits=[T('traits_0_40_samples.npy')], sample_list=T('45_samples.txt')),
                         'single_35.plink2.trait_0.glm.linear', False),
    'region': (dict(same_samples=True, region='1:993134-3781638'), None, False),
    'cutoff_5': (dict(same_samples=True, non_major_cutoff=5), 'single_cutoff_5.plink2.trait_0.glm.linear', True),
    'cutoff_default': (dict(same_samples=True, non_major_cutoff=20), None, False),
    'dosages': (dict(same_samples=True, beagle_dosages=True), 'single_dosages.plink2.trait_0.glm.linear', False),
    'dosage_sample_subset': (dict(same_samples=True, beagle_dosages=True, sample_list=T('samples_6_to_45.txt')),
                             'single_40_dosages.plink2.trait_0.glm.linear', False),
    'multiallelic': (dict(same_samples=True, tr_vcf=MULTI), None, False),
    'multiallelic_dosages': (dict(same_samples=True, tr_vcf=MULTI, beagle_dosages=True), None, False),
    'multiallelic_cutoff_8': (dict(same_samples=True, tr_vcf=MULTI, non_major_cutoff=8), None, False),
    'multiallelic_dosage_cutoff_20': (dict(same_samples=True, tr_vcf=MULTI, beagle_dosages=True, non_major_cutoff=20),
                                      None, False),
    # HipSTR call set with missing calls, many alleles, fractional lengths; seeded traits (3 covariates)
    'hipstr_covars': (dict(same_samples=True, tr_vcf=HIPSTR, traits=[os.path.join(OUT, 'hipstr_traits.npy')],
                           non_major_cutoff=3), None, False),
    'hipstr_subset': (dict(same_samples=True, tr_vcf=HIPSTR, traits=[os.path.join(OUT, 'hipstr_traits.npy')],
                           sample_list=os.path.join(OUT, 'hipstr_samples.txt'), non_major_cutoff=0), None, False),
}


def make_args(outfile, **kw):
    ns = argparse.Namespace(outfile=outfile, tr_vcf=BI, phenotype_name='test_pheno', traits=[T('traits_0.npy')],
                            vcftype='auto', same_samples=False, sample_list=None, region=None, non_major_cutoff=0,
                            beagle_dosages=False, plotting_phenotype=None, paired_genotype_plot=False,
                            plot_phenotype_residuals=False, plotting_ci_alphas=[],
                            imputed_ukb_strs_paper_period_check=False)
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


