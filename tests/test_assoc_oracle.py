"""The associaTR oracle (oracle/associatr_oracle.py) against (i) tables written by the real
reference in the build container (tools/gen_golden_associatr.py) and (ii) the reference's plink2
fixtures under the reference tests' own acceptance rule."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from oracle import associatr_oracle as ao          # noqa: E402
from assoc_compare import compare_tables, compare_to_plink   # noqa: E402
import assoc_cases                                  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden', 'associatr')


def run_oracle(out, kw, precision, pval_precision):
    """associaTR.main restated on top of the oracle; VCF decoding / harmonisation by the host layer."""
    from trtools_amd import vcfio
    from trtools_amd.utils import tr_harmonizer as trh
    a = assoc_cases.make_args(out, **kw)
    reader = vcfio.VCFReader(a.tr_vcf)
    samples = reader.samples
    traits = [np.load(t) for t in a.traits]
    subset = None
    if a.sample_list:
        subset = [line.strip() for line in open(a.sample_list)]
    sf, covars, outcome, pheno_std = ao.prepare_design(samples, traits, a.same_samples, subset)
    vcftype = trh.InferVCFType(reader, a.vcftype if a.vcftype else 'auto')
    it = reader
    region_start = None
    if a.region:
        region_start = int(a.region.split(':')[1].split('-')[0])
        it = reader(a.region)
    with open(out, 'w') as fh:
        fh.write(ao.header(a.phenotype_name, a.beagle_dosages))
        for rec in it:
            if region_start is not None and rec.POS < region_start:
                continue
            tr = trh.HarmonizeRecord(vcftype, rec)
            gt = rec.genotype.array()[:, :-1]
            lens = [tr.ref_allele_length] + list(tr.alt_allele_lengths)
            ap1 = ap2 = None
            if a.beagle_dosages:
                ap1, ap2 = rec.format('AP1'), rec.format('AP2')
            res = ao.scan_locus(gt, lens, sf, covars, outcome, pheno_std, a.non_major_cutoff, precision, ap1, ap2)
            fh.write(ao.format_row(tr.chrom, tr.pos, res, tr.motif, tr.ref_allele_length, pval_precision, precision))


@pytest.mark.parametrize('name', sorted(assoc_cases.CASES))
def test_oracle_reproduces_reference_tables(name, tmp_path):
    kw, plink, skip = assoc_cases.CASES[name]
    out = str(tmp_path / 'o.tsv')
    run_oracle(out, kw, 10, 15)
    n = compare_tables(out, os.path.join(GOLD, name + '.precise.tsv'), rtol=1e-9)
    assert n > 0 or 'cutoff' in name
    if plink:
        assert compare_to_plink(out, os.path.join(assoc_cases.DATA, plink), 'test_pheno', skip_filtered=skip) > 100
    # default text precision: byte for byte except the three full-repr float columns
    run_oracle(out, kw, 2, 2)
    compare_tables(out, os.path.join(GOLD, name + '.tsv'), rtol=1e-9, p_rtol=0.0)


def test_plink_agreement_is_tight():
    """The fixtures carry six significant digits; P, BETA and SE agree to ~1e-5, far inside the 2 % rule."""
    from assoc_compare import read_table
    h, rows = read_table(os.path.join(GOLD, 'one_trait_file.precise.tsv'))
    ph, prows = read_table(os.path.join(assoc_cases.DATA, 'single.plink2.trait_0.glm.linear'))
    worst = 0.0
    for r, p in zip(rows, prows):
        if r[4] != 'False':
            continue
        worst = max(worst, abs(float(r[5]) / float(p[ph.index('P')]) - 1))
    assert worst < 2e-5, worst
