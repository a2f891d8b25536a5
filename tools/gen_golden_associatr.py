#!/usr/bin/env python3
"""associaTR goldens produced by RUNNING THE REFERENCE here (build container only).

    python tools/gen_golden_associatr.py        # rewrites tests/golden/associatr/*

/root/reference/trtools/associaTR/associaTR.py is imported and its ``main`` run on the argument
sets of its own test-suite (associaTR/tests/test_associaTR.py:17-160) plus a HipSTR multi-allelic
case with missing calls.  cyvcf2 and statsmodels are absent from this image: tools/refshim
provides the VCF decoder (this repo's) and an OLS stand-in that restates statsmodels'
pinv fit; every biallelic case is checked against the reference's plink2 fixtures with the
reference test's own comparator rules (2 % on the third significant digit) before it is written,
which is what pins the stand-in.  Only data is written (the reference's output tables).

Each case is written twice: ``<case>.tsv`` with the reference's default text precision and
``<case>.precise.tsv`` with allele_len_precision=10 / pval_precision=15 (the reference tests
raise the precisions in the same way).
"""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, 'refshim'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import numpy as np  # noqa: E402

from assoc_cases import CASES, DATA, OUT, HIPSTR, make_args  # noqa: E402


def write_hipstr_inputs():
    """Seeded traits for the HipSTR fixture (phenotype + 3 covariates, two nan rows) and a sample subset."""
    from trtools_amd import vcfio
    samples = vcfio.VCFReader(HIPSTR).samples
    rng = np.random.default_rng(20260928)
    n = len(samples)
    tr = rng.normal(size=(n, 4))
    tr[:, 0] += 0.3 * tr[:, 1]
    tr[5, 2] = np.nan
    tr[17, 0] = np.nan
    os.makedirs(OUT, exist_ok=True)
    np.save(os.path.join(OUT, 'hipstr_traits.npy'), tr)
    keep = [s for i, s in enumerate(samples) if i % 5 != 3] + ['not_in_vcf']
    with open(os.path.join(OUT, 'hipstr_samples.txt'), 'w') as fh:
        fh.write('\n'.join(keep) + '\n')


def main():
    sys.path.insert(0, '/root/reference')
    import trtools.associaTR.associaTR as rassoc     # the reference
    from assoc_compare import compare_to_plink
    write_hipstr_inputs()
    lf = rassoc.load_and_filter_genotypes
    for name, (kw, plink, skip) in CASES.items():
        for tag, (alp, pvp) in (('', (2, 2)), ('.precise', (10, 15))):
            lf.allele_len_precision, rassoc.pval_precision = alp, pvp
            out = os.path.join(OUT, name + tag + '.tsv')
            with contextlib.redirect_stdout(io.StringIO()):
                rassoc.main(make_args(out, **kw))
            if plink and tag:
                compare_to_plink(out, os.path.join(DATA, plink), 'test_pheno', skip_filtered=skip)
        print('associatr/%s: ok%s' % (name, ' (plink-checked)' if plink else ''))


if __name__ == '__main__':
    main()
