#!/usr/bin/env python3
"""TRRecord.GetDosages goldens produced by RUNNING THE REFERENCE here (build container only).

    python tools/gen_golden_dosages.py        # rewrites tests/golden/dosages.npz

trtools/utils/tr_harmonizer.py:1098-1208 is run on every record of the
Beagle-annotated fixtures (all four dosage types) and on the first records of the HipSTR fixture
(best-guess types: missing calls, many alleles, fractional lengths).  Only arrays are written.
"""
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, 'refshim'))
sys.path.insert(0, REPO)
sys.path.insert(0, '/root/reference')

import numpy as np  # noqa: E402

DATA = os.path.join(REPO, 'tests', 'golden', 'data')
FILES = [('associaTR/many_samples_biallelic_dosages.vcf.gz', 'hipstr', True, 10 ** 9),
         ('associaTR/many_samples_multiallelic_dosages.vcf.gz', 'hipstr', True, 10 ** 9),
         ('many_samples.vcf.gz', 'hipstr', False, 300)]
TYPES = ('bestguess', 'beagleap', 'bestguess_norm', 'beagleap_norm')


def main():
    import trtools.utils.tr_harmonizer as trh       # the reference
    import trtools.utils.utils as rutils
    out = {}
    for rel, vt, has_ap, limit in FILES:
        reader = rutils.LoadSingleReader(os.path.join(DATA, rel), checkgz=False)
        key = rel.replace('/', '__')
        for t in TYPES:
            if not has_ap and t.startswith('beagle'):
                continue
            rows = []
            reader2 = rutils.LoadSingleReader(os.path.join(DATA, rel), checkgz=False)
            for i, rec in enumerate(trh.TRRecordHarmonizer(reader2, vt)):
                if i >= limit:
                    break
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    rows.append(np.asarray(rec.GetDosages(trh.TRDosageTypes[t], strict=False), dtype=np.float32))
            out['%s::%s' % (key, t)] = np.stack(rows)
            print(key, t, out['%s::%s' % (key, t)].shape)
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'dosages.npz'), **out)


if __name__ == '__main__':
    main()
