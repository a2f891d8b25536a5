#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING THE REAL REFERENCE (build container only).

Usage (from the repo root, in the build container where /root/reference exists):

    python tools/gen_golden.py            # rewrites tests/golden/*.json

The reference (gymrek-lab/TRTools, /root/reference) is pure Python but needs
the third-party ``cyvcf2`` / ``pysam`` wheels, which are absent from this
image; ``tools/refshim`` provides stand-ins built on this repo's own VCF
decoder.  Only *data* (inputs and the reference's outputs) is written to
``tests/golden``; no reference source is copied.

Vectors:
  trrecord_vectors.json   TRRecord reductions + statSTR column functions on
                          seeded random / edge-case genotype matrices
                          (reference: utils/tr_harmonizer.py:829-1575,
                          utils/utils.py:118-338, statSTR/statSTR.py:104-426).
  callfilter_vectors.json dumpSTR call filters + ApplyCallFilters +
                          ApplyLocusFilters on synthetic HipSTR / GangSTR /
                          PopSTR records (dumpSTR/filters.py, dumpSTR.py:613-973).
  binomtest_vectors.json  scipy.stats.binomtest known answers (the third-party
                          call at utils/utils.py:334-338).
"""
import argparse
import collections
import json
import math
import os
import sys
import types
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, 'refshim'))
sys.path.insert(0, REPO)
sys.path.insert(0, '/root/reference')

import numpy as np  # noqa: E402
import scipy.stats  # noqa: E402

import trtools.utils.tr_harmonizer as trh  # noqa: E402  (the reference)
import trtools.utils.utils as rutils  # noqa: E402
import trtools.statSTR.statSTR as rstat  # noqa: E402
import trtools.dumpSTR.dumpSTR as rdump  # noqa: E402
import trtools.dumpSTR.filters as rfilt  # noqa: E402

GOLD = os.path.join(REPO, 'tests', 'golden')


def jf(x):
    """JSON-safe float (nan/inf as strings)."""
    if x is None:
        return None
    x = float(x)
    if math.isnan(x):
        return "nan"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    return x


class DummyVariant:
    """Duck-typed cyvcf2.Variant (same idea as the reference's own
    utils/tests/test_trharmonizer.py:18-50 DummyCyvcf2Record)."""

    def __init__(self, gts, ref, alt, fmt=None):
        self.POS = 42
        self.CHROM = '1984'
        self.FORMAT = dict(fmt or {})
        self.INFO = {}
        self.ALT = list(alt)
        self.REF = ref
        g = np.asarray(gts, dtype=np.int16)
        self._gts = np.concatenate([g, np.zeros((g.shape[0], 1), dtype=np.int16)], axis=1)
        self.genotype = types.SimpleNamespace(array=lambda: self._gts, n_samples=g.shape[0])
        self.ploidy = g.shape[1]

    def format(self, key):
        return self.FORMAT.get(key, None)


def random_locus(rng, n_samples, ploidy, motif_len, n_alt, kind):
    motif = ''.join(rng.choice(list('ACGT'), size=motif_len))
    ref_copies = int(rng.integers(3, 12))
    ref = motif * ref_copies
    alts = []
    seen = {ref}
    tries = 0
    while len(alts) < n_alt and tries < 200:
        tries += 1
        mode = rng.random()
        copies = max(1, ref_copies + int(rng.integers(-3, 6)))
        a = motif * copies
        if mode < 0.2:      # fractional length (partial repeat)
            a = a + motif[: max(1, motif_len // 2)] if motif_len > 1 else a + motif
        elif mode < 0.4:    # same length, different sequence (impure)
            la = list(a)
            p = int(rng.integers(0, len(la)))
            la[p] = 'A' if la[p] != 'A' else 'C'
            a = ''.join(la)
        if a in seen:
            continue
        seen.add(a)
        alts.append(a)
    nall = 1 + len(alts)
    w = rng.dirichlet(np.full(nall, 0.6))
    gt = rng.choice(nall, size=(n_samples, ploidy), p=w).astype(np.int16)
    if kind == 'missing':
        m = rng.random(n_samples) < 0.25
        gt[m, :] = -1
        pm = rng.random(n_samples) < 0.1
        gt[pm, -1] = -1
    elif kind == 'allmissing':
        gt[:, :] = -1
    elif kind == 'partialonly':
        gt[:, -1] = -1
    elif kind == 'lowploidy' and ploidy > 1:
        m = rng.random(n_samples) < 0.3
        gt[m, -1] = -2
    elif kind == 'mono':
        gt[:, :] = 0
    elif kind == 'inbred' and ploidy == 2:
        m = rng.random(n_samples) < 0.5
        gt[m, 1] = gt[m, 0]
    return motif, ref, alts, gt


def capture(fn):
    try:
        return {"ok": fn()}
    except ValueError as e:
        return {"raises": "ValueError"}
    except IndexError as e:
        return {"raises": "IndexError"}


def dict_to_json(d, keyconv=str):
    return [[keyconv(k), (int(v) if isinstance(v, (int, np.integer)) else jf(v))] for k, v in d.items()]


def gen_trrecord_vectors():
    rng = np.random.default_rng(20260928)
    cases = []
    specs = []
    for kind in ['plain', 'missing', 'allmissing', 'partialonly', 'lowploidy', 'mono', 'inbred']:
        for ploidy in (1, 2, 3):
            for n_samples in (1, 7, 60):
                specs.append((kind, ploidy, n_samples))
    for kind, ploidy, n_samples in specs:
        for rep in range(2):
            motif_len = int(rng.integers(1, 7))
            n_alt = 0 if kind == 'mono' else int(rng.integers(0, 9))
            motif, ref, alts, gt = random_locus(rng, n_samples, ploidy, motif_len, n_alt, kind)
            var = DummyVariant(gt, ref, alts)
            rec = trh.TRRecord(var, ref, alts, motif, 'id', None)
            groups = [None]
            if n_samples >= 7:
                groups.append((rng.random(n_samples) < 0.5))
            for gi, si in enumerate(groups):
                c = collections.OrderedDict()
                c['kind'], c['ploidy'] = kind, ploidy
                c['gt'] = gt.tolist()
                c['ref'], c['alts'], c['motif'] = ref, alts, motif
                c['sample_index'] = None if si is None else [bool(x) for x in si]
                c['allele_lens'] = [jf(rec.ref_allele_length)] + [jf(x) for x in rec.alt_allele_lengths]
                c['called'] = [bool(x) for x in rec.GetCalledSamples()]
                c['called_nonstrict'] = [bool(x) for x in rec.GetCalledSamples(strict=False)]
                c['callrate'] = jf(rec.GetCallRate())
                c['ploidies'] = [int(x) for x in rec.GetSamplePloidies()]
                c['counts_len'] = dict_to_json(rec.GetAlleleCounts(sample_index=si, uselength=True), lambda k: repr(float(k)))
                c['counts_idx'] = dict_to_json(rec.GetAlleleCounts(sample_index=si, index=True), lambda k: str(int(k)))
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    c['counts_str'] = dict_to_json(rec.GetAlleleCounts(sample_index=si, uselength=False), str)
                    c['freqs_str'] = dict_to_json(rec.GetAlleleFreqs(sample_index=si, uselength=False), str)
                    gcs = rec.GetGenotypeCounts(sample_index=si, uselength=False)
                    c['gcounts_str'] = [[list(map(str, k)), int(v)] for k, v in gcs.items()]
                c['freqs_len'] = dict_to_json(rec.GetAlleleFreqs(sample_index=si, uselength=True), lambda k: repr(float(k)))
                gcl = rec.GetGenotypeCounts(sample_index=si, uselength=True)
                c['gcounts_len'] = [[[jf(x) for x in k], int(v)] for k, v in gcl.items()]
                c['maxallele'] = jf(rec.GetMaxAllele(sample_index=si))
                sis = [si]
                # statSTR column functions (statSTR.py:104-426)
                st = collections.OrderedDict()
                st['thresh'] = jf(rstat.GetThresh(rec, sis)[0])
                for ul in (True, False):
                    tag = 'len' if ul else 'str'
                    with warnings.catch_warnings():
                        warnings.simplefilter('ignore')
                        st['afreq_' + tag] = rstat.GetAFreq(rec, sis, uselength=ul)[0]
                        st['acount_' + tag] = rstat.GetAFreq(rec, sis, uselength=ul, count=True)[0]
                        st['nalleles_' + tag] = int(rstat.GetNAlleles(rec, sis, nalleles_thresh=0.1, uselength=ul)[0])
                        r = capture(lambda: rstat.GetHWEP(rec, sis, uselength=ul)[0])
                        st['hwep_' + tag] = {k: (jf(v) if k == 'ok' else v) for k, v in r.items()}
                        st['het_' + tag] = jf(rstat.GetHet(rec, sis, uselength=ul)[0])
                        st['entropy_' + tag] = jf(rstat.GetEntropy(rec, sis, uselength=ul)[0])
                st['mean'] = jf(rstat.GetMean(rec, sis)[0])
                st['mode'] = jf(rstat.GetMode(rec, sis)[0])
                st['var'] = jf(rstat.GetVariance(rec, sis)[0])
                st['numcalled'] = int(rstat.GetNumSamples(rec, sis)[0])
                c['statstr'] = st
                cases.append(c)
    with open(os.path.join(GOLD, 'trrecord_vectors.json'), 'w') as fh:
        json.dump({"generator": "tools/gen_golden.py gen_trrecord_vectors",
                   "reference": "gymrek-lab/TRTools v6.1.0 (imported)",
                   "cases": cases}, fh)
    print("trrecord_vectors.json:", len(cases), "cases")


def gen_binomtest_vectors():
    rng = np.random.default_rng(7)
    cases = []
    for n in [1, 2, 3, 5, 10, 17, 50, 100, 333, 1000, 4999, 10000, 20000]:
        for p in [1e-6, 0.001, 0.05, 0.25, 1 / 3, 0.5, 0.62, 0.9, 0.999, 1.0, 0.0]:
            ks = {0, n, n // 2, int(round(p * n)), max(0, int(round(p * n)) - 1), min(n, int(round(p * n)) + 1)}
            for _ in range(3):
                ks.add(int(rng.integers(0, n + 1)))
                # near the mode, where the two-sided search matters
                sd = math.sqrt(max(n * p * (1 - p), 1e-9))
                ks.add(int(min(n, max(0, round(p * n + rng.normal() * 2 * sd)))))
            for k in sorted(ks):
                pv = scipy.stats.binomtest(int(k), n=int(n), p=float(p)).pvalue
                cases.append([int(k), int(n), float(p), jf(pv)])
    # the reference's own known answers (utils/tests/test_utils.py:81-99 inputs)
    with open(os.path.join(GOLD, 'binomtest_vectors.json'), 'w') as fh:
        json.dump({"generator": "tools/gen_golden.py gen_binomtest_vectors",
                   "scipy": scipy.__version__, "cases": cases}, fh)
    print("binomtest_vectors.json:", len(cases), "cases")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    gens = collections.OrderedDict([
        ('trrecord', gen_trrecord_vectors),
        ('binomtest', gen_binomtest_vectors),
    ])
    try:
        import gen_golden_dumpstr  # noqa: F401  (second half, same directory)
        gens.update(gen_golden_dumpstr.GENERATORS)
    except ImportError:
        pass
    for name, fn in gens.items():
        if args.only and args.only != name:
            continue
        fn()


if __name__ == '__main__':
    main()
