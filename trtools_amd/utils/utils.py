"""Scalar statistics and allele-string helpers with the reference's names and
return values (trtools/utils/utils.py).

The per-sample reductions of the hot path never come through here: they run on
the GPU (libtrk).  These functions are the dict-level API the reference exposes
(utils.py:118-338) plus the O(allele length) host-side string helpers used
while harmonising a record (utils.py:340-602)."""
from .. import _knobs
import argparse
import math
import os

import numpy as np

from . import common

_NUC = {"A": 0, "C": 1, "G": 2, "T": 3}


# ---------------------------------------------------------------------------
# readers (utils.py:19-116)
# ---------------------------------------------------------------------------

def LoadSingleReader(vcf_loc, checkgz=True, lazy=False, samples=None):
    """Open a VCF; returns None (after a warning) when it cannot be used
    (utils.py:19-67).  The reader is this package's own decoder
    (trtools_amd.vcfio) instead of cyvcf2."""
    from .. import vcfio
    if not os.path.exists(vcf_loc) or os.path.isdir(vcf_loc):
        common.WARNING("Could not find VCF file %s" % vcf_loc)
        return None
    if checkgz:
        if not vcf_loc.endswith(".vcf.gz") and not vcf_loc.endswith(".vcf.bgz"):
            common.WARNING("Make sure %s is bgzipped and indexed" % vcf_loc)
            return None
        if not os.path.isfile(vcf_loc + ".tbi"):
            common.WARNING("Could not find VCF index %s.tbi" % vcf_loc)
            return None
    if samples is not None:
        if not isinstance(samples, set):
            common.WARNING("Samples cannot be loaded in a particular order. Order will be ignored")
        samples = list(samples)
    try:
        if _knobs.lab('TRK_NATIVE_VCF', '1') != '0':
            from .. import vcfnative
            return vcfnative.NativeVCFReader(vcf_loc, lazy=lazy, samples=samples)
        return vcfio.VCFReader(vcf_loc, lazy=lazy, samples=samples)
    except OSError:
        common.WARNING("Could not open VCF file %s. Is it really VCF?" % vcf_loc)
        return None


def LoadReaders(vcf_locs, checkgz=True):
    """utils.py:69-96."""
    readers = []
    for loc in vcf_locs:
        r = LoadSingleReader(loc, checkgz)
        if r is None:
            return None
        readers.append(r)
    return readers


def GetContigs(vcf):
    """Contig IDs declared in the header (utils.py:98-116)."""
    return [h['ID'] for h in vcf.header_iter() if h['HeaderType'].lower() == 'contig']


# ---------------------------------------------------------------------------
# allele-frequency statistics on dicts (utils.py:118-338)
# ---------------------------------------------------------------------------

def ValidateAlleleFreqs(allele_freqs):
    """True when the dict is non-empty and sums to 1 within 1e-3 (utils.py:118-140)."""
    if len(allele_freqs) == 0:
        return False
    return abs(1 - sum(allele_freqs.values())) <= 0.001


def GetHeterozygosity(allele_freqs):
    """1 - sum p_i^2, nan for an invalid distribution (utils.py:142-175)."""
    if not ValidateAlleleFreqs(allele_freqs):
        return np.nan
    return 1 - sum(p ** 2 for p in allele_freqs.values())


def GetEntropy(allele_freqs):
    """Bit entropy of the allele distribution (utils.py:178-212; the reference
    calls scipy.stats.entropy(base=2): normalise, -sum p ln p, divide by ln 2)."""
    if not ValidateAlleleFreqs(allele_freqs):
        return np.nan
    pk = np.asarray(list(allele_freqs.values()), dtype=float)
    pk = pk / np.sum(pk)
    nz = pk[pk > 0]
    s = -np.sum(nz * np.log(nz))
    return float(s / math.log(2)) + 0.0


def GetMean(allele_freqs):
    """utils.py:215-236."""
    if not ValidateAlleleFreqs(allele_freqs):
        return np.nan
    return sum(k * p for k, p in allele_freqs.items())


def GetMode(allele_freqs):
    """Most frequent allele length; the smallest one on ties (utils.py:238-271)."""
    if not ValidateAlleleFreqs(allele_freqs):
        return np.nan
    top = max(allele_freqs.values())
    return min(k for k, p in allele_freqs.items() if p == top)


def GetVariance(allele_freqs):
    """utils.py:273-296."""
    if not ValidateAlleleFreqs(allele_freqs):
        return np.nan
    mean = GetMean(allele_freqs)
    return sum(p * (k - mean) ** 2 for k, p in allele_freqs.items())


def GetHardyWeinbergBinomialTest(allele_freqs, genotype_counts):
    """Two-sided exact binomial test of the homozygote count (utils.py:298-338).

    The p-value comes from libtrk's ``trk_binomtest_two_sided`` (the same code
    the GPU finaliser runs), which reproduces ``scipy.stats.binomtest``.
    Raises what the reference raises: ValueError when there is no genotype
    (scipy: n must be >= 1), IndexError for haploid genotype tuples."""
    if not ValidateAlleleFreqs(allele_freqs):
        return np.nan
    exp_hom_frac = sum(v ** 2 for v in allele_freqs.values())
    total = sum(genotype_counts.values())
    num_hom = 0
    for gt, n in genotype_counts.items():
        if gt[0] not in allele_freqs:
            return np.nan
        if gt[1] not in allele_freqs:
            return np.nan
        if gt[0] == gt[1]:
            num_hom += n
    if total < 1:
        raise ValueError("n must be a positive integer (no called genotypes)")
    from .. import _lib
    return _lib.load().trk_binomtest_two_sided(int(num_hom), int(total), float(exp_hom_frac))


# ---------------------------------------------------------------------------
# string helpers (utils.py:340-602)
# ---------------------------------------------------------------------------

def GetHomopolymerRun(seq):
    """Length of the longest run of one nucleotide (utils.py:340-360)."""
    if len(seq) == 0:
        return 0
    seq = seq.upper()
    best = run = 1
    for i in range(1, len(seq)):
        run = run + 1 if seq[i] == seq[i - 1] else 1
        if run > best:
            best = run
    return best


def _nuc_key(s):
    return [_NUC[c] for c in s]


def GetCanonicalOneStrand(repseq):
    """Alphabetically first rotation of the motif (utils.py:396-427)."""
    repseq = repseq.upper()
    best = repseq
    best_key = _nuc_key(best)
    for i in range(1, len(repseq)):
        rot = repseq[i:] + repseq[:i]
        key = _nuc_key(rot)
        if key < best_key:
            best, best_key = rot, key
    return best


def ReverseComplement(seq):
    """utils.py:429-463 (anything but ACGT becomes N)."""
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    return "".join(comp.get(c, "N") for c in reversed(seq.upper()))


def GetCanonicalMotif(repseq):
    """Alphabetically first rotation over both strands (utils.py:362-394)."""
    repseq = repseq.upper()
    fwd = GetCanonicalOneStrand(repseq)
    rev = GetCanonicalOneStrand(ReverseComplement(repseq))
    return rev if _nuc_key(rev) < _nuc_key(fwd) else fwd


def InferRepeatSequence(seq, period):
    """Most frequent in-frame k-mer of ``seq`` (frame 0), canonicalised on one strand
    (utils.py:465-508: the k-mer that first reaches the maximal copy number)."""
    if period > len(seq):
        return "N" * period
    counts = {}
    best_kmer, best = None, 0
    for start in range(0, len(seq) - period + 1, period):
        kmer = seq[start:start + period]
        c = counts.get(kmer, 0) + 1
        counts[kmer] = c
        if c > best:
            best_kmer, best = kmer, c
    return GetCanonicalOneStrand(best_kmer)


def FabricateAllele(motif, length):
    """``length`` copies of ``motif`` (fractional part -> leading bases of the motif,
    rounded down; utils.py:566-602)."""
    fab = math.floor(length) * motif
    i = 0
    while (len(fab) + 1) / len(motif) < length:
        fab += motif[i]
        i += 1
    return fab


class ArgumentDefaultsHelpFormatter(argparse.HelpFormatter):
    """Like argparse.ArgumentDefaultsHelpFormatter but silent about None defaults
    (utils.py:605-626)."""

    def _get_help_string(self, action):
        text = action.help
        if '%(default)' not in text and action.default is not argparse.SUPPRESS \
                and action.default is not None:
            if action.option_strings or action.nargs in (argparse.OPTIONAL, argparse.ZERO_OR_MORE):
                text += ' (default: %(default)s)'
        return text
