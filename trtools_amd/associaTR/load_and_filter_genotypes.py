"""Genotype side of associaTR -- the public names of
trtools/associaTR/load_and_filter_genotypes.py (``dict_str``, ``clean_len_alleles``,
``clean_len_allele_pairs``, ``round_vals``, ``load_trs`` and the four ``*_precision`` module
knobs) on top of this package's harmoniser.

``load_trs`` is kept as the reference's per-locus generator for API compatibility; its
genotype / allele-frequency / filter arithmetic are plain array transforms of ONE record (the
same division as TRRecord.GetLengthGenotypes etc. in utils/tr_harmonizer.py).  The association
scan itself (associaTR.perform_gwas) does not go through it: it packs batches of loci and runs
``trk_assoc_scan`` on the device.
"""
import sys

import numpy as np

from ..utils import tr_harmonizer as trh
from ..utils import utils

allele_len_precision = 2
allele_frequency_precision = 2
dosage_precision = 2
r2_precision = 2

DETAIL_FIELDS = ('motif', 'period', 'ref_len', 'allele_frequency')
DOSAGE_DETAIL_FIELDS = ('dosage_estimated_r2_per_length_allele', 'r2_length_dosages_vs_best_guess_lengths')


def dict_str(d):
    """JSON-ish rendering of a dict with sorted, quoted keys (reference :23-35)."""
    items = ', '.join('{}: {}'.format(repr(str(k)), repr(d[k])) for k in sorted(d.keys()))
    text = '{' + items + '}'
    for old, new in (("'", '"'), ('(', '['), (')', ']'), ('nan', '"NaN"')):
        text = text.replace(old, new)
    return text


def _merge_rounded(d, keyfn):
    out = {}
    for key, val in d.items():
        nk = keyfn(key)
        if nk in out:
            out[nk] += val
        else:
            out[nk] = val
    return out


def clean_len_alleles(d):
    """Round the length keys to ``allele_len_precision`` and merge what collides (reference :37-45)."""
    return _merge_rounded(d, lambda k: round(k, allele_len_precision))


def clean_len_allele_pairs(d):
    return _merge_rounded(d, lambda k: (round(k[0], allele_len_precision), round(k[1], allele_len_precision)))


def round_vals(d, precision):
    return {key: round(val, precision) for key, val in d.items()}


def rounded_allele_lengths(trrecord):
    """[ref, alts...] lengths rounded the way load_trs does (python floats, python round)."""
    return [round(x, allele_len_precision) for x in [trrecord.ref_allele_length] + list(trrecord.alt_allele_lengths)]


def locus_details(trrecord, allele_frequency, extra=()):
    """The per-locus detail columns (reference :217-227)."""
    out = [trrecord.motif, str(len(trrecord.motif)), str(round(trrecord.ref_allele_length, allele_len_precision)),
           dict_str({key: '{:.2g}'.format(val) for key, val in allele_frequency.items()})]
    out.extend(extra)
    return out


def filter_reason(allele_frequency, n_samples, non_major_cutoff, beagle_dosages):
    """Locus filter of load_trs (reference :229-239): None or the reason string."""
    if len(allele_frequency) == 0:
        return 'No called samples'
    if len(allele_frequency) == 1:
        return 'Only one called allele'
    af = list(allele_frequency.values())
    af.pop(int(np.argmax(af)))
    if np.sum(af) * n_samples * 2 < non_major_cutoff:
        return 'non-major allele {}<{}'.format("dosage" if beagle_dosages else "count", non_major_cutoff)
    return None


def allele_frequency_from_counts(index_counts, allele_lens):
    """Counts per allele INDEX (tested samples) -> the allele_frequency dict of load_trs:
    GetAlleleFreqs keyed by length (ascending numpy keys, tr_harmonizer.py:1420-1540) then
    clean_len_alleles."""
    lens = np.asarray(allele_lens, dtype=np.float64)
    cnt = np.asarray(index_counts)
    keys = np.unique(lens[cnt > 0])
    by_len = {k: int(cnt[lens == k].sum()) for k in keys}
    total = float(sum(by_len.values()))
    return clean_len_alleles({k: v / total for k, v in by_len.items()})


def beagle_dosage_genotypes(trrecord, curr_samples, len_alleles):
    """Per-length dosage matrices from the AP1/AP2 fields (reference :179-189) and the two
    imputation-quality summaries (:191-215)."""
    n_samples = int(np.sum(curr_samples))
    gts = {_len: np.zeros((n_samples, 2)) for _len in np.unique(len_alleles)}
    for p in (1, 2):
        ap = trrecord.format['AP{}'.format(p)]
        gts[len_alleles[0]][:, (p - 1)] += np.maximum(0, 1 - np.sum(ap[curr_samples, :], axis=1))
        for i in range(ap.shape[1]):
            gts[len_alleles[i + 1]][:, (p - 1)] += ap[curr_samples, i]
    allele_frequency = {_len: np.sum(gts[_len]) / (2 * n_samples) for _len in gts}
    best_guesses = trrecord.GetLengthGenotypes()[curr_samples, :-1]
    rounded_best = np.around(best_guesses, allele_len_precision)
    allele_dosage_r2 = {}
    for length in len_alleles:
        if length in allele_dosage_r2:
            continue
        calls = rounded_best == length
        allele_dosage_r2[length] = np.corrcoef(calls.reshape(-1), gts[length].reshape(-1))[0, 1] ** 2
    length_r2 = np.corrcoef(best_guesses.flatten(),
                            np.add.reduce([len_ * dosages for len_, dosages in gts.items()]).flatten())[0, 1] ** 2
    return gts, allele_frequency, allele_dosage_r2, length_r2


def iter_records(vcf_fname, region=None, vcftype=None, beagle_dosages=False,
                 _imputed_ukb_strs_paper_period_check=False, attach=None):
    """Harmonised records of the file in the order / with the skipping rules of load_trs
    (reference :113-155): region restriction, records starting before the region, the PERIOD check."""
    vcf = utils.LoadSingleReader(vcf_fname, checkgz=False)
    if vcf is None:
        raise ValueError("could not open %s" % vcf_fname)
    inferred = trh.InferVCFType(vcf, vcftype if vcftype else 'auto')
    region_start = None
    if region is not None:
        region_start = int(region.split(':')[1].split('-')[0])
        vcf = vcf(region)
    elif attach is not None and attach.get('shard') is not None:
        # one process per GPU: this rank reads its contiguous share of the records only (statSTR._ShardedOut.attach)
        attach['shard'].attach(vcf)
    first = True
    for record in vcf:
        if first and beagle_dosages and "AP1" not in record.FORMAT:
            print("--beagle-dosages specified, missing required field AP1 for the TR")
            if "GP" in record.FORMAT:
                print("We could support the GP field, but currently only support the AP fields")
            print("Erroring out")
            sys.exit(1)
        first = False
        if region_start is not None and record.POS < region_start:
            continue
        if _imputed_ukb_strs_paper_period_check and record.INFO.get('PERIOD') is None:
            continue
        yield trh.HarmonizeRecord(vcfrecord=record, vcftype=inferred)


def load_trs(vcf_fname, samples, region=None, non_major_cutoff=20, beagle_dosages=False, vcftype=None,
             _imputed_ukb_strs_paper_period_check=False):
    """Generator with the reference's contract (:61-259): first the tuple of detail field names, then
    per locus ``(gts, unique_alleles, chrom, pos, called_samples_filter, locus_filtered, locus_details)``."""
    deets = list(DETAIL_FIELDS)
    if beagle_dosages:
        deets.extend(DOSAGE_DETAIL_FIELDS)
    yield deets
    for trrecord in iter_records(vcf_fname, region, vcftype, beagle_dosages, _imputed_ukb_strs_paper_period_check):
        called = trrecord.GetCalledSamples()
        if isinstance(samples, slice):
            assert samples == slice(None)
            called_samples_filter = called
            curr_samples = called
        else:
            called_samples_filter = called[samples]
            curr_samples = samples & called
        n_samples = int(np.sum(curr_samples))
        len_alleles = rounded_allele_lengths(trrecord)
        extra = ()
        if not beagle_dosages:
            gts = trrecord.GetLengthGenotypes()[curr_samples, :-1]
            allele_frequency = clean_len_alleles(trrecord.GetAlleleFreqs(curr_samples))
        else:
            gts, allele_frequency, dosage_r2, length_r2 = beagle_dosage_genotypes(trrecord, curr_samples, len_alleles)
            extra = (dict_str(round_vals(dosage_r2, r2_precision)), str(round(length_r2, r2_precision)))
        reason = filter_reason(allele_frequency, n_samples, non_major_cutoff, beagle_dosages)
        yield (None if reason else gts, np.unique(len_alleles), trrecord.chrom, trrecord.pos, called_samples_filter,
               reason, locus_details(trrecord, allele_frequency, extra))
