"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product (`trtools_amd/`).

A CPU restatement (numpy + scipy, one locus at a time, exactly like the
reference) of the TRTools per-locus hot path:

    tr_harmonizer.TRRecord reductions  ->  utils stat scalars  ->
    statSTR columns / dumpSTR call filters, sample counters, locus filters.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module, and only as the checker.

Every function cites the reference file:line (paths relative to the TRTools
checkout, v6.1.0) whose arithmetic it follows.  The reference consumes
`cyvcf2.Variant` objects; this restatement takes the arrays those objects hand
over (cyvcf2 conventions):

  * ``gt``      int array ``[S, P]`` of allele indices, ``-1`` = missing
                haplotype, ``-2`` = ploidy padding (the phased column of
                ``genotype.array()`` is already dropped);
  * ``alleles`` python list, one entry per allele index (0 = ref): floats
                (lengths in repeat units) or upper-case strings (sequences);
  * FORMAT fields as ``int32`` (missing = INT_MIN) / ``float32`` (missing = nan)
    arrays ``[S, k]`` or string arrays ``[S]``.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function
below against (i) vectors produced by importing the real reference in the build
container (`tools/gen_golden.py`, fixtures under `tests/golden/`), (ii) the
literal known answers in the reference's own unit tests
(utils/tests/test_utils.py:21-99, utils/tests/test_trharmonizer.py:441-715,
dumpSTR/tests/test_filters.py:86-452) and (iii) the reference's golden output
files (sample_stats/many_samples_all*.tab, dumpSTR_vcfs/*.{loc,samp}log.tab).

Third-party arithmetic on the path (not under /root/reference):
  scipy.stats.binomtest / entropy  -- called directly, as the reference does
  (utils.py:212,334-338); numpy.unique / sort -- called directly.
"""
import ast
import warnings
import collections

import numpy as np
import scipy.stats

INT_MISSING = -2147483648  # dumpSTR.py:610  _NOCALL_INT_FORMAT_VAL


# ----------------------------------------------------------------------------
# TRRecord reductions (tr_harmonizer.py)
# ----------------------------------------------------------------------------

def get_called_samples(gt, strict=True):
    """tr_harmonizer.py:864-897 GetCalledSamples."""
    gt = np.asarray(gt)
    if strict:
        return ~np.any(gt == -1, axis=1)
    return ~np.all(np.logical_or(gt == -1, gt == -2), axis=1)


def get_sample_ploidies(gt):
    """tr_harmonizer.py:899-919 GetSamplePloidies."""
    gt = np.asarray(gt)
    return gt.shape[1] - np.sum(gt == -2, axis=1)


def get_call_rate(gt, strict=True):
    """tr_harmonizer.py:921-946 GetCallRate."""
    called = get_called_samples(gt, strict)
    return np.sum(called) / called.shape[0]


def get_length_genotypes(gt, allele_lens):
    """tr_harmonizer.py:1210-1245 GetLengthGenotypes (phase column dropped)."""
    gt = np.asarray(gt).astype(int)
    lut = np.array([*allele_lens, -2, -1], dtype=float)   # :1239
    return lut[gt]                                        # :1242


def get_string_genotypes(gt, seq_alleles):
    """tr_harmonizer.py:948-961 _GetStringGenotypeArray (phase column dropped)."""
    gt = np.asarray(gt).astype(int)
    max_len = max(len(a) for a in seq_alleles)
    arr = np.empty(gt.shape, dtype="<U{}".format(max_len))
    for idx, seq in enumerate(seq_alleles):
        arr[gt == idx] = seq
    arr[gt == -1] = '.'
    arr[gt == -2] = ','
    return arr


def _rep_genotypes(gt, alleles, index=False):
    """Dispatch of GetAlleleCounts/GetGenotypeCounts :1381-1392 / :1466-1481."""
    gt = np.asarray(gt)
    if index:
        return gt.astype(int), -1, -2
    if len(alleles) and isinstance(alleles[0], str):
        return get_string_genotypes(gt, alleles), '.', ','
    return get_length_genotypes(gt, alleles), -1, -2


def get_allele_counts(gt, alleles, sample_index=None, index=False):
    """tr_harmonizer.py:1420-1499 GetAlleleCounts -> {allele: count}."""
    gts, nocall, lowploidy = _rep_genotypes(gt, alleles, index)
    if gts.shape[0] == 0:
        return {}
    if sample_index is not None:
        gts = gts[sample_index, :]                 # :1488-1489
    gts = gts[gts != nocall]                       # :1492
    gts = gts[gts != lowploidy]                    # :1493
    vals, counts = np.unique(gts, return_counts=True)  # :1495
    return dict(zip(vals, counts))


def get_allele_freqs(gt, alleles, sample_index=None, index=False):
    """tr_harmonizer.py:1501-1540 GetAlleleFreqs."""
    counts = get_allele_counts(gt, alleles, sample_index, index)
    total = float(sum(counts.values()))
    return {k: v / total for k, v in counts.items()}


def get_genotype_counts(gt, alleles, sample_index=None, index=False,
                        include_nocalls=False):
    """tr_harmonizer.py:1326-1418 GetGenotypeCounts -> {tuple: count}."""
    gts, nocall, _ = _rep_genotypes(gt, alleles, index)
    if gts.shape[0] == 0:
        return {}
    gts = np.sort(gts, axis=1)                     # :1398
    if sample_index is not None:
        gts = gts[sample_index, :]
    if gts.shape[0] == 0:
        return {}
    genotypes, counts = np.unique(gts, axis=0, return_counts=True)  # :1403
    out = dict(zip(tuple(map(tuple, genotypes)), counts))
    if not include_nocalls:
        for g in [g for g in out if nocall in g]:  # :1410-1416
            del out[g]
    return out


def get_max_allele(gt, allele_lens, sample_index=None):
    """tr_harmonizer.py:1542-1575 GetMaxAllele."""
    keys = get_allele_counts(gt, list(allele_lens), sample_index).keys()
    if len(keys) == 0:
        return np.nan
    return max(keys)


# ----------------------------------------------------------------------------
# utils stat scalars (utils/utils.py)
# ----------------------------------------------------------------------------

def validate_allele_freqs(freqs):
    """utils.py:118-140."""
    if len(freqs.keys()) == 0:
        return False
    return abs(1 - sum(freqs.values())) <= 0.001


def get_heterozygosity(freqs):
    """utils.py:142-175."""
    if not validate_allele_freqs(freqs):
        return np.nan
    return 1 - sum([f ** 2 for f in freqs.values()])


def get_entropy(freqs):
    """utils.py:178-212 (scipy.stats.entropy, base 2)."""
    if not validate_allele_freqs(freqs):
        return np.nan
    return float(scipy.stats.entropy(list(freqs.values()), base=2))


def get_mean(freqs):
    """utils.py:215-236."""
    if not validate_allele_freqs(freqs):
        return np.nan
    return sum([k * freqs[k] for k in freqs])


def get_mode(freqs):
    """utils.py:238-271 (ties -> smallest allele)."""
    if not validate_allele_freqs(freqs):
        return np.nan
    mode_freq = -1
    modes = set()
    for allele, freq in freqs.items():
        if freq > mode_freq:
            modes = {allele}
            mode_freq = freq
        if freq == mode_freq:
            modes.add(allele)
    return min(modes)


def get_variance(freqs):
    """utils.py:273-296."""
    if not validate_allele_freqs(freqs):
        return np.nan
    mean = get_mean(freqs)
    return sum([freqs[k] * (k - mean) ** 2 for k in freqs.keys()])


def get_hwe_binomial_test(freqs, genotype_counts):
    """utils.py:298-338 GetHardyWeinbergBinomialTest.

    Raises exactly what the reference raises: ValueError from scipy when the
    genotype table is empty (n < 1), IndexError for 1-tuples (haploid loci).
    """
    if not validate_allele_freqs(freqs):
        return np.nan
    exp_hom_frac = sum([v ** 2 for v in freqs.values()])
    total = sum(genotype_counts.values())
    num_hom = 0
    for gt in genotype_counts:
        if gt[0] not in freqs.keys():
            return np.nan
        if gt[1] not in freqs.keys():
            return np.nan
        if gt[0] == gt[1]:
            num_hom += genotype_counts[gt]
    return scipy.stats.binomtest(int(num_hom), n=int(total), p=exp_hom_frac).pvalue


# ----------------------------------------------------------------------------
# statSTR columns (statSTR/statSTR.py:104-426), one locus, one sample group
# ----------------------------------------------------------------------------

HWE_OK, HWE_VALUE_ERROR, HWE_INDEX_ERROR = 0, 2, 3


def locus_stats(gt, allele_lens, allele_strs, sample_index=None,
                use_length=True, nalleles_thresh=0.01):
    """All eleven statSTR statistics of one locus for one sample group.

    statSTR.py:104-426: thresh/mean/mode/var/numcalled are ALWAYS length based
    (:126,347,375,402,426); afreq/acount/nalleles/hwep/het/entropy follow
    ``use_length`` (:593-616).  Returns a dict; 'hwep_status' records the
    exception the reference would raise instead of a value.
    """
    gt = np.asarray(gt)
    lens = list(allele_lens)
    reps = lens if use_length else list(allele_strs)
    out = collections.OrderedDict()
    out['thresh'] = get_max_allele(gt, lens, sample_index)
    counts = get_allele_counts(gt, reps, sample_index)
    freqs = get_allele_freqs(gt, reps, sample_index)
    out['acount'] = counts
    out['afreq'] = freqs
    out['nalleles'] = len([None for _, f in freqs.items() if f >= nalleles_thresh])
    gcounts = get_genotype_counts(gt, reps, sample_index)
    out['hwep_status'] = HWE_OK
    try:
        out['hwep'] = get_hwe_binomial_test(freqs, gcounts)
    except ValueError:
        out['hwep'] = np.nan
        out['hwep_status'] = HWE_VALUE_ERROR
    except IndexError:
        out['hwep'] = np.nan
        out['hwep_status'] = HWE_INDEX_ERROR
    out['het'] = get_heterozygosity(freqs)
    out['entropy'] = get_entropy(freqs)
    lfreqs = get_allele_freqs(gt, lens, sample_index)
    out['mean'] = get_mean(lfreqs)
    out['mode'] = get_mode(lfreqs)
    out['var'] = get_variance(lfreqs)
    out['numcalled'] = sum(get_genotype_counts(gt, lens, sample_index).values())
    # integer reductions the device kernel emits (checked bit-exactly)
    idx_counts = get_allele_counts(gt, lens, sample_index, index=True)
    n_alleles = len(lens)
    out['index_counts'] = np.array([int(idx_counts.get(i, 0)) for i in range(n_alleles)],
                                   dtype=np.int64)
    sub = gt if sample_index is None else gt[sample_index, :]
    out['n_called'] = int(np.sum(get_called_samples(sub))) if sub.shape[0] else 0
    out['n_samples'] = int(sub.shape[0])
    return out


def format_afreq(d, count=False):
    """statSTR.py:158-172 GetAFreq text."""
    if len(d.keys()) == 0:
        return "."
    if count:
        return ",".join(["%s:%i" % (a, d.get(a, 0)) for a in sorted(d.keys())])
    return ",".join(["%s:%.3f" % (a, d.get(a, 0)) for a in sorted(d.keys())])


# ----------------------------------------------------------------------------
# dumpSTR call-level filters (dumpSTR/filters.py:327-867)
#   each returns float64[S]: nan = not filtered, else the triggering value
# ----------------------------------------------------------------------------

def filt_min_value(field, threshold):
    """filters.py:363-367 CallFilterMinValue.__call__ (compare in field dtype)."""
    vals = field[:, 0]
    out = np.full(vals.shape[0], np.nan)
    out[vals < threshold] = vals[vals < threshold]
    return out


def filt_max_value(field, threshold):
    """filters.py:405-409 CallFilterMaxValue.__call__."""
    vals = field[:, 0]
    out = np.full(vals.shape[0], np.nan)
    out[vals > threshold] = vals[vals > threshold]
    return out


def filt_ratio_gt(num, den, threshold):
    """filters.py:444-449 / 479-484 HipSTRCallFlankIndels / HipSTRCallStutter."""
    out = np.full(num.shape[0], np.nan)
    with np.errstate(divide='ignore', invalid='ignore'):
        ratio = num[:, 0] / den[:, 0]
    out[ratio > threshold] = ratio[ratio > threshold]
    return out


def filt_hipstr_min_supp_reads(gt, allreads, gb, threshold):
    """filters.py:519-567 HipSTRCallMinSuppReads.__call__.

    ``allreads`` / ``gb`` are the string FORMAT arrays (or None when ALLREADS
    is absent from the record).
    """
    n = np.asarray(gt).shape[0]
    called = get_called_samples(gt)
    if not np.any(called):
        return np.full(n, np.nan)
    if allreads is None:
        return np.zeros(n, dtype=float)
    to_check = called & (allreads != '') & (allreads != '.')
    if not np.any(to_check):
        out = np.full(n, np.nan)
        out[called] = 0
        return out
    first_gb = gb[to_check][0]
    if "/" in first_gb:
        delim = "/"
    elif "|" in first_gb:
        delim = "|"
    else:
        raise ValueError("Cant't identify phasing char ('|' or '/') in GB field")
    gbv = np.stack(np.char.split(gb[to_check], delim)).astype(int)
    ar = np.char.replace(allreads[to_check], ";", ",")
    ar = np.char.replace(ar, "|", ":")
    ar = np.char.add("{", np.char.add(ar, "}"))
    min_counts = np.full(n, np.nan)
    for i, single in enumerate(ar):
        reads = ast.literal_eval(single)
        mc = np.inf
        for g in gbv[i, :]:
            g = int(g)
            if g not in reads:
                mc = 0
            else:
                mc = min(mc, reads[g])
        min_counts[np.nonzero(to_check)[0][i]] = mc
    min_counts[min_counts >= threshold] = np.nan
    min_counts[called & ~to_check] = 0
    return min_counts


def filt_gangstr_qexp(gt, qexp, threshold, which):
    """filters.py:597-674 GangSTRCallExpansionProb{Hom,Het,Total}."""
    n = np.asarray(gt).shape[0]
    out = np.full(n, np.nan)
    called = get_called_samples(gt)
    if not np.any(called):
        return out
    if which == 'hom':
        prob = qexp[called, 2]
    elif which == 'het':
        prob = qexp[called, 1]
    else:
        prob = qexp[called, 1] + qexp[called, 2]
    out[np.nonzero(called)[0][prob < threshold]] = prob[prob < threshold]
    return out


def _parse_rc(rc):
    return np.stack(np.char.split(rc, ','), axis=0).astype(int)


def filt_gangstr_span_only(gt, rc, dp):
    """filters.py:686-697 GangSTRCallSpanOnly."""
    n = np.asarray(gt).shape[0]
    out = np.full(n, np.nan)
    called = get_called_samples(gt)
    if not np.any(called):
        return out
    rcv = _parse_rc(rc[called])
    hit = rcv[:, 1] == dp[called, 0]
    out[np.nonzero(called)[0][hit]] = rcv[:, 1][hit]
    return out


def filt_gangstr_spanbound_only(gt, rc, dp):
    """filters.py:711-722 GangSTRCallSpanBoundOnly."""
    n = np.asarray(gt).shape[0]
    out = np.full(n, np.nan)
    called = get_called_samples(gt)
    if not np.any(called):
        return out
    rcv = _parse_rc(rc[called])
    sb = rcv[:, 1] + rcv[:, 3]
    hit = sb == dp[called, 0]
    out[np.nonzero(called)[0][hit]] = sb[hit]
    return out


def filt_gangstr_bad_ci(gt, repcn, repci):
    """filters.py:739-757 GangSTRCallBadCI."""
    n = np.asarray(gt).shape[0]
    out = np.full(n, np.nan)
    called = get_called_samples(gt)
    if not np.any(called):
        return out
    ml = repcn[called]
    ci = np.stack(np.char.split(repci[called], ","))
    ci = np.char.split(ci, '-')
    ci = np.array(ci.tolist(), dtype=int)
    per_gt = np.logical_or(ml < ci[:, :, 0], ci[:, :, 1] < ml)
    hit = np.any(per_gt, axis=1)
    if not np.any(hit):
        return out
    which = np.argmax(per_gt[hit, :], axis=1)
    out[np.nonzero(called)[0][hit]] = ml[hit, which]
    return out


def filt_popstr_require_support(gt, ad, threshold):
    """filters.py:858-867 PopSTRCallRequireSupport (numpy negative indexing kept)."""
    gt = np.asarray(gt).astype(int)
    n = gt.shape[0]
    out = np.full(n, np.nan)
    rows = np.arange(n)
    for ploid in range(gt.shape[1]):
        new = ad[rows, gt[:, ploid]] < threshold
        out[new] = ad[new, gt[new, ploid]]
    return out


# ----------------------------------------------------------------------------
# dumpSTR ApplyCallFilters / ApplyLocusFilters (dumpSTR/dumpSTR.py:613-973)
# ----------------------------------------------------------------------------

def apply_call_filters(gt, filter_outputs, sample_info, dp=None, locus_ploidy=None,
                       want_text=False):
    """dumpSTR.py:613-774 ApplyCallFilters on arrays.

    ``filter_outputs`` is an ordered list of ``(name, float64[S])`` (the
    outputs of the filt_* functions above, in BuildCallFilters order :792-872).
    ``sample_info`` (OrderedDict name -> array) is updated in place.
    ``dp``: the DP (or LC) FORMAT array ``[S,1]`` or None (:688-713).
    Returns (masked_gt, filtered_samples[, filter_text]).
    """
    gt = np.array(gt)
    n = gt.shape[0]
    nocalls = ~get_called_samples(gt)                      # :651
    any_filter = np.zeros(n, dtype=bool)
    text = np.empty(n, 'U4')
    for name, out in filter_outputs:
        nans = np.isnan(out)
        if np.all(nans):                                   # :659
            continue
        sample_info[name] += np.logical_and(~nans, ~nocalls)   # :661
        any_filter |= ~nans
        if want_text:                                      # :664-673
            t = np.char.add(name, np.char.add('_', np.char.mod('%g', out)))
            t[nans] = ''
            nf = np.logical_and(~nans, text != '')
            text[nf] = np.char.add(text[nf], ',')
            text = np.char.add(text, t)
    # :676-683  NOCALL overrides any filter text, '' -> PASS
    extant = np.logical_and(~any_filter, ~nocalls)         # text == 'PASS'  (:686)
    if want_text:
        if np.any(nocalls):
            nt = np.empty(n, dtype='U6')
            nt[nocalls] = 'NOCALL'
            text[nocalls] = ''
            text = np.char.add(text, nt)
        text[text == ''] = 'PASS'
        assert np.array_equal(extant, text == 'PASS')
    sample_info['numcalls'] += extant                      # :687
    if dp is not None:                                     # :696-711
        dpv = dp.reshape(-1)
        neg = np.logical_and(np.logical_and(dpv < 0, dpv != INT_MISSING), extant)
        if np.any(neg):
            raise ValueError("negative DP for called samples")
        acc = np.logical_and(extant, dpv > 0)
        sample_info['totaldp'][acc] += dpv[acc]
        sample_info['totaldp'][np.logical_and(extant, dpv == INT_MISSING)] = np.nan
    else:
        sample_info['totaldp'][:] = np.nan                 # :713
    filtered = np.logical_and(any_filter, ~nocalls)        # :715-717
    ploidy = gt.shape[1] if locus_ploidy is None else locus_ploidy
    gt[filtered, :ploidy] = -1                             # :722-724
    if want_text:
        return gt, filtered, text
    return gt, filtered


def locus_filter_values(gt, allele_lens, allele_strs, use_length):
    """The scalars the tool-agnostic locus filters look at
    (filters.py:59-61, 98-103, 140-144, 181-185)."""
    reps = list(allele_lens) if use_length else list(allele_strs)
    freqs = get_allele_freqs(gt, reps)
    het = get_heterozygosity(freqs)
    gcounts = get_genotype_counts(gt, reps)
    hwep = get_hwe_binomial_test(freqs, gcounts)
    return get_call_rate(gt), hwep, het


def apply_locus_filters(gt, allele_lens, allele_strs, loc_info, use_length=False,
                        min_callrate=None, min_hwep=None, min_het=None, max_het=None,
                        extra_filters=()):
    """dumpSTR.py:917-973 ApplyLocusFilters with BuildLocusFilters order :890-915.

    ``extra_filters``: ordered (name, bool) pairs for the host-side filters
    (HRUN :190-217, region filters :219-300) evaluated elsewhere.
    Returns (filtered, [names of triggered filters]).
    """
    names = []
    need_stats = any(v is not None for v in (min_hwep, min_het, max_het))
    if min_callrate is not None:
        if get_call_rate(gt) < min_callrate:               # filters.py:60
            names.append('CALLRATE' + str(min_callrate))
    if need_stats:
        reps = list(allele_lens) if use_length else list(allele_strs)
        freqs = get_allele_freqs(gt, reps)
    if min_hwep is not None:
        hwep = get_hwe_binomial_test(freqs, get_genotype_counts(gt, reps))
        if hwep < min_hwep:                                # filters.py:102
            names.append('HWE' + str(min_hwep))
    if min_het is not None:
        if get_heterozygosity(freqs) < min_het:            # filters.py:142
            names.append('HETLOW' + str(min_het))
    if max_het is not None:
        if get_heterozygosity(freqs) > max_het:            # filters.py:183
            names.append('HETHIGH' + str(max_het))
    for name, hit in extra_filters:
        if hit:
            names.append(name)
    for nm in names:
        loc_info[nm] += 1                                  # :949
    filtered = len(names) > 0
    n_called = np.sum(get_called_samples(gt))              # :957
    if n_called == 0:
        loc_info['NO_CALLS_REMAINING'] += 1
        names.append('NO_CALLS_REMAINING')
        filtered = True
    if not filtered:
        loc_info['PASS'] += 1
        loc_info['totalcalls'] += n_called                 # :971
    return filtered, names


def locus_info_fields(gt, allele_lens, allele_strs, use_length):
    """dumpSTR.py:1313-1336 recomputed INFO fields HET/HWEP/AC/REFAC."""
    n_alleles = len(allele_lens)
    if np.sum(get_called_samples(gt)) > 0:
        reps = list(allele_lens) if use_length else list(allele_strs)
        freqs = get_allele_freqs(gt, reps)
        het = get_heterozygosity(freqs)
        hwep = get_hwe_binomial_test(freqs, get_genotype_counts(gt, reps))
        ac = get_allele_counts(gt, allele_lens, index=True)
        acl = [int(ac.get(i, 0)) for i in range(n_alleles)]
        return het, hwep, acl[1:], acl[0]
    return -1, -1, [0] * (n_alleles - 1), 0


def samplog_rows(sample_info, sample_names):
    """dumpSTR.py:553-588 WriteSampLog text rows (header + one row per sample)."""
    import itertools
    header = ["sample"] + list(sample_info.keys())
    header[header.index('totaldp')] = 'meanDP'
    rows = ["\t".join(header)]
    for i, s in enumerate(sample_names):
        numcalls = sample_info["numcalls"][i]
        cols = [s, str(numcalls)]
        if numcalls > 0:
            cols.append(str(sample_info["totaldp"][i] * 1.0 / numcalls))
        else:
            cols.append("0")
        for fc in itertools.islice(sample_info.values(), 2, None):
            cols.append(str(fc[i]))
        rows.append("\t".join(cols))
    return rows


def loclog_rows(loc_info):
    """dumpSTR.py:523-551 WriteLocLog text rows."""
    keys = list(loc_info.keys())
    keys.remove("totalcalls")
    if loc_info["PASS"] == 0:
        callrate = 0
    else:
        callrate = float(loc_info["totalcalls"]) / loc_info["PASS"]
    rows = ["MeanSamplesPerPassingSTR\t%s" % callrate]
    for k in keys:
        rows.append("FILTER:%s\t%s" % (k, loc_info[k]))
    return rows


# ----------------------------------------------------------------------------
# per-sample dosages (tr_harmonizer.py:1098-1208)
# ----------------------------------------------------------------------------

def get_dosages(gt, allele_lens, dosagetype='bestguess', ap1=None, ap2=None):
    """TRRecord.GetDosages(dosagetype, strict=False) on arrays: gt [S, P] allele indices, allele_lens
    [ref, alts...], ap1/ap2 float32 [S, A-1].  Returns float32 [S] (a row of nan where the reference
    warns and gives up)."""
    gt = np.asarray(gt).astype(int)
    n = gt.shape[0]
    lens = [float(x) for x in allele_lens]
    ref_len, alt_lens = lens[0], lens[1:]
    norm = dosagetype.endswith('_norm')
    if dosagetype.startswith('bestguess'):
        lut = np.array([*lens, -2, -1])                       # tr_harmonizer.py:1239
        lengts = lut[gt]
        fill = np.nan if norm else 0
        lengts[gt == -1] = fill
        lengts[gt == -2] = fill
        unnorm = lengts.sum(axis=1).astype(np.float32)
    else:
        ref1 = np.clip(1 - np.sum(ap1, axis=1), 0, 1)
        ref2 = np.clip(1 - np.sum(ap2, axis=1), 0, 1)
        if np.any(np.sum(ap1, axis=1) > 1.1) or np.any(np.sum(ap2, axis=1) > 1.1):
            return np.array([np.nan] * n, dtype=np.float32)
        if np.any(ap1 < 0) or np.any(ap2 < 0):
            return np.array([np.nan] * n, dtype=np.float32)
        if len(alt_lens) > 0:
            cap = max(alt_lens)
            h1 = np.clip(np.dot(ap1, alt_lens), 0, cap)
            h2 = np.clip(np.dot(ap2, alt_lens), 0, cap)
        else:
            h1 = h2 = 0
        unnorm = (h1 + h2 + ref1 * ref_len + ref2 * ref_len).astype(np.float32)
    if not norm:
        return unnorm
    lo, hi = min(lens), max(lens)
    if lo == hi:
        return np.zeros(n, dtype=np.float32)
    dos = (unnorm - 2 * lo) / (hi - lo)
    with np.errstate(invalid='ignore'):
        if np.any(dos >= 2.1) or np.any(dos <= -0.1):
            return np.array([np.nan] * n, dtype=np.float32)
    return np.clip(dos, 0, 2)


# ----------------------------------------------------------------------------
# qcSTR's per-record reductions (qcSTR/qcSTR.py:529-561)
# ----------------------------------------------------------------------------

def qc_record(gt, quality=None, sample_index=None, ignore_no_call=False):
    """One iteration of qcSTR's main loop on arrays: gt [S, P] allele indices of the record's own ploidy columns,
    quality float32 [S, 1] (GetQualityScores) or None, sample_index boolean [S] or None.
    Returns (calls bool [S'], q float32 [S', 1] after the nan handling or None, per-locus mean or None) where S' are
    the selected samples -- exactly the values qcSTR.py:533-556 computes."""
    gt = np.asarray(gt)
    if sample_index is None:
        sample_index = np.ones(gt.shape[0], dtype=bool)
    idx_gts = gt[sample_index, :]                                   # qcSTR.py:533
    nocall = np.full((1, idx_gts.shape[1]), -1)
    calls = ~np.all(idx_gts == nocall, axis=1)                      # 534-535
    if quality is None:
        return calls, None, None
    q = np.array(quality, dtype=np.float32)[sample_index, :]       # 539 (a copy: the reference writes into it)
    q[~calls] = np.nan                                              # 540
    if not ignore_no_call:
        q[np.isnan(q)] = 0                                          # 541-542
        mean = np.mean(q)                                           # 553-554
    else:
        idxs = ~np.isnan(q)                                         # 543-544
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            mean = np.mean(q[idxs])                                 # 555-556 (nan, with a warning, when none)
    return calls, q, mean


def qc_accumulate(records, sample_index=None, ignore_no_call=False):
    """qcSTR's accumulators over a list of (chrom, gt [S, P], quality [S, 1] or None): sample_calls (536),
    chrom_calls (537), per-sample quality totals (546-551) and the per-locus means (552-556), plus the per-sample
    means as qcSTR.py:619-621 takes them."""
    sample_calls = None
    chrom_calls = {}
    totals = None
    per_locus = []
    n = 0
    for chrom, gt, quality in records:
        calls, q, mean = qc_record(gt, quality, sample_index, ignore_no_call)
        if sample_calls is None:
            sample_calls = np.zeros(len(calls))
            totals = np.zeros(len(calls))
        sample_calls += calls
        chrom_calls[chrom] = chrom_calls.get(chrom, 0) + np.sum(calls)
        if q is not None:
            if not ignore_no_call:
                totals += q.reshape(-1)
            else:
                idxs = ~np.isnan(q)
                totals[idxs.reshape(-1)] += q[idxs].reshape(-1)
            per_locus.append(mean)
        n += 1
    with np.errstate(invalid='ignore', divide='ignore'):
        per_sample = None if totals is None else (totals / n if not ignore_no_call else totals / sample_calls)
    return dict(sample_calls=sample_calls, chrom_calls=chrom_calls, per_sample_total=totals,
                per_sample_quality=per_sample, per_locus_quality=per_locus, numrecords=n)
