#!/usr/bin/env python3
"""associaTR: association of TR length with a phenotype -- same command line, same
``main(args)`` / ``perform_gwas(...)`` and the same output table as the reference
(trtools/associaTR/associaTR.py), with the per-locus loop (associaTR.py:246-291: numpy
standardisation + one statsmodels OLS fit per locus) replaced by batches of loci scanned on
the GPU (``trk_assoc_scan``: genotype x trait cross-products in one pass over the genotype
tensor, a Cholesky solve and a Student-t tail per locus).

Host Python assembles the covariates (associaTR.py:138-204), parses / harmonises records and
formats text; ``--beagle-dosages`` regresses on the expected length from the AP1/AP2 fields
(``trk_assoc_scan_dosage``).  Not provided (refused loudly): the reference's hidden, unfinished
``--plotting-phenotype`` family of options.
"""
import argparse
import datetime
import shutil
import sys
import time

import numpy as np

from .. import __version__
from .. import _lib as L
from .. import runtime
from ..batch import pack_records
from ..utils import tr_harmonizer as trh
from ..utils import utils
from . import load_and_filter_genotypes

pval_precision = 2
BATCH_CELLS = 1 << 24     # loci x samples per device batch

_REASONS = {L.AS_NO_CALLED: 'No called samples', L.AS_ONE_ALLELE: 'Only one called allele',
            L.AS_N_COVARS: 'n covars >= n samples'}


def _merge_arrays(a, b):
    """``a`` left-outer-joined with ``b`` on the first (id) column (associaTR.py:22-54)."""
    assert len(a.shape) == 2 and len(b.shape) == 2
    assert len(set(a[:, 0]).intersection(b[:, 0])) > 0
    assert len(set(a[:, 0])) == a.shape[0]
    assert len(set(b[:, 0])) == b.shape[0]
    where = {key: row for row, key in enumerate(b[:, 0])}
    extra = np.full((a.shape[0], b.shape[1] - 1), np.nan)
    for row, key in enumerate(a[:, 0]):
        hit = where.get(key)
        if hit is not None:
            extra[row, :] = b[hit, 1:]
    return np.concatenate((a, extra), axis=1)


def _weighted_binom_conf(weights, successes, confidence):
    """Weighted Wilson interval (associaTR.py:56-103; kept for API compatibility, unused by the scan)."""
    import scipy.stats
    assert weights.shape == successes.shape
    assert len(weights.shape) == 1
    t = np.sum(weights)
    phat = np.dot(weights, successes) / t
    z = scipy.stats.norm.ppf(1 - confidence / 2)
    c = z * np.sqrt(np.dot(weights, weights))
    divisor = 2 + 2 * c ** 2 / t ** 2
    center = (2 * phat + c ** 2 / t ** 2) / divisor
    half = c / t * np.sqrt(4 * phat * (1 - phat) + c ** 2 / t ** 2) / divisor
    return (phat, center - half, center + half)


def _load_design(all_samples, trait_fnames, same_samples, sample_fname):
    """Covariate assembly of perform_gwas_helper (associaTR.py:138-204), messages included.
    Returns (sample_filter bool[S], covars [Sf, 1+T] with column 0 reserved for the genotype and
    column 1 the intercept, outcome [Sf], pheno_std)."""
    print('{} samples in the VCF'.format(len(all_samples)), flush=True)
    if not same_samples:
        covars = np.load(trait_fnames[0])
        if np.sum(np.isin(np.array(all_samples, dtype=float), covars[:, 0])) < 3:
            print(all_samples, covars[:, 0])
            print('Less than 3 samples matched between the covars array and the VCF. '
                  'Prehaps you meant to run with --same-samples? '
                  'Erroring out.')
            sys.exit(1)
        for trait_fname in trait_fnames[1:]:
            covars = _merge_arrays(covars, np.load(trait_fname))
        covars = _merge_arrays(np.array(all_samples, dtype=float).reshape(-1, 1), covars)
    else:
        arrays = []
        for trait_fname in trait_fnames:
            arrays.append(np.load(trait_fname))
            if not arrays[-1].shape[0] == len(all_samples):
                print("different number of samples in covariates file {trait_fname} than VCF, "
                      "and --same-samples was specified. Erroring out.")
                sys.exit(1)
        covars = np.hstack([np.full((arrays[0].shape[0], 1), -1), *arrays])

    if sample_fname:
        with open(sample_fname) as sample_file:
            sample_subset = [line.strip() for line in sample_file.readlines()]
        sample_filter = np.isin(all_samples, sample_subset)
        print(('{} samples remain after subsetting to samples '
               'from the file {}.\n'
               '{} samples from the sample file '
               'were not present in the VCF and were discarded.'
               ).format(np.sum(sample_filter), sample_fname, len(sample_subset) - np.sum(sample_filter)))
    else:
        sample_filter = np.array([True] * len(all_samples))

    before = sum(sample_filter)
    sample_filter = sample_filter & ~np.any(np.isnan(covars), axis=1)
    after = sum(sample_filter)
    print(('Removing {} samples which had missing '
           'phenotypes or covariates.\n'
           'Using {} for the regression.\n'
           'The number of samples used in each variant\'s regression will only be lower '
           'if that variant has missing calls.\n'
           ).format(before - after, after))

    covars = covars[sample_filter, :]
    pheno_std = np.std(covars[:, 1])
    covars = (covars - np.mean(covars, axis=0)) / np.std(covars, axis=0)
    outcome = covars[:, 1].copy()
    covars[:, 1] = 1
    return sample_filter, covars, outcome, pheno_std


def _device_vectors(sample_filter, covars, outcome, beagle_dosages=False):
    """[M, S] float64 for trk_assoc_params.vec: row 0 the outcome, rows 1.. the covariates; samples
    outside the regression set hold zeros (ignored on the device).  Up to 62 rows are scanned in one
    pass, up to 126 as pairs of row groups (trk.h TRK_ASSOC_MAX_VEC_WIDE); --beagle-dosages: pairs of row groups
    from 32 rows on."""
    S, M = len(sample_filter), covars.shape[1] - 1
    limit = L.ASSOC_MAX_VEC_WIDE
    if M > limit:
        raise ValueError("associaTR: %d trait columns (phenotype + covariates); this build scans at most %d"
                         % (M, limit))
    vec = np.zeros((M, S), dtype=np.float64)
    vec[0, sample_filter] = outcome
    for k in range(1, M):
        vec[k, sample_filter] = covars[:, 1 + k]
    return vec


def _r2(n2, sx, sxx, sy, syy, sxy):
    """np.corrcoef(x, y)[0, 1] ** 2 from the sums over n2 entries (nan when either side is constant)."""
    with np.errstate(all='ignore'):
        num = np.float64(n2) * sxy - np.float64(sx) * sy
        den = (np.float64(n2) * sxx - np.float64(sx) * sx) * (np.float64(n2) * syy - np.float64(sy) * sy)
        return np.float64(num * num) / den if den > 0 else np.float64(np.nan)


def _dosage_details(rec, n, cs, ls):
    """allele_frequency and the two imputation-quality columns of load_trs's dosage branch
    (load_and_filter_genotypes.py:179-227) from the device's per-class / per-locus sums."""
    lf_mod = load_and_filter_genotypes
    len_alleles = lf_mod.rounded_allele_lengths(rec)
    uniq = np.unique(len_alleles)
    with np.errstate(all='ignore'):
        allele_frequency = {u: np.float64(cs[k, 0]) / (2 * n) for k, u in enumerate(uniq)}
    rank = {float(u): k for k, u in enumerate(uniq)}
    r2 = {}
    for length in len_alleles:
        if length in r2:
            continue
        k = rank[float(length)]
        r2[length] = _r2(2 * n, cs[k, 2], cs[k, 2], cs[k, 0], cs[k, 1], cs[k, 3])
    if n > 0 and ls[6] == ls[7]:
        # every best-guess length is the same number x: np.corrcoef centres it with numpy's pairwise mean, which
        # differs from x by a rounding error unless x sums exactly -- the result is 0 (to ~1e-30) or nan
        x = np.float64(ls[6])
        length_r2 = np.float64(np.nan) if np.mean(np.full(2 * n, x)) == x else np.float64(0.0)
    else:
        length_r2 = _r2(ls[5], ls[0], ls[1], ls[2], ls[3], ls[4])
    extra = (lf_mod.dict_str(lf_mod.round_vals(r2, lf_mod.r2_precision)), str(round(length_r2, lf_mod.r2_precision)))
    return allele_frequency, extra


class _Facts:
    """What a row reads of a record: the harmonised position, the motif and the allele lengths (the batch pipeline builds
    these from the native harmoniser's tables; a TRRecord offers the same attributes)."""
    __slots__ = ('chrom', 'pos', 'motif', 'ref_allele_length', 'alt_allele_lengths')

    def __init__(self, chrom, pos, motif, lens):
        self.chrom, self.pos, self.motif = chrom, pos, motif
        self.ref_allele_length, self.alt_allele_lengths = lens[0], lens[1:]


# Memo tables of the row writer: the same few hundred allele lengths come back in every batch, and what the reference does
# with one of them -- python's round on a python float, numpy's on a numpy scalar, numpy's text of a float64 -- is done
# ONCE per distinct value, by the very call the per-record code makes (exact by construction), instead of per row.
_py_round, _np_round, _np_text = {}, {}, {}


def _rounded_py(x, precision):
    key = (x, precision)
    r = _py_round.get(key)
    if r is None:
        r = _py_round[key] = round(float(x), precision)
    return r


def _rounded_np(x, precision):
    key = (x, precision)
    r = _np_round.get(key)
    if r is None:
        r = _np_round[key] = round(np.float64(x), precision)
    return r


def _text_np(x):
    t = _np_text.get(x)
    if t is None:
        t = _np_text[x] = str(np.array([x], dtype=np.float64).astype(str)[0])
    return t


_key_text = {}


def _text_key(k):
    """str() of a rounded-length key (a numpy scalar), once per distinct value."""
    t = _key_text.get(k)
    if t is None:
        t = _key_text[k] = str(k)
    return t


def _gt_row_parts(rec, counts, lens):
    """(alleles column, allele_frequency dict, detail columns) of a GT-mode row: what np.unique(rounded lengths).astype(str),
    load_and_filter_genotypes.allele_frequency_from_counts and locus_details give for the record, without an array per
    record (reference load_and_filter_genotypes.py:157-227 through the per-record functions of this package, which the
    tests compare with this one)."""
    lf_mod = load_and_filter_genotypes
    prec = lf_mod.allele_len_precision
    alleles = ','.join(_text_np(v) for v in sorted({_rounded_py(x, prec) for x in lens}))
    # GetAlleleFreqs by length (ascending keys) over the tested samples' counts, then clean_len_alleles
    by_len = {}
    for x, c in zip(lens, counts):
        if c > 0:
            by_len[x] = by_len.get(x, 0) + c
    total = float(sum(by_len.values()))
    freq = {}
    for k in sorted(by_len):
        nk = _rounded_np(k, prec)
        v = by_len[k] / total
        freq[nk] = freq[nk] + v if nk in freq else v
    # dict_str of {rounded length: '{:.2g}' of its frequency}: keys ascending already, text of a key memoised; the general
    # function (quotes, brackets, NaN rewritten) only when a value could need it
    vals = ['{:.2g}'.format(val) for val in freq.values()]
    if any('n' in v for v in vals):                      # (nan / inf: never for counted alleles)
        afreq = lf_mod.dict_str(dict(zip(freq.keys(), vals)))
    else:
        afreq = '{' + ', '.join(['"%s": "%s"' % (_text_key(k), v) for k, v in zip(freq.keys(), vals)]) + '}'
    details = [rec.motif, str(len(rec.motif)), str(_rounded_py(rec.ref_allele_length, prec)), afreq]
    return alleles, freq, details


def _write_rows(outfile, recs, allele_off, allele_lens, res, pheno_std, non_major_cutoff, dosage=None):
    """The output rows of one device batch (associaTR.py:246-293).  ``recs``: TRRecords or _Facts; ``dosage``:
    (class_sums, locus_sums) of a --beagle-dosages batch."""
    lf_mod = load_and_filter_genotypes
    fmt = "{:." + str(pval_precision) + "e}\t{}\t{}\t{}\t"
    LI = res.locus_int
    # python floats from here on: the same IEEE arithmetic and the same shortest-digits text as numpy scalars, without a
    # numpy scalar object per number
    LF = np.asarray(res.locus_f64, dtype=np.float64).tolist()
    pheno_std = float(pheno_std)
    status_col, ntest_col = LI[:, L.AI_STATUS].tolist(), LI[:, L.AI_N_TESTED].tolist()
    counts_all = None if dosage is not None else res.allele_count.tolist()
    offs = [int(x) for x in allele_off]
    out = []
    for l, rec in enumerate(recs):
        o, e = offs[l], offs[l + 1]
        status = status_col[l]
        if dosage is not None:
            n = ntest_col[l]
            allele_frequency, extra = _dosage_details(rec, n, dosage[0][o:e], dosage[1][l])
            details = lf_mod.locus_details(rec, allele_frequency, extra)
            reason = lf_mod.filter_reason(allele_frequency, n, non_major_cutoff, True)
            if reason:                       # load_trs's filters come before the regression's own
                status = -1
            alleles = ','.join(list(np.unique(lf_mod.rounded_allele_lengths(rec)).astype(str)))
        else:
            alleles, _, details = _gt_row_parts(rec, counts_all[o:e], allele_lens[l])
            reason = None
        head = "{}\t{}\t{}\t{}\t".format(rec.chrom, rec.pos, alleles, ntest_col[l])
        if status == L.AS_OK:
            lf = LF[l]
            std = lf[L.AF_GT_STD]
            out.append(head + 'False\t' + fmt.format(lf[L.AF_PVAL], lf[L.AF_COEF] / std * pheno_std,
                                                     lf[L.AF_SE] / std * pheno_std, lf[L.AF_RSQUARED]) +
                       '\t'.join(details) + '\n')
        else:
            if status == -1:
                pass
            elif status == L.AS_NON_MAJOR:
                reason = 'non-major allele count<{}'.format(non_major_cutoff)
            elif status in _REASONS:
                reason = _REASONS[status]
            elif status == L.AS_ZERO_VARIANCE:
                # the reference divides 0/0 here and statsmodels raises on the empty design
                raise ValueError("locus %s:%s: the summed genotype is constant over the tested samples"
                                 % (rec.chrom, rec.pos))
            else:
                raise ValueError("locus %s:%s: genotype collinear with the covariates" % (rec.chrom, rec.pos))
            out.append(head + '{}\tnan\tnan\tnan\tnan\t'.format(reason) + '\t'.join(details) + '\n')
    outfile.write(''.join(out))
    if hasattr(outfile, 'flush'):
        outfile.flush()


def _flush(records, outfile, vec, sample_filter, pheno_std, non_major_cutoff, beagle_dosages=False):
    """One device batch of record objects -> output rows."""
    if not records:
        return
    lf_mod = load_and_filter_genotypes
    hb = pack_records(records)
    if beagle_dosages:
        from ..batch import stack_plane
        ap1 = stack_plane([np.asarray(r.format['AP1'], dtype=np.float32) for r in records], np.float32)
        ap2 = stack_plane([np.asarray(r.format['AP2'], dtype=np.float32) for r in records], np.float32)
        res, class_sums, locus_sums, _ = runtime.get_compute().assoc_dosage_batch(
            hb, vec, sample_filter, ap1, ap2, precision=lf_mod.allele_len_precision)
        _write_rows(outfile, records, hb.allele_off, hb.allele_lens, res, pheno_std, non_major_cutoff,
                    dosage=(class_sums, locus_sums))
    else:
        res = runtime.get_compute().assoc_batch(hb, vec, sample_filter, non_major_cutoff,
                                                precision=lf_mod.allele_len_precision)
        _write_rows(outfile, records, hb.allele_off, hb.allele_lens, res, pheno_std, non_major_cutoff)


# ---- the batch pipeline (round 6): native reader -> native batch harmoniser -> device scan -> rows, no object per record ----
# what the last perform_gwas call ran through (the tests and tools/e2e_assoc_only.py read it): 'batch', 'mixed' (some
# batches went through the record objects) or 'per-record'
LAST_RUN = {}


def _batch_path_ok(reader, vcftype, region, beagle_dosages, period_check):
    """GT-based runs over a file the native reader reads, of a caller the native harmoniser covers; --beagle-dosages (AP
    planes per record) and the hidden PERIOD check keep the per-record loop.  TRK_ASSOC_BATCH=0 (lab) forces it."""
    from .. import _knobs
    from ..vcfnative import NativeVCFReader, VT_CODES
    return (isinstance(reader, NativeVCFReader) and vcftype.name in VT_CODES and not beagle_dosages and not period_check and
            len(reader.samples) > 0 and _knobs.lab('TRK_ASSOC_BATCH', '1') != '0')


def _run_batches(reader, vcftype, region, shard, vec, sample_filter, pheno_std, non_major_cutoff, batch_loci):
    """perform_gwas_helper's loop (associaTR.py:246-293) a batch of records at a time.  Returns the number of loci."""
    from .. import _knobs
    from ..batch import HostBatch
    from ..synth import assoc_tables_from_classes
    compute = runtime.get_compute()
    lf_mod = load_and_filter_genotypes
    reader.use_buffers(getattr(compute, 'host_buffer', None), ring=2, release=getattr(compute, 'host_release', None))
    # the sample columns are parsed on the device (trk_parse_samples) and, for whole-file runs, the BGZF members inflated
    # there (statSTR's settings: the scan reads the genotype tensor where the parse kernel left it)
    device_parse = (_knobs.env('TRK_DEVICE_PARSE', '1') == '1' and hasattr(reader, 'device_parse') and
                    getattr(compute, 'eng', None) is not None and reader.device_parse(compute.eng))
    LAST_RUN['device_parse'] = bool(device_parse)
    LAST_RUN['device_inflate'] = bool(device_parse and not region and shard.world == 1 and
                                      _knobs.env('TRK_DEVICE_INFLATE', _knobs.DEVICE_INFLATE_DEFAULT['statSTR']) == '1' and
                                      hasattr(reader, 'device_inflate') and reader.device_inflate(compute.eng))
    reader.read_ahead(_knobs.env('TRK_VCF_READ_AHEAD', '1') == '1')
    region_start, region_done = None, False
    if region is not None:
        region_start = int(region.split(':')[1].split('-')[0]) if ':' in region else None
        reader(region)
    n_loci = 0
    last_rb = None
    while not region_done:
        if last_rb is not None:
            last_rb.release_device()
        rb = last_rb = reader.read_raw_batch(batch_loci)
        if rb.n == 0:
            break
        hz = rb.harmonize(vcftype.name)
        keep = None
        if region is not None:
            keep, region_done = reader.region_keep(rb, hz)
            if region_start is not None:
                keep = keep & (hz.pos >= region_start)       # load_trs skips the records that start before the region
            if not keep.any():
                continue
        LAST_RUN['batches'] += 1
        if hz.n_python or (keep is not None and not keep.all()):
            # something the native harmoniser does not cover, or a batch the region cuts: through the record objects
            if hz.n_python:
                LAST_RUN['fallback_batches'] += 1
                LAST_RUN['path'] = 'mixed'
            recs = [trh.HarmonizeRecord(vcftype, record) for l, record in enumerate(rb.records())
                    if keep is None or keep[l]]
            n_loci += len(recs)
            if shard.next_batch():
                _flush(recs, shard, vec, sample_filter, pheno_std, non_major_cutoff)
                shard.end_batch()
            continue
        n_loci += rb.n
        if not shard.next_batch():
            continue
        gt_in = rb.dev['gt'] if rb.dev is not None else rb.gt
        hb = HostBatch.from_tables(gt_in, rb.locus_ploidy, hz.allele_off, hz.len_class, hz.str_class,
                                   hz.len_class_value, lists=hz.lists)
        tables = assoc_tables_from_classes(hz.allele_off, hz.allele_len, hz.len_class_value, hz.n_len_classes,
                                           lf_mod.allele_len_precision)
        if getattr(compute, 'eng', None) is not None:
            res = compute.assoc_batch(hb, vec, sample_filter, non_major_cutoff, precision=lf_mod.allele_len_precision,
                                      tables=tables)
        else:                                     # (the tests' stand-in computes from the per-locus lists)
            res = compute.assoc_batch(hb, vec, sample_filter, non_major_cutoff, precision=lf_mod.allele_len_precision)
        if rb.dev is not None and gt_in is rb.dev['gt']:
            rb.dev['gt'] = None               # (the batch built from it freed the tensor with its other arrays)
            rb.release_device()
        off = hz.allele_off.tolist()
        alen = hz.allele_len.tolist()
        chroms, poss, motifs = rb.chrom_column(), hz.tr_pos.tolist(), rb.motifs(hz, vcftype.name)
        lens = [alen[off[l]:off[l + 1]] for l in range(rb.n)]
        facts = [_Facts(chroms[l], poss[l], motifs[l], lens[l]) for l in range(rb.n)]
        _write_rows(shard, facts, off, lens, res, pheno_std, non_major_cutoff)
        shard.end_batch()
    if last_rb is not None:
        last_rb.release_device()
    return n_loci


def perform_gwas_helper(outfile, all_samples, record_iter, phenotype_name, trait_fnames, same_samples, sample_fname,
                        non_major_cutoff, beagle_dosages=False, attach=None):
    """Header, covariates, batched scan (associaTR.py:117-372 without the plotting statistics)."""
    from .. import dist
    rank = dist.get_comm()[0]
    if rank == 0:
        outfile.write("chrom\tpos\talleles\tn_samples_tested\tlocus_filtered\tp_{}\tcoeff_{}\t".format(
            phenotype_name, phenotype_name))
        outfile.write('se_{}\tregression_R^2\t'.format(phenotype_name))
        outfile.flush()
    sample_filter, covars, outcome, pheno_std = _load_design(all_samples, trait_fnames, same_samples, sample_fname)
    vec = _device_vectors(sample_filter, covars, outcome, beagle_dosages)
    if rank == 0:
        fields = list(load_and_filter_genotypes.DETAIL_FIELDS)
        if beagle_dosages:
            fields.extend(load_and_filter_genotypes.DOSAGE_DETAIL_FIELDS)
        outfile.write('\t'.join(fields) + '\n')

    # one process per GPU (WORLD_SIZE > 1): every rank scans its contiguous share of the records (or, where the input
    # cannot be cut, batch b mod WORLD_SIZE); rank 0 writes the merged table (statSTR._ShardedOut)
    from ..statSTR.statSTR import _ShardedOut
    shard = _ShardedOut(outfile)
    if attach is not None:
        attach['shard'] = shard        # iter_records cuts the reader into contiguous shards when it starts
    batch_loci = max(1, min(4096, BATCH_CELLS // max(1, len(all_samples))))
    n_loci, start_time = 0, time.time()
    records = []

    def emit(recs):
        if recs and shard.next_batch():
            _flush(recs, shard, vec, sample_filter, pheno_std, non_major_cutoff, beagle_dosages)
            shard.end_batch()

    batch = attach.get('batch') if attach is not None else None
    LAST_RUN.clear()
    LAST_RUN.update(path='batch' if batch else 'per-record', batches=0, fallback_batches=0)
    if batch:
        reader, vcftype, region = batch
        shard.attach(reader, region)
        n_loci = _run_batches(reader, vcftype, region, shard, vec, sample_filter, pheno_std, non_major_cutoff, batch_loci)
    else:
        for trrecord in record_iter:
            records.append(trrecord)
            n_loci += 1
            if len(records) >= batch_loci:
                emit(records)
                records = []
        emit(records)
    shard.finish()
    total_time = time.time() - start_time
    if n_loci > 0:
        print("Done.\nTotal loci: {}\nTotal time: {}s\ntime/locus: {}s\n".format(
            n_loci, total_time, total_time / n_loci), flush=True)
    else:
        print("No variants found in the region being looked at\n", flush=True)


def perform_gwas(outfname, tr_vcf, phenotype_name, traits_fnames, vcftype, same_samples, sample_fname, region,
                 non_major_cutoff, beagle_dosages, plotting_phenotype_fname, paired_genotype_plot,
                 plot_phenotype_residuals, plotting_ci_alphas, imputed_ukb_strs_paper_period_check):
    """Signature of the reference (associaTR.py:432-482)."""
    if plotting_phenotype_fname or paired_genotype_plot or plot_phenotype_residuals or plotting_ci_alphas:
        raise NotImplementedError("the hidden --plotting-phenotype options of the reference are not provided")
    reader = utils.LoadSingleReader(tr_vcf, checkgz=False)
    if reader is None:
        raise ValueError("could not open %s" % tr_vcf)
    all_samples = reader.samples
    attach = {}
    inferred = trh.InferVCFType(reader, vcftype if vcftype else 'auto')
    if _batch_path_ok(reader, inferred, region, beagle_dosages, imputed_ukb_strs_paper_period_check):
        attach['batch'] = (reader, inferred, region)          # the batch pipeline reads `reader` itself
        record_iter = None
    else:
        record_iter = load_and_filter_genotypes.iter_records(
            tr_vcf, region, vcftype, beagle_dosages, imputed_ukb_strs_paper_period_check, attach=attach)
    from .. import dist
    rank = dist.get_comm()[0]
    temp = outfname + '.temp' if rank == 0 else outfname + '.rank%d.temp' % rank
    print("Writing output to {}.temp".format(outfname), flush=True)
    with open(temp, 'w') as outfile:
        perform_gwas_helper(outfile, all_samples, record_iter, phenotype_name, traits_fnames, same_samples,
                            sample_fname, non_major_cutoff, beagle_dosages, attach=attach)
    if rank != 0:
        import os
        os.remove(temp)               # only rank 0's file holds the merged table
        return
    print("Moving {}.temp to {}".format(outfname, outfname), flush=True)
    shutil.move(outfname + '.temp', outfname)
    print("Done.", flush=True)


def getargs():  # pragma: no cover
    parser = argparse.ArgumentParser(__doc__, formatter_class=utils.ArgumentDefaultsHelpFormatter)
    parser.add_argument('outfile')
    parser.add_argument('tr_vcf')
    parser.add_argument('phenotype_name', help='name of the phenotype being regressed against')
    parser.add_argument('traits', nargs='+',
                        help='At least one .npy 2d float array file of trait values per sample. The first trait of '
                             'the first file is the phenotype, every other column of every file a covariate. Without '
                             '--same-samples the first column of each file is the numeric sample ID and files are '
                             'joined on it; with --same-samples every array has one row per VCF sample, in VCF order, '
                             'and no ID column. Traits are standardised before the regression; coefficients and '
                             'standard errors are reported on the original scale.')
    parser.add_argument('--vcftype', choices=[str(item) for item in trh.VcfTypes.__members__],
                        help="Specify which caller produced the TR VCF, useful when the VCF is ambiguous "
                             "and the caller cannot be automatically inferred.")
    parser.add_argument('--same-samples', default=False, action='store_true', help='see the traits help string')
    parser.add_argument('--sample-list', help="File containing list of samples to use, one sample ID per line. "
                                              "If not specified, all samples are used.")
    parser.add_argument('--region', help="Restrict to \"chr:start-end\"")
    parser.add_argument('--non-major-cutoff', type=float, default=20,
                        help='Filter loci whose non-major-allele count (alleles coalesced by length) is below this '
                             'cutoff. Set to 0 to disable this filter.')
    parser.add_argument('--beagle-dosages', action='store_true', default=False,
                        help="regress against Beagle dosages from the AP{1,2} fields instead of from the GT field. "
                             "(The GP field is not supported)")
    parser.add_argument('--plotting-phenotype', help=argparse.SUPPRESS)
    parser.add_argument('--paired-genotype-plot', action='store_true', default=False, help=argparse.SUPPRESS)
    parser.add_argument('--plot-phenotype-residuals', action='store_true', default=False, help=argparse.SUPPRESS)
    parser.add_argument('--plotting-ci-alphas', type=float, nargs='*', default=[], help=argparse.SUPPRESS)
    parser.add_argument('--imputed-ukb-strs-paper-period-check', default=False, action='store_true',
                        help=argparse.SUPPRESS)
    parser.add_argument("--version", action="version", version='{}'.format(__version__))
    return parser.parse_args()


def main(args):
    today = datetime.datetime.now().strftime("%Y_%m_%d")
    print('-------Running AssociaTR (trtools v{}) ----------'.format(__version__))
    print("Run date: {}".format(today))
    print(args, flush=True)
    perform_gwas(args.outfile, args.tr_vcf, args.phenotype_name, args.traits, args.vcftype, args.same_samples,
                 args.sample_list, args.region, args.non_major_cutoff, args.beagle_dosages, args.plotting_phenotype,
                 args.paired_genotype_plot, args.plot_phenotype_residuals, args.plotting_ci_alphas,
                 args.imputed_ukb_strs_paper_period_check)


def run():  # pragma: no cover
    main(getargs())


if __name__ == '__main__':  # pragma: no cover
    run()
