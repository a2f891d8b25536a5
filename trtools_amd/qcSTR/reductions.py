"""The numbers behind qcSTR's plots, computed a batch at a time on the device.

Mirrors the accumulation of the reference's main loop (trtools/qcSTR/qcSTR.py:441-575, 613-660): what it passes to
``OutputSampleCallrate`` / ``OutputChromCallrate`` / ``OutputQualityPerSample`` / ``OutputQualityPerLocus`` /
``OutputDiffRefHistogram`` / ``OutputDiffRefBias``.  Per record the reference does a handful of numpy reductions
over the samples; here a batch of records goes through ONE device pass (``trk_qc_reduce`` for calls and quality,
``trk_locus_stats`` for the allele counts by length) and the host only adds up batch results."""
import numpy as np

from ..utils import common
from ..utils import tr_harmonizer as trh
from ..utils import utils


def _flush(compute, batch, sample_index, use_q, ignore, acc):
    from ..batch import pack_records
    recs = [r for _, r in batch]
    subset = not bool(np.all(sample_index))
    hb = pack_records(recs, group_masks=[sample_index] if subset else None)
    q = None
    if use_q:
        q = np.stack([np.asarray(r.GetQualityScores(), dtype=np.float32).reshape(-1) for r in recs])
    res = compute.qc_batch(hb, q, sample_index if subset else None, ignore)
    acc['sample_calls'] += res['sample_calls'][sample_index]
    for (chrom, _), c in zip(batch, res['locus_calls']):
        acc['chrom_calls'][chrom] = acc['chrom_calls'].get(chrom, 0) + int(c)
    if use_q:
        acc['per_sample_total'] += res['sample_qual_sum'][sample_index]
        with np.errstate(invalid='ignore', divide='ignore'):
            acc['per_locus'].extend((res['locus_qual_sum'] / res['locus_qual_n']).tolist())
    # allele counts by length over the selected samples (qcSTR.py:529-530, 563-568)
    st = compute.locus_stats(hb)
    cnt = st.allele_count[0]
    for l, r in enumerate(recs):
        lo, hi = int(hb.allele_off[l]), int(hb.allele_off[l + 1])
        lens = np.asarray([r.ref_allele_length] + list(r.alt_allele_lengths), dtype=np.float64)
        c = cnt[lo:hi].astype(np.int64)
        period = len(r.motif)
        diff_unit = lens - r.ref_allele_length
        # the joint distribution the two diff-from-reference plots are drawn from (qcSTR.py:238-327: the histogram of
        # diff_unit; per reference-length bin the mean / median of diff_bp, bins with fewer than mingts calls
        # dropped): (reference length in bp, period, difference in repeat units) -> allele calls.  Small (a few
        # entries per locus) and enough to rebuild either plot's data exactly.
        hist = acc['diff_hist']
        reflen_bp = float(r.ref_allele_length * period)
        for d, k in zip(diff_unit.tolist(), c.tolist()):
            if k:
                key = (reflen_bp, period, d)
                hist[key] = hist.get(key, 0) + k
        acc['n_alleles'] += int(c.sum())
        acc['sum_diff_unit'] += float((diff_unit * c).sum())
        acc['sum_diff_bp'] += float((diff_unit * period * c).sum())
        acc['sum_reflen_bp'] += float(r.ref_allele_length * period * c.sum())


LAST_RUN = {}          # which road the last call took: path ('batch' | 'per-record' | 'mixed'), batches, fallback_batches

_QUALITY_KEY = {'hipstr': 'Q', 'longtr': 'Q', 'gangstr': 'Q', 'advntr': 'ML'}      # TRRecord.quality_field per caller


def _batch_road_ok(invcf, vcftype):
    """A file the native reader reads, of a caller the native batch harmoniser covers (TRK_QC_BATCH=0, lab: the record
    objects)."""
    from .. import _knobs
    from ..vcfnative import NativeVCFReader, VT_CODES
    return (isinstance(invcf, NativeVCFReader) and vcftype.name in VT_CODES and len(invcf.samples) > 0 and
            _knobs.lab('TRK_QC_BATCH', '1') != '0')


def _flush_tables(compute, rb, hz, motifs, q, sample_index, use_q, ignore, acc):
    """``_flush`` for a batch the native harmoniser covered: the same device passes over the batch's tables, the same
    accumulation -- per locus the very expressions of ``_flush`` on slices of the batch's arrays, no record objects."""
    from ..batch import HostBatch
    subset = not bool(np.all(sample_index))
    hb = HostBatch.from_tables(rb.gt, rb.locus_ploidy, hz.allele_off, hz.len_class, hz.str_class, hz.len_class_value,
                               group_bits=sample_index.astype(np.uint8) if subset else None, n_groups=1, lists=hz.lists)
    res = compute.qc_batch(hb, q, sample_index if subset else None, ignore)
    acc['sample_calls'] += res['sample_calls'][sample_index]
    for chrom, c in zip(rb.chrom_column(), res['locus_calls'].tolist()):
        acc['chrom_calls'][chrom] = acc['chrom_calls'].get(chrom, 0) + int(c)
    if use_q:
        acc['per_sample_total'] += res['sample_qual_sum'][sample_index]
        with np.errstate(invalid='ignore', divide='ignore'):
            acc['per_locus'].extend((res['locus_qual_sum'] / res['locus_qual_n']).tolist())
    st = compute.locus_stats(hb)
    cnt = np.asarray(st.allele_count[0]).astype(np.int64)
    off = hb.allele_off.tolist()
    alen = np.asarray(hz.allele_len, dtype=np.float64)
    hist = acc['diff_hist']
    for l in range(rb.n):
        lo, hi = off[l], off[l + 1]
        lens, c = alen[lo:hi], cnt[lo:hi]
        ref_len = lens[0]
        period = len(motifs[l])
        diff_unit = lens - ref_len
        reflen_bp = float(ref_len * period)
        for d, k in zip(diff_unit.tolist(), c.tolist()):
            if k:
                key = (reflen_bp, period, d)
                hist[key] = hist.get(key, 0) + k
        acc['n_alleles'] += int(c.sum())
        acc['sum_diff_unit'] += float((diff_unit * c).sum())
        acc['sum_diff_bp'] += float((diff_unit * period * c).sum())
        acc['sum_reflen_bp'] += float(ref_len * period * c.sum())


def _run_batches(compute, invcf, vcftype, sample_index, use_q, ignore, period, numrecords, batch_loci, acc):
    """The main loop a batch of records at a time: native reader -> native batch harmoniser -> the device passes.  A batch
    the harmoniser leaves records of to this side, one that --period cuts, or one with a record whose FORMAT lacks the
    quality field goes through the record objects (``_flush``).  Returns the number of records taken."""
    qkey = _QUALITY_KEY.get(vcftype.name) if use_q else None
    if qkey is not None:
        invcf.select_format(qkey)
    invcf.read_ahead(False)            # (numrecords may end the run inside a batch: nothing is read beyond what is asked for)
    n = 0
    while numrecords is None or n < numrecords:
        want = batch_loci if numrecords is None else min(batch_loci, numrecords - n)
        rb = invcf.read_raw_batch(want)
        if rb.n == 0:
            break
        hz = rb.harmonize(vcftype.name)
        LAST_RUN['batches'] += 1
        motifs = None if hz.n_python else rb.motifs(hz, vcftype.name)
        whole = (not hz.n_python and (period is None or all(len(m) == period for m in motifs)) and
                 (qkey is None or all(qkey in cols for cols in rb.format_columns())))
        if not whole:
            LAST_RUN['fallback_batches'] += 1
            LAST_RUN['path'] = 'mixed'
            batch = []
            for record in rb.records():
                tr = trh.HarmonizeRecord(vcftype, record)
                if period is not None and len(tr.motif) != period:
                    continue
                acc['chrom_calls'].setdefault(tr.chrom, 0)
                batch.append((tr.chrom, tr))
            n += len(batch)
            if batch:
                _flush(compute, batch, sample_index, use_q, ignore, acc)
            continue
        for chrom in rb.chroms():
            acc['chrom_calls'].setdefault(chrom, 0)
        q = None
        if use_q:
            q = np.ascontiguousarray(np.asarray(rb.planes[qkey], dtype=np.float32).reshape(rb.n, -1))
        _flush_tables(compute, rb, hz, motifs, q, sample_index, use_q, ignore, acc)
        n += rb.n
    return n


def qc_reductions(vcf, vcftype='auto', samples=None, period=None, quality=(), quality_ignore_no_call=False,
                  numrecords=None, batch_loci=1024):
    """Returns a dict: samples (the selected names), sample_calls, chrom_calls, numrecords, per_sample_quality and
    per_locus_quality (None without a quality request), the sums behind the two diff-from-reference plots
    (n_alleles, sum_diff_unit, sum_diff_bp, sum_reflen_bp) and ``diff_ref_histogram``: {(reference length in bp,
    period, difference from the reference in repeat units): allele calls} -- the joint distribution from which
    OutputDiffRefHistogram's histogram and OutputDiffRefBias's per-bin mean / median (with its mingts rule) can be
    rebuilt; the per-call quality matrix of the sample-stratified / per-call plots is not kept (plotting is out of
    scope).  Per-locus quality means are float64 sums / n here where the reference takes
    np.mean over float32: equal to float32 rounding (stated in tests/test_gpu_qc.py).  Arguments as the reference's command line
    (qcSTR.py:343-419); returns None where the reference returns 1."""
    from .. import runtime
    compute = runtime.get_compute()
    invcf = utils.LoadSingleReader(vcf, checkgz=False)
    if invcf is None:
        return None
    harmonizer = trh.TRRecordHarmonizer(invcf, vcftype) if vcftype != 'auto' else trh.TRRecordHarmonizer(invcf)
    quality = list(quality)
    if len(quality) > 0 and not harmonizer.HasQualityScore():
        common.WARNING("Requested a quality plot, but the input vcf doesn't have quality scores!")
        return None
    if samples:
        wanted = [item.strip() for item in open(samples, "r").readlines()]
        sample_index = np.isin(np.array(invcf.samples), wanted)
        sample_list = list(np.array(invcf.samples)[sample_index])
    else:
        sample_list = list(invcf.samples)
        sample_index = np.ones(len(sample_list), dtype=bool)
    if len(quality) == 0 and harmonizer.HasQualityScore():   # the default quality plot (qcSTR.py:472-479)
        quality = ['sample-stratified'] if len(sample_list) <= 5 else ['per-locus']
    use_q = len(quality) != 0
    acc = dict(sample_calls=np.zeros(len(sample_list)), chrom_calls={}, per_sample_total=np.zeros(len(sample_list)),
               per_locus=[], n_alleles=0, sum_diff_unit=0.0, sum_diff_bp=0.0, sum_reflen_bp=0.0, diff_hist={})
    batch, n = [], 0
    LAST_RUN.clear()
    LAST_RUN.update(path='per-record', batches=0, fallback_batches=0)
    if _batch_road_ok(invcf, harmonizer.vcftype):
        LAST_RUN['path'] = 'batch'
        n = _run_batches(compute, invcf, harmonizer.vcftype, sample_index, use_q, quality_ignore_no_call, period, numrecords,
                         batch_loci, acc)
        harmonizer = ()
    for trrecord in harmonizer:
        if numrecords is not None and n >= numrecords:
            break
        if period is not None and len(trrecord.motif) != period:
            continue
        acc['chrom_calls'].setdefault(trrecord.chrom, 0)
        batch.append((trrecord.chrom, trrecord))
        n += 1
        if len(batch) >= batch_loci:
            _flush(compute, batch, sample_index, use_q, quality_ignore_no_call, acc)
            batch = []
    if batch:
        _flush(compute, batch, sample_index, use_q, quality_ignore_no_call, acc)
    per_sample = None
    if use_q:
        with np.errstate(invalid='ignore', divide='ignore'):
            per_sample = acc['per_sample_total'] / (acc['sample_calls'] if quality_ignore_no_call else n)
    return dict(samples=sample_list, sample_calls=acc['sample_calls'], chrom_calls=acc['chrom_calls'], numrecords=n,
                per_sample_quality=per_sample, per_locus_quality=acc['per_locus'] if use_q else None,
                n_alleles=acc['n_alleles'], sum_diff_unit=acc['sum_diff_unit'], sum_diff_bp=acc['sum_diff_bp'],
                sum_reflen_bp=acc['sum_reflen_bp'], diff_ref_histogram=acc['diff_hist'])
