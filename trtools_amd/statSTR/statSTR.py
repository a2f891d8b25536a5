"""statSTR: per-locus statistics of a TR VCF -- same command line, same
``main(args) -> int`` and same ``<out>.tab`` as the reference
(trtools/statSTR/statSTR.py), with the per-locus loop (statSTR.py:575-639)
replaced by batches of loci reduced on the GPU (trk_locus_stats).

Host Python parses / harmonises records and formats text; every statistic in
the table comes from the device (allele histograms + finaliser), for all sample
groups of a batch in one kernel pass.
"""
from .. import _knobs
from .._knobs import DEVICE_INFLATE_DEFAULT
import argparse
import os
import sys
import time

import numpy as np

from .. import __version__
from .. import _lib as L
from ..batch import pack_records
from ..utils import common, utils
from ..utils import tr_harmonizer as trh

BATCH_CELLS = 1 << 24     # loci x samples per device batch
MAX_GROUPS_PER_PASS = 8   # sample groups evaluated per kernel pass (trk.h)


def GetHeader(header, sample_prefixes):
    """Column names of one statistic (statSTR.py:82-102)."""
    if len(sample_prefixes) == 0:
        return [header]
    return [header + "-" + sp for sp in sample_prefixes]


# ---- the reference's per-record statistic functions (statSTR.py:31-80, 104-426) -----------------------------------------
# main() never calls them -- a batch of loci and every sample group is one kernel pass (_flush) -- but they are part of
# the module's surface: ``statSTR.GetHet(trrecord, sample_indexes=[...], uselength=...)`` answers for ONE record, one
# value per entry of ``sample_indexes`` (None: every sample).  Each is the reference's composition of TRRecord methods
# and utils functions; the record's histogram comes from the device (TRRecord._device_stats: a one-locus batch).
MAXPLOTS = 10


def _per_group(sample_indexes, fn):
    return [fn(si) for si in sample_indexes]


def PlotAlleleFreqs(trrecord, outprefix, sample_indexes=[None], sampleprefixes=None):
    """statSTR.py:31-80."""
    from . import plots
    if sample_indexes == [None]:
        sampleprefixes = ["sample"]
    plots.PlotAlleleFreqs(trrecord, outprefix, sample_indexes=sample_indexes, sampleprefixes=sampleprefixes)


def GetThresh(trrecord, sample_indexes=[None]):
    """Largest called allele length per sample group, nan without calls (statSTR.py:104-127)."""
    return _per_group(sample_indexes, lambda si: trrecord.GetMaxAllele(sample_index=si))


def GetAFreq(trrecord, sample_indexes=[None], count=False, uselength=True):
    """'allele:freq,...' (three decimals) or 'allele:count,...' per sample group, alleles ascending; '.' for a group
    without calls (statSTR.py:129-173)."""
    def one(si):
        table = (trrecord.GetAlleleCounts if count else trrecord.GetAlleleFreqs)(uselength=uselength, sample_index=si)
        if not table:
            return "."
        item = "%s:%i" if count else "%s:%.3f"
        return ",".join(item % (allele, table[allele]) for allele in sorted(table))
    return _per_group(sample_indexes, one)


def GetNAlleles(trrecord, sample_indexes=[None], nalleles_thresh=0.01, uselength=True):
    """Alleles whose frequency reaches ``nalleles_thresh`` (statSTR.py:175-209)."""
    def one(si):
        freqs = trrecord.GetAlleleFreqs(uselength=uselength, sample_index=si)
        return sum(1 for f in freqs.values() if f >= nalleles_thresh)
    return _per_group(sample_indexes, one)


def GetHWEP(trrecord, sample_indexes=[None], uselength=True):
    """Two-sided binomial test of the homozygote count against Hardy-Weinberg expectation (statSTR.py:211-252)."""
    def one(si):
        return utils.GetHardyWeinbergBinomialTest(trrecord.GetAlleleFreqs(sample_index=si, uselength=uselength),
                                                  trrecord.GetGenotypeCounts(sample_index=si, uselength=uselength))
    return _per_group(sample_indexes, one)


def GetHet(trrecord, sample_indexes=[None], uselength=True):
    """1 - sum p^2 (statSTR.py:254-291)."""
    return _per_group(sample_indexes, lambda si: utils.GetHeterozygosity(
        trrecord.GetAlleleFreqs(sample_index=si, uselength=uselength)))


def GetEntropy(trrecord, sample_indexes=[None], uselength=True):
    """Bit entropy of the allele distribution (statSTR.py:293-332)."""
    return _per_group(sample_indexes, lambda si: utils.GetEntropy(
        trrecord.GetAlleleFreqs(sample_index=si, uselength=uselength)))


def GetMean(trrecord, sample_indexes=[None], uselength=True):
    """Mean allele length -- by length whatever ``uselength`` says, as the reference (statSTR.py:334-358)."""
    return _per_group(sample_indexes, lambda si: utils.GetMean(trrecord.GetAlleleFreqs(sample_index=si, uselength=True)))


def GetMode(trrecord, sample_indexes=[None], uselength=True):
    """statSTR.py:360-383."""
    return _per_group(sample_indexes, lambda si: utils.GetMode(trrecord.GetAlleleFreqs(sample_index=si, uselength=True)))


def GetVariance(trrecord, sample_indexes=[None], uselength=True):
    """statSTR.py:385-409."""
    return _per_group(sample_indexes, lambda si: utils.GetVariance(trrecord.GetAlleleFreqs(sample_index=si, uselength=True)))


def GetNumSamples(trrecord, sample_indexes=[None]):
    """Samples with a full genotype: the genotype counts summed (statSTR.py:411-431)."""
    return _per_group(sample_indexes, lambda si: sum(trrecord.GetGenotypeCounts(sample_index=si).values()))


def format_nan_precision(precision_format, val):
    """statSTR.py:490-494."""
    if np.isnan(val):
        return "\tnan"
    return precision_format.format(val)


def getargs():  # pragma: no cover
    parser = argparse.ArgumentParser(__doc__, formatter_class=utils.ArgumentDefaultsHelpFormatter)
    io = parser.add_argument_group("Input/output")
    io.add_argument("--vcf", help="Input STR VCF file", type=str, required=True)
    io.add_argument("--out", help="Output file prefix. Use stdout to print file to standard output. In "
                    "addition, if not stdout then timing diagnostics are print to stdout.", type=str,
                    required=True)
    io.add_argument("--vcftype", help="Options=%s" % [str(i) for i in trh.VcfTypes.__members__], type=str,
                    default="auto")
    io.add_argument("--precision", help="How much precision to use when printing decimals", type=int, default=3)
    fg = parser.add_argument_group("Filtering group")
    fg.add_argument("--samples", help="File containing list of samples to include. Or a comma-separated list "
                    "of files to compute stats separate for each group of samples", type=str)
    fg.add_argument("--sample-prefixes", help="Prefixes to name output for each samples group. By default "
                    "uses 1,2,3 etc.", type=str)
    fg.add_argument("--region", help="Restrict to the region chrom:start-end. Requires file to bgzipped and "
                    "tabix indexed.", type=str)
    fg.add_argument("--only-passing", help="Only process records  where FILTER==PASS", action="store_true")
    name = "Stats group"
    sg = parser.add_argument_group(name)
    sg.add_argument("--thresh", help="Output threshold field (max allele size, used for GangSTR strinfo).",
                    action="store_true")
    sg.add_argument("--afreq", help="Output allele frequencies", action="store_true")
    sg.add_argument("--acount", help="Output allele counts", action="store_true")
    sg.add_argument("--nalleles", help="Output number of alleles with frequency exceeding a specified "
                    "threshold", action="store_true")
    sg.add_argument("--nalleles-thresh", help="The threshold for nalleles", type=float, default=0.01)
    sg.add_argument("--hwep", help="Output HWE p-values per loci.", action="store_true")
    sg.add_argument("--het", help="Output the heterozygosity of each locus.", action="store_true")
    sg.add_argument("--entropy", help="Output the entropy of each locus.", action="store_true")
    sg.add_argument("--mean", help="Output mean of the allele frequencies.", action="store_true")
    sg.add_argument("--mode", help="Output mode of the allele frequencies.", action="store_true")
    sg.add_argument("--var", help="Output variance of the allele frequencies.", action="store_true")
    sg.add_argument("--numcalled", help="Output number of samples called.", action="store_true")
    sg.add_argument("--use-length", help="Calculate per-locus stats (het, HWE) collapsing alleles by length. "
                    "This is implicitly true for genotypers which only emit length based genotypes.",
                    action="store_true")
    pg = parser.add_argument_group("Plotting group")
    pg.add_argument("--plot-afreq", help="Output allele frequency plot. Will only do for a maximum of 10 TRs.",
                    action="store_true")
    vg = parser.add_argument_group("Version")
    vg.add_argument("--version", action="version", version='{version}'.format(version=__version__))
    args = parser.parse_args()
    chosen = {}
    for grp in parser._action_groups:
        if grp.title == name:
            chosen = {a.dest: getattr(args, a.dest, None) for a in grp._group_actions}
    if not any(chosen.values()):
        common.WARNING("Error: Please use at least one of the flags in the Stats group. See statSTR --help "
                       "for options.")
        return None
    return args


class _RowFormatter:
    """Text of one output row from the device results of one batch."""

    def __init__(self, args, n_groups):
        self.args = args
        self.n_groups = n_groups
        self.pfmt = "\t{:." + str(args.precision) + "}"

    def afreq_text(self, hb, cnt, l, count):
        """statSTR.py:158-172: 'allele:freq' joined in sorted key order."""
        keys, ranks = hb.class_keys(l, self.args.use_length)
        o, e = int(hb.allele_off[l]), int(hb.allele_off[l + 1])
        cc = np.zeros(len(keys), dtype=np.int64)
        np.add.at(cc, ranks, cnt[o:e])
        present = [c for c in range(len(keys)) if cc[c]]
        if not present:
            return "."
        if count:
            return ",".join("%s:%i" % (keys[c], cc[c]) for c in present)
        total = float(cc.sum())
        return ",".join("%s:%.3f" % (keys[c], cc[c] / total) for c in present)

    def row(self, hb, st, l, rec, trrec):
        a = self.args
        ul = a.use_length
        G = self.n_groups
        I, F = st.locus_int, st.locus_f64
        out = [str(rec.CHROM), "\t", str(rec.POS), "\t", str(rec.POS + len(trrec.ref_allele))]
        if a.thresh:
            for g in range(G):
                out.append(format_nan_precision(self.pfmt, F[g, l, L.LF_THRESH]))
        if a.afreq:
            for g in range(G):
                out.append("\t" + self.afreq_text(hb, st.allele_count[g], l, False))
        if a.acount:
            for g in range(G):
                out.append("\t" + self.afreq_text(hb, st.allele_count[g], l, True))
        if a.nalleles:
            for g in range(G):
                out.append("\t" + str(int(I[g, l, L.LI_NALLELES_LEN if ul else L.LI_NALLELES_STR])))
        if a.hwep:
            for g in range(G):
                status = I[g, l, L.LI_HWE_STATUS_LEN if ul else L.LI_HWE_STATUS_STR]
                if status == L.HWE_VALUE_ERROR:
                    raise ValueError("binomtest: n must be a positive integer (no fully called genotype at "
                                     "{}:{})".format(rec.CHROM, rec.POS))
                if status == L.HWE_INDEX_ERROR:
                    raise IndexError("tuple index out of range (haploid genotypes at {}:{} have no HWE test)"
                                     .format(rec.CHROM, rec.POS))
                out.append(format_nan_precision(self.pfmt, F[g, l, L.LF_HWEP_LEN if ul else L.LF_HWEP_STR]))
        if a.het:
            for g in range(G):
                out.append(format_nan_precision(self.pfmt, F[g, l, L.LF_HET_LEN if ul else L.LF_HET_STR]))
        if a.entropy:
            for g in range(G):
                out.append(format_nan_precision(self.pfmt, F[g, l, L.LF_ENTROPY_LEN if ul else L.LF_ENTROPY_STR]))
        for flag, col in ((a.mean, L.LF_MEAN), (a.mode, L.LF_MODE), (a.var, L.LF_VAR)):
            if flag:
                for g in range(G):
                    out.append(format_nan_precision(self.pfmt, F[g, l, col]))
        if a.numcalled:
            for g in range(G):
                out.append("\t" + str(int(I[g, l, L.LI_N_CALLED])))
        out.append("\n")
        return "".join(out)


class _ShardedOut:
    """Under a one-process-per-GPU launcher (WORLD_SIZE > 1) every rank keeps the text of its batches and rank 0
    writes the merged table.  Who owns what: when the reader can be cut (``attach``: the native reader on a bgzipped
    or plain-text file, no region query) each rank reads ONLY its contiguous share of the records -- SURVEY 8(e)'s
    contiguous locus ranges: a rank inflates and parses 1 / WORLD_SIZE of the file -- and every batch it reads is
    its own; otherwise every rank reads everything and batch b belongs to rank b mod WORLD_SIZE.  Either way the
    parts are merged in record order.  With one process it is a pass-through to the output file."""

    def __init__(self, outf):
        from .. import dist
        self.rank, self.world, self.comm = dist.get_comm()
        self.outf = outf
        self.parts = []
        self.batch_no = -1
        self._cur = None
        self.contiguous = False

    def attach(self, reader, region=None):
        """Cut the input into contiguous shards if the reader allows it (call before the first read)."""
        if self.world > 1 and not region:
            fn = getattr(reader, 'shard', None)
            self.contiguous = bool(fn and fn(self.rank, self.world))
        return self.contiguous

    def next_batch(self):
        self.batch_no += 1
        mine = self.world == 1 or self.contiguous or self.batch_no % self.world == self.rank
        self._cur = [] if (mine and self.world > 1) else None
        return mine

    def write(self, text):
        if self._cur is None:
            self.outf.write(text)
        else:
            self._cur.append(text)

    def end_batch(self):
        if self._cur is not None:
            key = (self.rank << 40) + self.batch_no if self.contiguous else self.batch_no
            self.parts.append((key, ''.join(self._cur).encode()))
            self._cur = None

    def finish(self):
        if self.world > 1:
            from .. import dist
            merged = dist.merge_parts(self.parts, self.comm)
            if self.rank == 0:
                self.outf.write(merged.decode())


def _flush(batch, group_masks, fmt, outf, nalleles_thresh):
    """Reduce one batch of (variant, TRRecord) pairs on the device and write its rows."""
    from .. import runtime
    if not batch:
        return
    recs = [t for _, t in batch]
    compute = runtime.get_compute()
    masks = group_masks if group_masks[0] is not None else None
    if masks is None or len(masks) <= MAX_GROUPS_PER_PASS:
        hb = pack_records(recs, masks)
        st = compute.locus_stats(hb, nalleles_thresh=nalleles_thresh)
    else:
        # more than 8 strata: several passes over the same genotype tensor
        parts = []
        for i in range(0, len(masks), MAX_GROUPS_PER_PASS):
            hb = pack_records(recs, masks[i:i + MAX_GROUPS_PER_PASS])
            parts.append(compute.locus_stats(hb, nalleles_thresh=nalleles_thresh))
        from ..compute import StatsHost
        st = StatsHost(np.concatenate([p.allele_count for p in parts]),
                       np.concatenate([p.locus_int for p in parts]),
                       np.concatenate([p.locus_f64 for p in parts]))
    if np.any(st.locus_int[:, :, L.LI_N_BAD]):
        l = int(np.argwhere(st.locus_int[:, :, L.LI_N_BAD])[0][1])
        raise IndexError("genotype index out of range at {}:{}".format(batch[l][0].CHROM, batch[l][0].POS))
    for l, (rec, trrec) in enumerate(batch):
        outf.write(fmt.row(hb, st, l, rec, trrec))


def _batch_path_ok(args, invcf, vcftype):
    """The batch pipeline (native reader -> native batch harmoniser -> device -> native row formatter: no Python
    object per record) covers the callers whose records carry allele SEQUENCES, with or without a --region query
    (index seek, then batches cut at the region's end); everything else takes the per-record loop below.
    --plot-afreq does not change the path: its eleven plots come from a short read of their own (_plot_first).
    TRK_STATSTR_BATCH=0 forces the per-record loop."""
    from ..vcfnative import NativeVCFReader, VT_CODES
    return (isinstance(invcf, NativeVCFReader) and vcftype.name in VT_CODES and
            len(invcf.samples) > 0 and _knobs.lab('TRK_STATSTR_BATCH', '1') != '0')


def _plot_first(args, vcftype, group_masks, sample_prefixes):
    """--plot-afreq (statSTR.py:603-607): allele-frequency plots of the first eleven records the run reports
    (``num_plotted <= 10``), from a reader of their own -- the table itself stays on the batch pipeline instead of
    sending the whole file through record objects for the sake of eleven plots."""
    from .plots import PlotAlleleFreqs
    rd = utils.LoadSingleReader(args.vcf, checkgz=args.region is not None)
    if rd is None:
        return
    try:
        num_plotted = 0
        for record in (rd(args.region) if args.region else rd):
            trrecord = trh.HarmonizeRecord(vcftype, record)
            if args.only_passing and record.FILTER is not None:
                continue
            PlotAlleleFreqs(trrecord, args.out, sample_indexes=group_masks, sampleprefixes=sample_prefixes)
            num_plotted += 1
            if num_plotted > 10:
                break
    finally:
        if hasattr(rd, 'close'):
            rd.close()


# what the last main() call ran through (bench.py's end-to-end extra and the tests read it): 'batch' = the batch
# pipeline for every batch, 'mixed' = some batches went through the record objects, 'per-record' = the loop
LAST_RUN = {}


def _run_batches(args, invcf, vcftype, group_masks, fmt, shard, batch_loci, start_time):
    """statSTR.py:575-639 a batch at a time.  Returns the number of records read."""
    from .. import runtime
    from ..batch import HostBatch
    from ..vcfnative import SS_FLAGS
    compute = runtime.get_compute()
    flags = 0
    for name, bit in SS_FLAGS.items():
        if getattr(args, name):
            flags |= bit
    masks = group_masks if group_masks[0] is not None else None
    passes = [(None, 1)]                     # (group bits, groups) per device pass: up to eight strata each
    if masks is not None:
        passes = []
        for i in range(0, len(masks), MAX_GROUPS_PER_PASS):
            part = masks[i:i + MAX_GROUPS_PER_PASS]
            gb = np.zeros(len(invcf.samples), dtype=np.uint8)
            for g, m in enumerate(part):
                gb |= (np.asarray(m, dtype=bool).astype(np.uint8) << g)
            passes.append((gb, len(part)))
    # Sample groups (one pass of <= 8): the reader lays the genotype columns out by sample CLASS while it parses them
    # (trk_vcf_set_sample_map) -- every class a column range the ungrouped count kernel streams, no gather on the
    # device, no per-call group look-ups (TRK_CLASS_SORT=0: the grouped kernel on file-order columns)
    layout = None
    # (round 6) ... unless the sample columns are parsed on the device: with one pass of groups the grouped count kernel
    # reads the tensor where the parse kernel left it, in file order (a command line's batches are a few thousand records
    # -- launch latency either way -- and the host parse it replaces is most of the run: 0.17-0.18 -> 0.08 s per GB with two
    # groups, tools/e2e_stat_groups.py); more than eight groups (several passes over one tensor) keep the host parse
    groups_on_device = (masks is not None and len(passes) == 1 and _knobs.env('TRK_DEVICE_PARSE', '1') == '1' and
                        hasattr(invcf, 'device_parse') and getattr(compute, 'eng', None) is not None and
                        _knobs.lab('TRK_GROUPS_DEVICE_PARSE', '1') != '0')
    if (masks is not None and len(passes) == 1 and not groups_on_device and hasattr(invcf, 'set_sample_map') and
            _knobs.lab('TRK_CLASS_SORT', '1') != '0' and getattr(compute, 'supports_class_layout', False)):
        from ..engine import class_layout
        layout = class_layout(passes[0][0], passes[0][1], row_align=32 if len(invcf.samples) >= 512 else 4)
        if layout is not None:
            invcf.set_sample_map(layout['col_of'], layout['n_out'])
    invcf.use_buffers(getattr(compute, 'host_buffer', None), ring=2, release=getattr(compute, 'host_release', None))
    # The sample columns are parsed on the device (trk_parse_samples, round 4; TRK_DEVICE_PARSE=0: on the host): the reader
    # stops at the FORMAT keys, the batch's text goes over PCIe instead of the genotype tensor.  Ungrouped runs on the
    # device engine only.
    device_parse = ((masks is None or groups_on_device) and _knobs.env('TRK_DEVICE_PARSE', '1') == '1' and
                    hasattr(invcf, 'device_parse') and getattr(compute, 'eng', None) is not None and
                    invcf.device_parse(compute.eng))
    LAST_RUN['device_parse'] = bool(device_parse)
    # ... and the file's BGZF members are inflated there too (trk_inflate_blocks, round 5; TRK_DEVICE_INFLATE=0: by the
    # reader's threads): the compressed bytes cross PCIe, the host sees the newlines and the heads of the lines.  Whole-file
    # runs only (a region seeks).
    LAST_RUN['device_inflate'] = bool(device_parse and not args.region and _knobs.env('TRK_DEVICE_INFLATE', DEVICE_INFLATE_DEFAULT['statSTR']) == '1' and
                                      hasattr(invcf, 'device_inflate') and invcf.device_inflate(compute.eng))
    # Batch n + 1 is read while batch n is counted and written (TRK_VCF_READ_AHEAD=0: off; it was off while the command line
    # was bound by CPU seconds on the 16-CPU grant of the GPU boxes: with the parse on the device 0.18 -> 0.125 s per GB,
    # profiles/r04_notes.md section 15)
    invcf.read_ahead(_knobs.env('TRK_VCF_READ_AHEAD', '1') == '1')
    nrecords = 0
    region_done = False
    LAST_RUN.update(path='batch', batches=0, fallback_batches=0)
    if args.region:
        invcf(args.region)                   # statSTR.py:568-570: seek to the region's first index window
    last_rb = None
    while not region_done:
        if last_rb is not None:
            last_rb.release_device()         # (a device-parsed batch that went another way than the device's)
        rb = last_rb = invcf.read_raw_batch(batch_loci)
        if rb.n == 0:
            break
        hz = rb.harmonize(vcftype.name)
        keep = None
        if args.region:
            keep, region_done = invcf.region_keep(rb, hz)
            if not keep.any():
                continue
        nrecords += rb.n if keep is None else int(keep.sum())
        LAST_RUN['batches'] += 1
        if hz.n_python:
            LAST_RUN['fallback_batches'] += 1
            LAST_RUN['path'] = 'mixed'
            # something the native harmoniser does not cover: this batch goes through the Python objects (and
            # raises the reference's errors where the reference does)
            batch = []
            for l, record in enumerate(rb.records()):
                if keep is not None and not keep[l]:
                    continue
                trrecord = trh.HarmonizeRecord(vcftype, record)
                if args.only_passing and record.FILTER is not None:
                    continue
                batch.append((record, trrecord))
            if shard.next_batch():
                _flush(batch, group_masks, fmt, shard, args.nalleles_thresh)
                shard.end_batch()
            continue
        if not shard.next_batch():
            continue
        parts = []
        for gb, ng in passes:
            sorted_ok = (layout is not None and rb.gt_mapped is not None and rb.gt.shape[2] == 2 and
                         bool(np.all(np.asarray(rb.locus_ploidy) == 2)))
            gt_in = rb.gt_mapped if sorted_ok else (rb.dev['gt'] if rb.dev is not None else rb.gt)
            hb = HostBatch.from_tables(gt_in, rb.locus_ploidy, hz.allele_off, hz.len_class,
                                       hz.str_class, hz.len_class_value, layout['bits'] if sorted_ok else gb, ng,
                                       lists=hz.lists)
            if sorted_ok:
                hb.class_layout = layout
            parts.append(compute.locus_stats(hb, nalleles_thresh=args.nalleles_thresh))
            if rb.dev is not None and gt_in is rb.dev['gt']:
                rb.dev['gt'] = None           # (the batch built from it freed the tensor with its other arrays)
                rb.release_device()
        st = parts[0]
        if len(parts) > 1:                   # more than eight strata: the passes' rows side by side
            from ..compute import StatsHost
            st = StatsHost(np.concatenate([p.allele_count for p in parts]),
                           np.concatenate([p.locus_int for p in parts]),
                           np.concatenate([p.locus_f64 for p in parts]))
        skip = (hz.passing == 0) if args.only_passing else None
        if keep is not None:
            skip = ~keep if skip is None else (skip | ~keep)
        text, el, ek = rb.statstr_rows(st, flags, args.precision, args.use_length, skip)
        if ek:
            chrom, pos = rb.chrom_pos(el)
            if ek == 1:
                raise ValueError("binomtest: n must be a positive integer (no fully called genotype at "
                                 "{}:{})".format(chrom, pos))
            if ek == 2:
                raise IndexError("tuple index out of range (haploid genotypes at {}:{} have no HWE test)"
                                 .format(chrom, pos))
            raise IndexError("genotype index out of range at {}:{}".format(chrom, pos))
        shard.write(text.decode())
        shard.end_batch()
        if args.out != "stdout" and shard.rank == 0:
            print("Finished {} records, time/record={:.5}sec".format(
                nrecords, (time.time() - start_time) / nrecords), flush=True, end="\r")
    if last_rb is not None:
        last_rb.release_device()
    return nrecords


def main(args):
    if not os.path.exists(args.vcf):
        common.WARNING("Error: %s does not exist" % args.vcf)
        return 1
    if not os.path.exists(os.path.dirname(os.path.abspath(args.out))):
        common.WARNING("Error: The directory which contains the output location {} does"
                       " not exist".format(args.out))
        return 1
    if os.path.isdir(args.out) and args.out.endswith(os.sep):
        common.WARNING("Error: The output location {} is a directory".format(args.out))
        return 1

    invcf = utils.LoadSingleReader(args.vcf, checkgz=args.region is not None)
    if invcf is None:
        return 1
    vcftype = trh.VcfTypes[args.vcftype] if args.vcftype != 'auto' else trh.InferVCFType(invcf)

    # sample groups (statSTR.py:520-542)
    sample_prefixes, group_masks = [], [None]
    if args.samples:
        all_samples = np.array(invcf.samples)
        sfiles = args.samples.split(",")
        sample_prefixes = args.sample_prefixes.split(",") if args.sample_prefixes \
            else [str(i) for i in range(1, len(sfiles) + 1)]
        if len(sfiles) != len(sample_prefixes):
            common.WARNING("--sample-prefixes must be same length as --samples")
            return 1
        group_masks = []
        for sf in sfiles:
            with open(sf, "r") as fh:
                wanted = np.array([line.strip() for line in fh.readlines()])
            mask = np.isin(all_samples, wanted)
            if not np.any(mask):
                common.WARNING("No samples from {} found in the VCF file".format(sf))
                return 1
            group_masks.append(mask)

    header = ["chrom", "start", "end"]
    for flag, name in ((args.thresh, "thresh"), (args.afreq, "afreq"), (args.acount, "acount"),
                       (args.nalleles, "nalleles"), (args.hwep, "hwep"), (args.het, "het"),
                       (args.entropy, "entropy"), (args.mean, "mean"), (args.mode, "mode"),
                       (args.var, "var"), (args.numcalled, "numcalled")):
        if flag:
            header.extend(GetHeader(name, sample_prefixes))

    fmt = _RowFormatter(args, len(group_masks))
    outf = None
    try:
        if args.out == "stdout":
            if args.plot_afreq:
                common.WARNING("Cannot use --out stdout when generating plots")
                return 1
            outf = sys.stdout
        else:
            # only rank 0 writes the table: the other ranks of a sharded run must not even open it (mode "w" would
            # truncate what rank 0 has flushed by then)
            from .. import dist as _dist
            outf = open(args.out + ".tab" if _dist.get_comm()[0] == 0 else os.devnull, "w")
        shard = _ShardedOut(outf)
        if shard.rank == 0:
            outf.write("\t".join(header) + "\n")
        shard.attach(invcf, args.region)
        if args.plot_afreq and shard.rank == 0:
            _plot_first(args, vcftype, group_masks, sample_prefixes)
        region = invcf(args.region) if args.region else invcf
        n_samples = max(len(invcf.samples), 1)
        batch_loci = max(1, min(4096, BATCH_CELLS // n_samples))
        start_time = time.time()
        nrecords = 0
        batch = []
        LAST_RUN.clear()
        LAST_RUN.update(path='per-record', batches=0, fallback_batches=0)
        if _batch_path_ok(args, invcf, vcftype) and \
                _run_batches(args, invcf, vcftype, group_masks, fmt, shard, batch_loci, start_time) is not None:
            region = ()
        else:
            LAST_RUN['path'] = 'per-record'
        for record in region:
            nrecords += 1
            trrecord = trh.HarmonizeRecord(vcftype, record)
            if args.only_passing and record.FILTER is not None:
                continue
            batch.append((record, trrecord))
            if len(batch) >= batch_loci:
                if shard.next_batch():
                    _flush(batch, group_masks, fmt, shard, args.nalleles_thresh)
                    shard.end_batch()
                batch = []
                outf.flush()
                if args.out != "stdout" and shard.rank == 0:
                    print("Finished {} records, time/record={:.5}sec".format(
                        nrecords, (time.time() - start_time) / nrecords), flush=True, end="\r")
        if batch and shard.next_batch():
            _flush(batch, group_masks, fmt, shard, args.nalleles_thresh)
            shard.end_batch()
        shard.finish()
    finally:
        if outf is not None and args.out != "stdout":
            outf.close()
        if hasattr(invcf, 'close'):
            invcf.close()            # hands the reader's pinned staging buffers back to the pool
    if args.out != "stdout":
        print("\nDone", flush=True)
    return 0


def run():  # pragma: no cover
    args = getargs()
    if args is None:
        sys.exit(1)
    sys.exit(main(args))


if __name__ == "__main__":  # pragma: no cover
    run()
