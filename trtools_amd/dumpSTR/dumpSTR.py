"""dumpSTR: call-level and locus-level filtering of TR VCFs -- same command line,
``main(args) -> int``, ``.vcf`` / ``.samplog.tab`` / ``.loclog.tab`` outputs as
the reference (trtools/dumpSTR/dumpSTR.py), with the per-record loop
(dumpSTR.py:1270-1338) replaced by batches of loci processed on the GPU:

  k_call_filter   every call-level predicate, the FORMAT/FILTER bit mask, the masked
                  genotypes and the per-sample counters of ``sample_info``
  k_locus_count / k_locus_finalize   allele histograms + statistics of the masked genotypes
  k_locus_filter  locus filter decisions and the ``loc_info`` counters

Host Python parses / harmonises records, pre-parses string FORMAT fields,
evaluates the string-only locus filters (HRUN, BED regions) and writes text.
"""
from .. import _knobs
from .._knobs import DEVICE_INFLATE_DEFAULT
import argparse
import collections
import itertools
import os
import time
import subprocess as sp
import sys

import numpy as np

from .. import __version__
from .. import _lib as L
from .. import vcfio
from ..batch import pack_records, stack_plane
from ..utils import common, utils
from ..utils import tr_harmonizer as trh
from . import filters

_NOCALL_INT_FORMAT_VAL = -2147483648
BATCH_CELLS = 1 << 23


def MakeWriter(outfile, invcf, command):
    """VCF writer whose header records the dumpSTR command (dumpSTR.py:24-46)."""
    invcf.add_to_header("##command-DumpSTR=" + command)
    return vcfio.VCFWriter(outfile, invcf)


# ---------------------------------------------------------------------------
# argument validation (dumpSTR.py:48-521), table driven
# ---------------------------------------------------------------------------
# (argument, rule, needed FORMAT fields); rule: '01' value in [0,1], '>=0' non-negative
_CALLER_RULES = collections.OrderedDict([
    ('hipstr', (trh.VcfTypes.hipstr, "HipSTR", [
        ('hipstr_max_call_flank_indel', '01', ('DP', 'DFLANKINDEL')), ('hipstr_max_call_stutter', '01', ('DP', 'DSTUTTER')),
        ('hipstr_min_supp_reads', '>=0', ('ALLREADS', 'GB')), ('hipstr_min_call_DP', '>=0', ('DP',)),
        ('hipstr_max_call_DP', '>=0', ('DP',)), ('hipstr_min_call_Q', '01', ('Q',))],
        [('hipstr_min_call_DP', 'hipstr_max_call_DP')])),
    ('longtr', (trh.VcfTypes.longtr, "LongTR", [
        ('longtr_max_call_flank_indel', '01', ('DP', 'DFLANKINDEL')), ('longtr_min_supp_reads', '>=0', ('ALLREADS', 'GB')),
        ('longtr_min_call_DP', '>=0', ('DP',)), ('longtr_max_call_DP', '>=0', ('DP',)),
        ('longtr_min_call_Q', '01', ('Q',))],
        [('longtr_min_call_DP', 'longtr_max_call_DP')])),
    ('gangstr', (trh.VcfTypes.gangstr, "GangSTR", [
        ('gangstr_min_call_DP', '>=0', ('DP',)), ('gangstr_max_call_DP', '>=0', ('DP',)),
        ('gangstr_min_call_Q', '01', ('Q',)), ('gangstr_expansion_prob_het', '01', ('QEXP',)),
        ('gangstr_expansion_prob_hom', '01', ('QEXP',)), ('gangstr_expansion_prob_total', '01', ('QEXP',))],
        [('gangstr_min_call_DP', 'gangstr_max_call_DP')])),
    ('advntr', (trh.VcfTypes.advntr, "adVNTR", [
        ('advntr_min_call_DP', '>=0', ('DP',)), ('advntr_max_call_DP', '>=0', ('DP',)),
        ('advntr_min_spanning', '>=0 ', ('SR',)), ('advntr_min_flanking', '>=0 ', ('FR',)),
        ('advntr_min_ML', '>=0', ('ML',))],
        [('advntr_min_call_DP', 'advntr_max_call_DP')])),
    ('eh', (trh.VcfTypes.eh, "ExpansionHunter", [
        ('eh_min_ADFL', '>=0', ('ADFL',)), ('eh_min_ADIR', '>=0', ('ADIR',)), ('eh_min_ADSP', '>=0', ('ADSP',)),
        ('eh_min_call_LC', '>=0', ('LC',)), ('eh_max_call_LC', '>=0', ('LC',))],
        [('eh_min_call_LC', 'eh_max_call_LC')])),
    ('popstr', (trh.VcfTypes.popstr, "popSTR", [
        ('popstr_min_call_DP', '>=0', ('DP',)), ('popstr_max_call_DP', '>=0', ('DP',)),
        ('popstr_require_support', '>=0', ('AD',))],
        [('popstr_min_call_DP', 'popstr_max_call_DP')])),
])
_GANGSTR_FLAGS = ('gangstr_filter_span_only', 'gangstr_filter_spanbound_only', 'gangstr_filter_badCI')
_BEAGLE_NAMES = {"adVNTR": "AdVNTR"}


def _flag(arg):
    """argparse destination -> command-line spelling (case of DP/Q/... preserved)."""
    return "--" + arg.replace("_", "-")


def CheckLocusFilters(args, vcftype, is_beagle):
    """dumpSTR.py:48-99."""
    if args.min_locus_callrate is not None and is_beagle:
        common.WARNING("--min-locus-callrate cannot be applied to Beagle imputed files at the moment "
                       "as there are currently no call level Beagle filters")
        return False
    for arg in ('min_locus_hwep', 'min_locus_het', 'max_locus_het'):
        v = getattr(args, arg)
        if v is not None and (v < 0 or v > 1):
            common.WARNING("Invalid {}. Must be between 0 and 1".format(_flag(arg)))
            return False
    if args.min_locus_het is not None and args.max_locus_het is not None \
            and args.max_locus_het < args.min_locus_het:
        common.WARNING("Cannot have --max-locus-het less than --min-locus-het")
        return False
    hip_like = vcftype in (trh.VcfTypes.hipstr, trh.VcfTypes.longtr)
    if args.use_length and not hip_like:
        common.WARNING("--use-length is only meaningful for HipSTR or LongTR, which report sequence level "
                       "differences.")
    if args.filter_hrun and not hip_like:
        common.WARNING("--filter-hrun only relevant to HipSTR or LongTR files. This filter will have no effect.")
    if args.filter_regions is not None and args.filter_regions_names is not None:
        if len(args.filter_regions_names.split(",")) != len(args.filter_regions.split(",")):
            common.WARNING("Length of --filter-regions-names must match --filter-regions.")
            return False
    return True


def _check_caller(format_fields, args, rules, pairs):
    for arg, rule, fields in rules:
        v = getattr(args, arg, None)
        if v is None:
            continue
        if rule == '01' and (v < 0 or v > 1):
            common.WARNING("{} must be between 0 and 1".format(_flag(arg)))
            return False
        if rule.startswith('>=0') and v < 0:
            common.WARNING("{} must be {}".format(_flag(arg), ">=0" if rule.endswith(' ') else ">= 0"))
            return False
        for f in fields:
            assert f in format_fields
    for lo, hi in pairs:
        a, b = getattr(args, lo, None), getattr(args, hi, None)
        if a is not None and b is not None and b < a:
            common.WARNING("{} must be >= {}".format(_flag(hi), _flag(lo)))
            return False
    return True


def _caller_check(key, lines):
    def check(format_fields, args):
        _, _, rules, pairs = _CALLER_RULES[key]
        return _check_caller(format_fields, args, rules, pairs)
    check.__doc__ = ("Are the %s call-level options in range, and does the VCF carry the FORMAT fields they read?  "
                     "False after a WARNING otherwise (dumpSTR.py:%s)." % (_CALLER_RULES[key][1], lines))
    return check


# the reference's per-caller validators under their own names (CheckFilters below walks the same table)
CheckHipSTRFilters = _caller_check('hipstr', '101-151')
CheckLongTRFilters = _caller_check('longtr', '153-198')
CheckGangSTRFilters = _caller_check('gangstr', '200-263')
CheckAdVNTRFilters = _caller_check('advntr', '265-310')
CheckEHFilters = _caller_check('eh', '312-357')
CheckPopSTRFilters = _caller_check('popstr', '359-394')


def CheckFilters(format_fields, args, vcftype, is_beagle):
    """Validate the user's filters against the VCF's caller (dumpSTR.py:396-521)."""
    if not CheckLocusFilters(args, vcftype, is_beagle):
        return False
    for _, (vt, label, rules, pairs) in _CALLER_RULES.items():
        used = any(getattr(args, arg, None) is not None for arg, _, _ in rules)
        if label == "GangSTR":
            used = used or any(getattr(args, f, False) for f in _GANGSTR_FLAGS)
        if not used:
            continue
        if vcftype != vt:
            common.WARNING("{} options can only be applied to {} VCFs".format(label, label))
            return False
        if is_beagle and label != "popSTR":
            common.WARNING("{} call level filters cannot be applied to Beagle VCFs".format(
                _BEAGLE_NAMES.get(label, label)))
            return False
        if not _check_caller(format_fields, args, rules, pairs):
            return False
    return True


# ---------------------------------------------------------------------------
# logs (dumpSTR.py:523-588)
# ---------------------------------------------------------------------------

def WriteLocLog(loc_info, fname):
    keys = list(loc_info.keys())
    assert "totalcalls" in keys and "PASS" in keys
    keys.remove("totalcalls")
    callrate = 0 if loc_info["PASS"] == 0 else float(loc_info["totalcalls"]) / loc_info["PASS"]
    with open(fname, "w") as f:
        f.write("MeanSamplesPerPassingSTR\t%s\n" % callrate)
        for k in keys:
            f.write("FILTER:%s\t%s\n" % (k, loc_info[k]))
    return True


def WriteSampLog(sample_info, sample_names, fname):
    header = ["sample"] + list(sample_info.keys())
    header[header.index('totaldp')] = 'meanDP'
    # (columns as Python lists first: str() of a Python int / float prints what str() of the numpy scalar prints, at a
    # fifth of the cost -- 5000 samples x 6 columns were 12 ms of a 0.22 s run)
    def plain(col):
        a = np.asarray(col)
        # (only 64-bit columns print alike as Python numbers; anything else keeps its numpy scalars)
        return a.tolist() if a.dtype in (np.int64, np.float64) else list(a)
    numcalls_l = plain(sample_info["numcalls"])
    totaldp_l = plain(sample_info["totaldp"])
    rest = [plain(c) for c in itertools.islice(sample_info.values(), 2, None)]
    lines = ["\t".join(header)]
    for i, s in enumerate(sample_names):
        numcalls = numcalls_l[i]
        cols = [s, str(numcalls), str(totaldp_l[i] * 1.0 / numcalls) if numcalls > 0 else "0"]
        cols.extend(str(c[i]) for c in rest)
        lines.append("\t".join(cols))
    with open(fname, "w") as f:
        f.write("\n".join(lines) + "\n")


def GetAllCallFilters(call_filters):
    return [f.name for f in call_filters]


# ---------------------------------------------------------------------------
# filter lists (dumpSTR.py:777-915)
# ---------------------------------------------------------------------------

def BuildCallFilters(args):
    """Call filters in the reference's fixed order (= samplog column order)."""
    F = filters
    out = []

    def add(arg, make):
        v = getattr(args, arg, None)
        if v is not None and v is not False:
            out.append(make(v))
    add('hipstr_max_call_flank_indel', lambda v: F.HipSTRCallFlankIndels(v))
    add('hipstr_max_call_stutter', lambda v: F.HipSTRCallStutter(v))
    add('hipstr_min_supp_reads', lambda v: F.HipSTRCallMinSuppReads(v))
    add('hipstr_min_call_DP', lambda v: F.CallFilterMinValue("HipSTRCallMinDepth", "DP", v))
    add('hipstr_max_call_DP', lambda v: F.CallFilterMaxValue("HipSTRCallMaxDepth", "DP", v))
    add('hipstr_min_call_Q', lambda v: F.CallFilterMinValue("HipSTRCallMinQ", "Q", v))
    add('longtr_max_call_flank_indel', lambda v: F.HipSTRCallFlankIndels(v, rename="LongTRCallFlankIndels"))
    add('longtr_min_supp_reads', lambda v: F.HipSTRCallMinSuppReads(v, rename="LongTRMinSuppReads"))
    add('longtr_min_call_DP', lambda v: F.CallFilterMinValue("LongTRCallMinDepth", "DP", v))
    add('longtr_max_call_DP', lambda v: F.CallFilterMaxValue("LongTRCallMaxDepth", "DP", v))
    add('longtr_min_call_Q', lambda v: F.CallFilterMinValue("LongTRCallMinQ", "Q", v))
    add('gangstr_min_call_DP', lambda v: F.CallFilterMinValue("GangSTRCallMinDepth", "DP", v))
    add('gangstr_max_call_DP', lambda v: F.CallFilterMaxValue("GangSTRCallMaxDepth", "DP", v))
    add('gangstr_min_call_Q', lambda v: F.CallFilterMinValue("GangSTRCallMinQ", "Q", v))
    add('gangstr_expansion_prob_het', lambda v: F.GangSTRCallExpansionProbHet(v))
    add('gangstr_expansion_prob_hom', lambda v: F.GangSTRCallExpansionProbHom(v))
    add('gangstr_expansion_prob_total', lambda v: F.GangSTRCallExpansionProbTotal(v))
    add('gangstr_filter_span_only', lambda v: F.GangSTRCallSpanOnly())
    add('gangstr_filter_spanbound_only', lambda v: F.GangSTRCallSpanBoundOnly())
    add('gangstr_filter_badCI', lambda v: F.GangSTRCallBadCI())
    add('advntr_min_call_DP', lambda v: F.CallFilterMinValue("AdVNTRCallMinDepth", "DP", v))
    add('advntr_max_call_DP', lambda v: F.CallFilterMaxValue("AdVNTRCallMaxDepth", "DP", v))
    add('advntr_min_spanning', lambda v: F.CallFilterMinValue("AdVNTRCallMinSpanning", "SR", v))
    add('advntr_min_flanking', lambda v: F.CallFilterMinValue("AdVNTRCallMinFlanking", "FR", v))
    add('advntr_min_ML', lambda v: F.CallFilterMinValue("AdVNTRCallMinML", "ML", v))
    add('eh_min_call_LC', lambda v: F.CallFilterMinValue("EHCallMinDepth", "LC", v))
    add('eh_max_call_LC', lambda v: F.CallFilterMaxValue("EHCallMaxDepth", "LC", v))
    add('eh_min_ADFL', lambda v: F.CallFilterMinValue("EHCallMinADFL", "ADFL", v))
    add('eh_min_ADIR', lambda v: F.CallFilterMinValue("EHCallMinADFL", "ADIR", v))
    add('eh_min_ADSP', lambda v: F.CallFilterMinValue("EHCallMinADSP", "ADSP", v))
    add('popstr_min_call_DP', lambda v: F.CallFilterMinValue("PopSTRMinCallDepth", "DP", v))
    add('popstr_max_call_DP', lambda v: F.CallFilterMaxValue("PopSTRMaxCallDepth", "DP", v))
    add('popstr_require_support', lambda v: F.PopSTRCallRequireSupport(v))
    return out


def BuildLocusFilters(args):
    """Locus filters in the reference's order (dumpSTR.py:875-915)."""
    out = []
    if args.min_locus_callrate is not None:
        out.append(filters.Filter_MinLocusCallrate(args.min_locus_callrate))
    if args.min_locus_hwep is not None:
        out.append(filters.Filter_MinLocusHWEP(args.min_locus_hwep, args.use_length))
    if args.min_locus_het is not None:
        out.append(filters.Filter_MinLocusHet(args.min_locus_het, args.use_length))
    if args.max_locus_het is not None:
        out.append(filters.Filter_MaxLocusHet(args.max_locus_het, args.use_length))
    if args.filter_hrun:
        out.append(filters.Filter_LocusHrun())
    if args.filter_regions is not None:
        files = args.filter_regions.split(",")
        names = args.filter_regions_names.split(",") if args.filter_regions_names is not None \
            else ['FILTER' + str(i) for i in range(len(files))]
        for nm, fn in zip(names, files):
            rf = filters.create_region_filter(nm, fn)
            if rf is None:
                raise ValueError('Could not load regions file: {}'.format(fn))
            out.append(rf)
    return out


# ---------------------------------------------------------------------------
# batch evaluation
# ---------------------------------------------------------------------------

# what the last main() call ran through: 'batch' (every batch through the batch pipeline), 'mixed' (some batches
# through the record objects), 'per-record' (the loop); read by bench.py's end-to-end extra and the tests
LAST_RUN = {}


def _tick(phase, seconds):
    """Wall time of the batch pipeline's phases, summed over the batches of a run (LAST_RUN['seconds'])."""
    d = LAST_RUN.setdefault('seconds', {})
    d[phase] = d.get(phase, 0.0) + seconds


class _Planes:
    """FORMAT planes a set of call filters needs, stacked over a batch of records."""
    MIN_WIDTH = {'QEXP': 3, 'REPCN': 2, '__repci': 4, '__rc': 4}      # columns the GangSTR filters index (filters.py:573-757)

    def __init__(self, records, call_filters, want_dp=True):
        self.keys, builders = [], {}
        for f in call_filters:
            for key, build in f.planes():
                if key not in builders:
                    builders[key] = build
                    self.keys.append(key)
        self.dp_key = None
        if want_dp and records:
            fmt = records[0].format
            for cand in ('DP', 'LC'):          # dumpSTR.py:688-695
                if cand in fmt:
                    self.dp_key = cand
                    if cand not in builders:
                        builders[cand] = filters._field(cand)[1]
                        self.keys.append(cand)
                    break
        self.index = {k: i for i, k in enumerate(self.keys)}
        self.arrays = []
        for k in self.keys:
            per = [np.asarray(builders[k](r)) for r in records]
            per = [a.reshape(a.shape[0], -1) for a in per]
            kind = per[0].dtype.kind if per else 'i'
            if kind == 'f':
                plane = stack_plane([a.astype(np.float32) for a in per], np.float32)
            elif kind in 'iu':
                plane = stack_plane([a.astype(np.int32) for a in per], np.int32)
            else:
                raise ValueError("Found an unexpected format dtype for format field " + k)
            # a record without a single call carries ONE missing value per sample whatever the field's Number (htslib,
            # cyvcf2, vcfio alike); the reference's filters return before they index such an array (filters.py:598-600).
            # A batch of nothing but such records -- the per-record API's one-record batches -- still gets a plane of the
            # field's width: missing values in the columns the filters name
            need = self.MIN_WIDTH.get(k, 1)
            if plane.shape[2] < need:
                wide = np.full(plane.shape[:2] + (need,), np.nan if plane.dtype.kind == 'f' else _NOCALL_INT_FORMAT_VAL,
                               dtype=plane.dtype)
                wide[:, :, :plane.shape[2]] = plane
                plane = wide
            self.arrays.append(plane)

    def get(self, key):
        return self.arrays[self.index[key]]

    @property
    def dp_plane(self):
        return -1 if self.dp_key is None else self.index[self.dp_key]


def _filter_values(f, planes, hb, l):
    """float64[S]: what the reference writes after '<filter name>_' for a fired call."""
    if isinstance(f, filters.PopSTRCallRequireSupport):
        ad = planes.get('__ad')[l]
        out = np.full(ad.shape[0], np.nan)
        rows = np.arange(ad.shape[0])
        for j in range(int(hb.locus_ploidy[l])):          # last offending haplotype wins
            idx = hb.gt[l][:, j].astype(int)
            idx = np.where(idx < 0, idx + ad.shape[1], idx)
            ok = (idx >= 0) & (idx < ad.shape[1])
            v = np.where(ok, ad[rows, np.clip(idx, 0, ad.shape[1] - 1)], f.threshold)
            hit = ok & (v < f.threshold)
            out[hit] = v[hit]
        return out
    return f.value(planes.get, l)


def evaluate_call_filters(records, call_filters):
    """Device evaluation of call filters on records; returns (mask [L,S], values[k][l] float64[S])."""
    from .. import runtime
    hb = pack_records(records)
    planes = _Planes(records, call_filters, want_dp=False)
    specs = [f.spec(planes.index) for f in call_filters]
    ch, _, _, _ = runtime.get_compute().dumpstr_batch(hb, planes.arrays, specs, -1, {})
    values = [[_filter_values(f, planes, hb, l) for l in range(hb.n_loci)] for f in call_filters]
    return ch.mask, values


def _call_filter_text(mask_row, names, values_l):
    """FORMAT/FILTER strings of one record (dumpSTR.py:648-683)."""
    return vcfio.CallFilterColumn(mask_row, names, values_l).to_list()


def _null_filtered(vcfrecord, filtered, ploidy):
    """Mask the genotype and every other FORMAT field of filtered calls (dumpSTR.py:721-746)."""
    g = np.array(vcfrecord.genotype.array())
    g[filtered, :ploidy] = -1
    g[filtered, -1] = 0
    vcfrecord.set_gt_array(g)
    for field in list(vcfrecord.FORMAT):
        if field in ('GT', 'FILTER'):
            continue
        vals = np.array(vcfrecord.format(field))
        kind = vals.dtype.kind
        if kind in 'US':
            vals[filtered] = '.'      # fixed-width arrays hold at least one character
        elif kind == 'f':
            vals[filtered] = np.nan
        elif kind == 'i':
            vals[filtered] = _NOCALL_INT_FORMAT_VAL
        else:
            raise ValueError("Found an unexpected format dtype for format field " + field)
        vcfrecord.set_format(field, vals)


def ApplyCallFilters(record, call_filters, sample_info, sample_names):
    """The reference's per-record entry (dumpSTR.py:613-774) with its arguments, side effects and return value: the
    FORMAT/FILTER text of every sample is set on ``record.vcfrecord``, ``sample_info`` (filter name -> int[S],
    'numcalls', 'totaldp') is updated, filtered calls lose their genotype and every other FORMAT value, and a TRRecord
    over the modified variant comes back.  main() does not call it -- a batch of records is one pass of the
    call-filter kernel (_Run) -- but the decisions here come from that same kernel: the record is a one-locus batch."""
    from .. import runtime
    hb = pack_records([record])
    planes = _Planes([record], call_filters)
    specs = [f.spec(planes.index) for f in call_filters]
    nothing = dict(min_callrate=None, min_hwep=None, min_het=None, max_het=None, use_length=False, n_extern=0,
                   extern_bits=None)
    ch, _, _, _ = runtime.get_compute().dumpstr_batch(hb, planes.arrays, specs, planes.dp_plane, nothing)
    if ch.error[0]:
        bad = np.zeros(record.GetNumSamples(), dtype=bool)
        bad[int(ch.error[2])] = True
        raise ValueError("The following samples have calls but negative DP values at chromosome {} pos {}: {}"
                         .format(record.chrom, record.pos, str(np.asarray(sample_names)[bad])))
    sample_info['numcalls'] += ch.sample_counters[0]
    for k, f in enumerate(call_filters):
        sample_info[f.name] += ch.sample_counters[1 + k]
    if planes.dp_plane >= 0:
        sample_info['totaldp'] += ch.totaldp
        sample_info['totaldp'][ch.dp_missing > 0] = np.nan
    else:
        sample_info['totaldp'][:] = np.nan
    v, mrow = record.vcfrecord, ch.mask[0]
    vals = [_filter_values(f, planes, hb, 0) if np.any((mrow >> np.uint32(k)) & 1) else None
            for k, f in enumerate(call_filters)]
    v.set_format('FILTER', vcfio.CallFilterColumn(mrow, [f.name for f in call_filters], vals))
    filtered = ((mrow & np.uint32(0x7fffffff)) != 0) & ((mrow & np.uint32(L.TRK_MASK_NOCALL)) == 0)
    if not np.any(filtered):
        return record
    _null_filtered(v, filtered, record.GetMaxPloidy())
    # a record of its own over the modified variant; alleles that were given by length stay so (dumpSTR.py:748-773)
    by_len_alt, by_len_ref = record.HasFabricatedAltAlleles(), record.HasFabricatedRefAllele()
    return trh.TRRecord(v, None if by_len_ref else record.ref_allele, None if by_len_alt else record.alt_alleles,
                        record.motif, record.record_id, record.quality_field, harmonized_pos=record.pos,
                        full_alleles=record.full_alleles,
                        ref_allele_length=record.ref_allele_length if by_len_ref else None,
                        alt_allele_lengths=record.alt_allele_lengths if by_len_alt else None,
                        quality_score_transform=record.quality_score_transform)


def ApplyLocusFilters(record, locus_filters, loc_info, drop_filtered):
    """The reference's per-record entry (dumpSTR.py:917-973): every filter is asked about the record, ``loc_info``
    counts the ones that fire (and 'NO_CALLS_REMAINING', 'PASS', 'totalcalls'), the variant's FILTER column is set
    unless filtered records are dropped; True when the locus is filtered.  ``record`` is a TRRecord (statistics from the
    device) or anything that offers the methods the filters read."""
    fired = []
    for filt in locus_filters:
        if filt(record) is not None:
            loc_info[filt.filter_name()] += 1
            fired.append(filt.filter_name())
    n_called = np.sum(record.GetCalledSamples())
    if n_called == 0:
        loc_info['NO_CALLS_REMAINING'] += 1
        fired.append('NO_CALLS_REMAINING')
    if not drop_filtered:
        record.vcfrecord.FILTER = ';'.join(fired) if fired else 'PASS'
    if not fired:
        loc_info['PASS'] += 1
        loc_info['totalcalls'] += n_called
    return bool(fired)


class _Run:
    """State of one dumpSTR run: filters, counters, writer."""

    def __init__(self, args, invcf, call_filters, locus_filters, outvcf):
        self.args, self.invcf, self.outvcf = args, invcf, outvcf
        self.call_filters, self.locus_filters = call_filters, locus_filters
        n = len(invcf.samples)
        self.sample_names = np.array(invcf.samples)
        self.sample_info = collections.OrderedDict()
        self.sample_info['numcalls'] = np.zeros(n, dtype=int)
        self.sample_info['totaldp'] = np.zeros(n, dtype=float)
        for nm in GetAllCallFilters(call_filters):
            self.sample_info[nm] = np.zeros(n, dtype=int)
        self.loc_info = collections.OrderedDict([("totalcalls", 0), ("PASS", 0), ("NO_CALLS_REMAINING", 0)])
        for f in locus_filters:
            self.loc_info[f.filter_name()] = 0
        # device bit of every locus filter: fixed bits for the statistic filters,
        # extern bits (in order) for the string filters
        self.host_filters = [f for f in locus_filters if f.bit is None]
        self.bit_of = {}
        for f in locus_filters:
            self.bit_of[id(f)] = f.bit if f.bit is not None \
                else L.LOCF_EXTERN0 + self.host_filters.index(f)
        self.spec = dict(min_callrate=args.min_locus_callrate, min_hwep=args.min_locus_hwep,
                         min_het=args.min_locus_het, max_het=args.max_locus_het,
                         use_length=bool(args.use_length), n_extern=len(self.host_filters))
        # locus sharding (one process per GPU): each rank reads its contiguous share of the records where the reader
        # can be cut (native reader on a bgzipped / plain-text file: trk_vcf_shard), else batch b belongs to rank
        # b mod WORLD_SIZE
        from .. import dist
        self.rank, self.world, self.comm = dist.get_comm()
        self.batch_no = -1
        self.parts = []
        self.no_dp = False
        self.contiguous = False
        if self.world > 1 and args.num_records is None:
            fn = getattr(invcf, 'shard', None)
            self.contiguous = bool(fn and fn(self.rank, self.world))

    def _mine(self):
        return self.world == 1 or self.contiguous or self.batch_no % self.world == self.rank

    def _key(self):
        return (self.rank << 40) + self.batch_no if self.contiguous else self.batch_no

    def emit(self, variant):
        if self.world == 1:
            self.outvcf.write_record(variant)
        else:
            self._cur.append(str(variant))

    def finish(self):
        """Cohort-wide counters and the merged record stream (rank 0 writes)."""
        if self.world == 1:
            return
        from .. import dist
        flag = self.comm.allreduce_sum_i64(np.array([1 if self.no_dp else 0], dtype=np.int64))
        self.sample_info = dist.reduce_sample_info(self.sample_info, self.comm)
        if flag[0] > 0:
            self.sample_info['totaldp'][:] = np.nan
        self.loc_info = dist.reduce_loc_info(self.loc_info, self.comm)
        # contigs the header does not declare: every rank's, in rank order (== file order for contiguous shards)
        seen = []
        mine = dist.pack_frames([str(c).encode() for c in self.invcf.contigs_seen])
        for blob in self.comm.allgather_bytes(np.frombuffer(mine, dtype=np.uint8)):
            for c in dist.unpack_frames(blob.tobytes()):
                c = c.decode()
                if c not in seen:
                    seen.append(c)
        self.invcf.contigs_seen[:] = seen
        merged = dist.merge_parts(self.parts, self.comm)
        if self.rank == 0:
            self.outvcf.write_text(merged.decode())

    def process(self, records):
        from .. import runtime
        if not records:
            return
        self.batch_no += 1
        if not self._mine():
            return
        self._cur = []
        # the depth field (DP, else LC, else none) is looked up in EVERY record (dumpSTR.py:688-695): a chunk is cut
        # into runs of records that agree on it, each run one device batch (normally the whole chunk)
        def dp_key_of(r):
            fmt = r.format
            return 'DP' if 'DP' in fmt else 'LC' if 'LC' in fmt else None
        lo = 0
        k0 = dp_key_of(records[0])
        for i in range(1, len(records) + 1):
            k = dp_key_of(records[i]) if i < len(records) else object()
            if k != k0:
                self._process(records[lo:i])
                lo, k0 = i, k
        if self.world > 1:
            self.parts.append((self._key(), ''.join(self._cur).encode()))

    # ---- the batch pipeline: no Python object per record -------------------------------------------------------
    _SIMPLE_VALUES = (filters.CallFilterMinValue, filters._HipSTRRatio, filters.HipSTRCallMinSuppReads,
                      filters._GangSTRQexp, filters.GangSTRCallSpanOnly, filters.GangSTRCallSpanBoundOnly,
                      filters.GangSTRCallBadCI, filters.PopSTRCallRequireSupport)
    AD_COLUMNS = 16     # PopSTR's AD (Number=R) is decoded into this many columns; a batch with a record of more
                        # alleles takes the per-record loop

    def batch_path_ok(self, vcftype):
        """The batch pipeline (native reader -> native batch harmoniser -> device -> native record writer) covers
        callers whose records carry allele sequences, call filters whose reported value is a FORMAT number or a
        ratio of two, and the statistic locus filters; everything else takes the per-record loop.
        TRK_DUMPSTR_BATCH=0 forces the per-record loop."""
        from ..vcfnative import NativeVCFReader, VT_CODES
        a = self.args
        return (isinstance(self.invcf, NativeVCFReader) and vcftype.name in VT_CODES and
                all(isinstance(f, self._SIMPLE_VALUES) for f in self.call_filters) and
                len(self.invcf.samples) > 0 and
                _knobs.lab('TRK_DUMPSTR_BATCH', '1') != '0')

    def process_raw(self, rb, hz, format_kinds):
        """One batch through the batch pipeline.  False: a record is outside what it covers (nothing has been
        written or counted): the caller runs the batch through ``process``."""
        from .. import runtime
        from ..batch import HostBatch
        args = self.args
        # FORMAT planes the filters read, as the native reader decoded them
        keys = []
        for f in self.call_filters:
            for key, _ in f.planes():
                if key not in keys:
                    keys.append(key)
        # Every record must carry every FORMAT key a filter reads (the reference raises KeyError at
        # `record.format[self.field]`, filters.py:327-409; a plane of the native reader holds missing values where
        # a record lacks the key) and all records must agree on the depth field (DP, else LC: dumpSTR.py:688-695 is
        # judged per record) -- otherwise the per-record loop takes the batch and behaves as the reference does.
        formats = rb.format_columns()
        real = {'__minsupp': ('ALLREADS', 'GB'), '__rc': ('RC',), '__repci': ('REPCI',), '__ad': ('AD',)}
        dp_keys = set()
        for fmt in formats:
            for k in keys:
                if any(r not in fmt for r in real.get(k, (k,))):
                    return False
            dp_keys.add('DP' if 'DP' in fmt else 'LC' if 'LC' in fmt else None)
        if len(dp_keys) > 1:
            return False
        dp_key = dp_keys.pop() if dp_keys else None
        if dp_key is not None and dp_key not in keys:
            keys.append(dp_key)
        if any(k not in rb.plane_arrays() for k in keys):
            return False
        if any(isinstance(f, filters.PopSTRCallRequireSupport) for f in self.call_filters):
            # filters.py:858-867 indexes AD by the genotype index, negative ones from the END of the record's own
            # list: the batch's fixed-width plane answers alike where every allele has a column, every record is of
            # the tensor's ploidy and no call is half missing
            g = rb.gt
            if (int(np.max(np.diff(hz.allele_off), initial=0)) > self.AD_COLUMNS or
                    bool(np.any(np.asarray(rb.locus_ploidy) != g.shape[2])) or
                    bool(np.any((g < 0).any(axis=2) & (g >= 0).any(axis=2)))):
                return False
        index = {k: i for i, k in enumerate(keys)}
        arrays = [rb.plane_arrays()[k] for k in keys]
        specs = [f.spec(index) for f in self.call_filters]
        self.batch_no += 1
        # contigs the header does not declare are registered for EVERY record of the batch, on every rank, before the
        # ownership and --drop-filtered decisions (the per-record reader registers them as it parses)
        for chrom in rb.chroms():
            if chrom not in self.invcf.contigs_declared and chrom not in self.invcf.contigs_seen:
                self.invcf.contigs_seen.append(chrom)
        if not self._mine():
            return True
        compute = runtime.get_compute()
        dev = getattr(rb, 'dev', None)
        if dev is not None and getattr(compute, 'eng', None) is not None:
            # the sample columns were parsed on the device (TRK_DEVICE_PARSE=1): the tensor and the planes are there already;
            # the device arrays are the batch's from here on.  The host copies (the record writer's host tiers read the
            # values of fired filters, its decode path the genotypes) are made only if somebody asks: the arrays' memory
            # is held until the batch is released (RawBatch.host_defer)
            if not rb.host_defer():
                rb._host()
            hb = HostBatch.from_tables(dev['gt'], rb.locus_ploidy, hz.allele_off, hz.len_class, hz.str_class,
                                       hz.len_class_value, lists=hz.lists)
            arrays = [dev['planes'][k] for k in keys]
            dev['gt'] = None
            for k in keys:
                dev['planes'][k] = None
            # (the text and the offsets stay on the device for the record writer's device half; released with the batch)
        else:
            hb = HostBatch.from_tables(rb.gt, rb.locus_ploidy, hz.allele_off, hz.len_class, hz.str_class,
                                       hz.len_class_value, lists=hz.lists)
        kw = dict(compact=True) if getattr(compute, 'supports_compact', False) else {}
        # the string-only locus filters (HRUN, BED regions: filters.py:190-300) for the whole batch at once, from the
        # harmoniser's per-record tables: extern bits of the locus-filter kernel
        ext = None
        if self.host_filters:
            ext = np.zeros(rb.n, dtype=np.uint32)
            chroms = None
            for j, f in enumerate(self.host_filters):
                if isinstance(f, filters.Filter_LocusHrun):
                    per = hz.period
                    fire = ((per == 5) | (per == 6)) & (hz.hrun >= per)
                else:
                    if chroms is None:
                        chroms = rb.chrom_column()
                    ref_len = hz.allele_len[hz.allele_off[:-1]]          # record.ref_allele_length (repeat units)
                    # record.pos is the HARMONISED position (INFO START for HipSTR records with flanking bases,
                    # tr_harmonizer.py:407), as the per-record Filter_Regions.__call__ reads it
                    fire = f.overlaps_batch(chroms, hz.tr_pos, hz.tr_pos + ref_len)
                ext |= fire.astype(np.uint32) << np.uint32(j)
        t_dev = time.perf_counter()
        if dev is not None and kw.get('compact') and _knobs.env('TRK_DEVICE_FORMAT', '1') == '1':
            kw['keep_device'] = True         # (the mask and the planes stay on the device for trk_format_samples)
        ch, st, bits, lc = compute.dumpstr_batch(hb, arrays, specs, -1 if dp_key is None else index[dp_key],
                                                 dict(self.spec, extern_bits=ext), **kw)
        t_heads = time.perf_counter()
        _tick('upload_kernels_download', t_heads - t_dev)
        # the native writer first: if it declines, the batch has left no trace
        names = [f.name for f in self.call_filters]
        cfv = []
        cf_plane_idx = []            # plain-value filters: which of the uploaded planes holds the value
        for f in self.call_filters:
            if isinstance(f, filters._HipSTRRatio):
                cfv.append((f.name, 1, (rb.plane_arrays()[f.numerator], 0), (rb.plane_arrays()['DP'], 0)))
            elif isinstance(f, filters._GangSTRQexp):
                if len(f.cols) == 1:
                    cfv.append((f.name, 0, (rb.plane_arrays()['QEXP'], f.cols[0]), None))
                else:
                    cfv.append((f.name, 2, (rb.plane_arrays()['QEXP'], f.cols[0], f.cols[1]), None))
            elif isinstance(f, filters.GangSTRCallSpanOnly):
                cfv.append((f.name, 0, (rb.plane_arrays()['__rc'], 1), None))
            elif isinstance(f, filters.GangSTRCallSpanBoundOnly):
                cfv.append((f.name, 2, (rb.plane_arrays()['__rc'], 1, 3), None))
            elif isinstance(f, filters.GangSTRCallBadCI):
                cfv.append((f.name, 3, (rb.plane_arrays()['REPCN'], 0), (rb.plane_arrays()['__repci'], 0)))
            elif isinstance(f, filters.PopSTRCallRequireSupport):
                cfv.append((f.name, 4, (rb.plane_arrays()['__ad'], int(f.threshold)), None))
            else:
                key = f.planes()[0][0]
                cfv.append((f.name, 0, (rb.plane_arrays()[key], 0), None))
                cf_plane_idx.append(index[key])
        ul = bool(args.use_length)
        I0, F0 = st.locus_int[0], st.locus_f64[0]
        have = I0[:, L.LI_N_CALLED] > 0
        status = I0[:, L.LI_HWE_STATUS_LEN if ul else L.LI_HWE_STATUS_STR]
        if np.any(have & (status == L.HWE_INDEX_ERROR)) and not args.drop_filtered:
            raise IndexError("tuple index out of range (haploid genotypes have no HWE test)")

        def fired_names_of(b):
            names_ = [f.filter_name() for f in self.locus_filters if (b >> self.bit_of[id(f)]) & 1]
            if (b >> L.LOCF_NO_CALLS) & 1:
                names_.append('NO_CALLS_REMAINING')
            return names_

        def py_head(l):
            """The nine leading columns of output record l, in Python (vcfio.rewrite_info): what the native writer
            does for every record whose INFO column it covers."""
            fired_names = fired_names_of(int(bits[l]))
            f = rb.head_fields(l)
            if not args.drop_filtered:
                f[6] = ';'.join(fired_names) if fired_names else 'PASS'
            upd = [('HRUN', int(hz.hrun[l]))]
            I, Fv = I0[l], F0[l]
            o = int(hz.allele_off[l])
            n_alt = int(hz.allele_off[l + 1]) - o - 1
            if I[L.LI_N_CALLED] > 0:
                if I[L.LI_HWE_STATUS_LEN if ul else L.LI_HWE_STATUS_STR] == L.HWE_INDEX_ERROR:
                    raise IndexError("tuple index out of range (haploid genotypes have no HWE test)")
                ac = st.allele_count[0, o:o + n_alt + 1].tolist()
                upd += [('HET', float(Fv[L.LF_HET_LEN if ul else L.LF_HET_STR])),
                        ('HWEP', float(Fv[L.LF_HWEP_LEN if ul else L.LF_HWEP_STR])),
                        ('AC', 0 if n_alt == 0 else ",".join(str(x) for x in ac[1:])), ('REFAC', ac[0])]
            else:
                upd += [('HET', -1), ('HWEP', -1), ('AC', 0 if n_alt == 0 else ','.join(['0'] * n_alt)), ('REFAC', 0)]
            f[7] = vcfio.rewrite_info(self.invcf, f[7], upd)
            if 'FILTER' not in f[8].split(':'):
                f[8] = f[8] + ':FILTER'
            return '\t'.join(f)

        heads, native = None, None
        if _knobs.lab('TRK_DUMPSTR_NATIVE_HEADS', '1') != '0':
            # the heads are built by the native writer (trk_vcf_dumpstr_records): per record only the FILTER text, one
            # string per distinct pattern of fired locus filters
            ub, inv = np.unique(bits, return_inverse=True)
            texts, dropped = [], []
            for b in ub.tolist():
                fn = fired_names_of(int(b))
                dropped.append(bool(args.drop_filtered and fn))
                texts.append(None if args.drop_filtered else (';'.join(fn) if fn else 'PASS').encode())
            keep_rec = None
            if args.drop_filtered:
                keep_rec = (~np.asarray(dropped, dtype=bool)[inv]).astype(np.uint8)
                if np.any(have & (status == L.HWE_INDEX_ERROR) & (keep_rec != 0)):
                    raise IndexError("tuple index out of range (haploid genotypes have no HWE test)")
            ftext = None if args.drop_filtered else [texts[i] for i in inv.tolist()]
            native = dict(keep=keep_rec, filter_text=ftext, hrun=hz.hrun, have_stats=have.astype(np.uint8),
                          het=F0[:, L.LF_HET_LEN if ul else L.LF_HET_STR], hwep=F0[:, L.LF_HWEP_LEN if ul else L.LF_HWEP_STR],
                          allele_count=st.allele_count[0], allele_off=hz.allele_off, info_types=self.invcf.info_types,
                          py_head=py_head,
                          dev_call=getattr(ch, 'dev', None) if len(cf_plane_idx) == len(cfv) else None,
                          cf_plane_idx=cf_plane_idx)
        else:
            heads = []
            for l in range(rb.n):
                if args.drop_filtered and fired_names_of(int(bits[l])):
                    heads.append(None)
                    continue
                if len(rb.head_fields(l)) < 9:
                    return self._undo_batch()
                heads.append(py_head(l))
        t_lines = time.perf_counter()
        _tick('record_heads_python' if native is None else 'record_heads_setup', t_lines - t_heads)
        if not hasattr(self, '_out_ring'):
            self._out_ring = {}          # two output buffers for the run, taken in turn (one block is with the writer)
        text = rb.dumpstr_lines(heads, ch.mask, cfv, format_kinds, out_ring=self._out_ring, native=native)
        if getattr(ch, 'dev', None) is not None:
            ch.release_device()
        _tick('record_text_native', time.perf_counter() - t_lines)
        if text is None:
            return self._undo_batch()
        if ch.error[0]:
            l, s = int(ch.error[1]), int(ch.error[2])
            bad = np.zeros(len(self.sample_names), dtype=bool)
            bad[s] = True
            chrom, pos = rb.chrom_pos(l)
            raise ValueError("The following samples have calls but negative DP values at chromosome {} pos {}: {}"
                             .format(chrom, pos, str(self.sample_names[bad])))
        if lc[L.LC_HWE_ERRORS]:
            raise ValueError("binomtest: n must be a positive integer (a locus has allele calls but no fully "
                             "called genotype; the HWE filter cannot be evaluated)")
        if np.any(st.locus_int[0, :, L.LI_N_BAD]):
            raise IndexError("genotype index out of range for the alleles of a record")
        self._count(ch, lc, dp_key is not None)
        t_w = time.perf_counter()
        if self.world == 1:
            self.outvcf.write_bytes(text)
            _tick('write_output', time.perf_counter() - t_w)
        else:
            self.parts.append((self._key(), bytes(text)))
        return True

    def _undo_batch(self):
        self.batch_no -= 1
        return False

    def _count(self, ch, lc, have_dp):
        """sample_info / loc_info of one batch (dumpSTR.py:661, 686-713 and 946-971)."""
        si = self.sample_info
        si['numcalls'] += ch.sample_counters[0]
        for k, f in enumerate(self.call_filters):
            si[f.name] += ch.sample_counters[1 + k]
        if have_dp:
            si['totaldp'] += ch.totaldp
            si['totaldp'][ch.dp_missing > 0] = np.nan
        else:
            si['totaldp'][:] = np.nan
            self.no_dp = True
        li = self.loc_info
        li['totalcalls'] += int(lc[L.LC_TOTALCALLS])
        li['PASS'] += int(lc[L.LC_PASS])
        li['NO_CALLS_REMAINING'] += int(lc[L.LC_NO_CALLS])
        for f in self.locus_filters:
            li[f.filter_name()] += int(lc[L.LC_FILTER0 + self.bit_of[id(f)]])

    def _process(self, records):
        from .. import runtime
        args = self.args
        hb = pack_records(records)
        planes = _Planes(records, self.call_filters)
        specs = [f.spec(planes.index) for f in self.call_filters]
        ext = None
        if self.host_filters:
            ext = np.zeros(len(records), dtype=np.uint32)
            for l, r in enumerate(records):
                for j, f in enumerate(self.host_filters):
                    if f(r) is not None:
                        ext[l] |= np.uint32(1 << j)
        spec = dict(self.spec, extern_bits=ext)
        ch, st, bits, lc = runtime.get_compute().dumpstr_batch(hb, planes.arrays, specs, planes.dp_plane, spec)
        if ch.error[0]:
            l, s = int(ch.error[1]), int(ch.error[2])
            bad = np.zeros(len(self.sample_names), dtype=bool)
            bad[s] = True
            raise ValueError("The following samples have calls but negative DP values at chromosome {} pos {}: {}"
                             .format(records[l].chrom, records[l].pos, str(self.sample_names[bad])))
        if lc[L.LC_HWE_ERRORS]:
            raise ValueError("binomtest: n must be a positive integer (a locus has allele calls but no fully "
                             "called genotype; the HWE filter cannot be evaluated)")
        if np.any(st.locus_int[0, :, L.LI_N_BAD]):
            raise IndexError("genotype index out of range for the alleles of a record")
        # ---- counters (dumpSTR.py:661,686-713 and 946-971) ----
        si = self.sample_info
        si['numcalls'] += ch.sample_counters[0]
        for k, f in enumerate(self.call_filters):
            si[f.name] += ch.sample_counters[1 + k]
        if planes.dp_plane >= 0:
            si['totaldp'] += ch.totaldp
            si['totaldp'][ch.dp_missing > 0] = np.nan
        else:
            si['totaldp'][:] = np.nan
            self.no_dp = True
        li = self.loc_info
        li['totalcalls'] += int(lc[L.LC_TOTALCALLS])
        li['PASS'] += int(lc[L.LC_PASS])
        li['NO_CALLS_REMAINING'] += int(lc[L.LC_NO_CALLS])
        for f in self.locus_filters:
            li[f.filter_name()] += int(lc[L.LC_FILTER0 + self.bit_of[id(f)]])
        # ---- per-record text (host) ----
        names = [f.name for f in self.call_filters]
        ul = bool(args.use_length)
        for l, r in enumerate(records):
            v = r.vcfrecord
            mrow = ch.mask[l]
            fired = (mrow & np.uint32(0x7fffffff)) != 0
            vals = [_filter_values(f, planes, hb, l) if np.any((mrow >> np.uint32(k)) & 1) else None
                    for k, f in enumerate(self.call_filters)]
            v.set_format('FILTER', vcfio.CallFilterColumn(mrow, names, vals))
            filtered = fired & ((mrow & np.uint32(L.TRK_MASK_NOCALL)) == 0)
            if np.any(filtered):
                _null_filtered(v, filtered, r.GetMaxPloidy())
            b = int(bits[l])
            fired_names = [f.filter_name() for f in self.locus_filters if (b >> self.bit_of[id(f)]) & 1]
            if (b >> L.LOCF_NO_CALLS) & 1:
                fired_names.append('NO_CALLS_REMAINING')
            if not args.drop_filtered:
                v.FILTER = ';'.join(fired_names) if fired_names else 'PASS'
            if args.drop_filtered and fired_names:
                continue
            # INFO recompute (dumpSTR.py:1304-1336)
            seq = r.full_alleles[0] if r.HasFullStringGenotypes() else r.ref_allele
            v.INFO['HRUN'] = utils.GetHomopolymerRun(seq)
            I, Fv = st.locus_int[0, l], st.locus_f64[0, l]
            n_alt = len(r.alt_alleles)
            if I[L.LI_N_CALLED] > 0:
                status = I[L.LI_HWE_STATUS_LEN if ul else L.LI_HWE_STATUS_STR]
                if status == L.HWE_INDEX_ERROR:
                    raise IndexError("tuple index out of range (haploid genotypes have no HWE test)")
                v.INFO['HET'] = float(Fv[L.LF_HET_LEN if ul else L.LF_HET_STR])
                v.INFO['HWEP'] = float(Fv[L.LF_HWEP_LEN if ul else L.LF_HWEP_STR])
                o = int(hb.allele_off[l])
                ac = st.allele_count[0, o:o + n_alt + 1]
                v.INFO['AC'] = 0 if n_alt == 0 else ",".join(str(int(x)) for x in ac[1:])
                v.INFO['REFAC'] = int(ac[0])
            else:
                v.INFO['HET'] = -1
                v.INFO['HWEP'] = -1
                v.INFO['AC'] = 0 if n_alt == 0 else ','.join(['0'] * n_alt)
                v.INFO['REFAC'] = 0
            self.emit(v)


def getargs():  # pragma: no cover
    parser = argparse.ArgumentParser(__doc__, formatter_class=utils.ArgumentDefaultsHelpFormatter)
    io = parser.add_argument_group("Input/output")
    io.add_argument("--vcf", help="Input STR VCF file", type=str, required=True)
    io.add_argument("--out", help="Prefix for output files", type=str, required=True)
    io.add_argument("--zip", help="Produce a bgzipped and tabix indexed output VCF", action="store_true")
    io.add_argument("--vcftype", help="Options=%s" % [str(i) for i in trh.VcfTypes.__members__], type=str,
                    default="auto")
    lg = parser.add_argument_group("Locus-level filters (tool agnostic)")
    lg.add_argument("--min-locus-callrate", help="Minimum locus call rate", type=float)
    lg.add_argument("--min-locus-hwep", help="Filter loci failing HWE at this p-value threshold", type=float)
    lg.add_argument("--min-locus-het", help="Minimum locus heterozygosity", type=float)
    lg.add_argument("--max-locus-het", help="Maximum locus heterozygosity", type=float)
    lg.add_argument("--use-length", help="Calculate per-locus stats (het, HWE) collapsing alleles by length",
                    action="store_true")
    lg.add_argument("--filter-regions", help="Comma-separated list of BED files of regions to filter. Must be "
                    "bgzipped and tabix indexed", type=str)
    lg.add_argument("--filter-regions-names", help="Comma-separated list of filter names for each BED filter "
                    "file", type=str)
    lg.add_argument("--filter-hrun", help="Filter STRs with long homopolymer runs.", action="store_true")
    lg.add_argument("--drop-filtered", help="Drop filtered records from output", action="store_true")
    typed = {'01': float, '>=0': int, '>=0 ': int}
    helps = {'max_call_flank_indel': "Maximum call flank indel rate", 'max_call_stutter': "Maximum call stutter rate",
             'min_supp_reads': "Minimum supporting reads for each allele", 'min_call_DP': "Minimum call coverage",
             'max_call_DP': "Maximum call coverage", 'min_call_Q': "Minimum call quality score",
             'min_call_LC': "Minimum call coverage", 'max_call_LC': "Maximum call coverage"}
    for key, (_, label, rules, _) in _CALLER_RULES.items():
        grp = parser.add_argument_group("Call-level filters specific to %s output" % label)
        for arg, rule, _ in rules:
            t = float if arg.endswith('_ML') else typed[rule]
            grp.add_argument(_flag(arg), help=helps.get(arg.split('_', 1)[1], arg.replace('_', ' ')), type=t)
        if key == 'gangstr':
            grp.add_argument("--gangstr-filter-span-only", help="Filter out all calls that only have spanning read "
                             "support", action="store_true")
            grp.add_argument("--gangstr-filter-spanbound-only", help="Filter out all reads except spanning and "
                             "bounding", action="store_true")
            grp.add_argument("--gangstr-filter-badCI", help="Filter regions where the ML estimate is not in the CI",
                             action="store_true")
    dg = parser.add_argument_group("Debugging parameters")
    dg.add_argument("--num-records", help="Only process this many records", type=int)
    dg.add_argument("--die-on-warning", help="Quit if a record can't be parsed", action="store_true")
    dg.add_argument("--verbose", help="Print out extra info", action="store_true")
    vg = parser.add_argument_group("Version")
    vg.add_argument("--version", action="version", version='{version}'.format(version=__version__))
    return parser.parse_args()


_INFO_FIELDS = [  # (ID, Description, Type, Number)  dumpSTR.py:1130-1208
    ('AC', 'Alternate allele counts', 'Integer', 'A'),
    ('REFAC', 'Reference allele count', 'Integer', 1),
    ('HET', 'Heterozygosity', 'Float', 1),
    ('HWEP', 'HWE p-value for obs. vs. exp het rate', 'Float', 1),
    ('HRUN', 'Length of longest homopolymer run', 'Integer', 1),
]
_FIELD_ISSUE = (
    "Error: The {} field '{}' is present in the input VCF and doesn't have the expected Type and Number "
    "so it can't be worked with. Please use 'bcftools annotate --rename-annots' or another equivalent tool to "
    "rename or remove the field and then rerun dumpSTR. (--rename-annots is a flag available in the development "
    "version of bcftools which can be installed from https://samtools.github.io/bcftools/) (You can pipe the "
    "output of that command into dumpSTR if you wish to avoid writing another file to disk)")


def main(args):
    invcf = utils.LoadSingleReader(args.vcf, checkgz=False)
    if invcf is None:
        return 1
    if not os.path.exists(os.path.dirname(os.path.abspath(args.out))):
        common.WARNING("Error: The directory which contains the output location {} does"
                       " not exist".format(args.out))
        return 1
    if os.path.isdir(args.out + ".vcf"):
        common.WARNING("Error: The output location {} is a directory".format(args.out))
        return 1
    if args.out[-1] in {'.', '/'}:
        common.WARNING("Output prefix must not end in '/' or '.'")
        return 1

    harmonizer = trh.TRRecordHarmonizer(invcf, args.vcftype)
    is_beagle = harmonizer.IsBeagleVCF()
    vcftype = harmonizer.vcftype

    format_fields, info_fields, old_filters = {}, {}, {}
    for h in invcf.header_iter():
        kind = h['HeaderType']
        if kind == 'INFO':
            info_fields[h['ID']] = h
        elif kind == 'FORMAT':
            format_fields[h['ID']] = h
        elif kind == 'FILTER':
            old_filters[h['ID']] = h
    if not CheckFilters(format_fields, args, vcftype, is_beagle):
        return 1

    issues = False
    if 'FILTER' not in format_fields:
        invcf.add_format_to_header({'ID': 'FILTER', 'Description': 'call-level filters that have been applied',
                                    'Type': 'String', 'Number': 1})
    elif format_fields['FILTER']['Type'] != 'String' or format_fields['FILTER']['Number'] != '1':
        issues = True
        common.WARNING(_FIELD_ISSUE.format('format', 'FILTER'))
    for fid, desc, typ, num in _INFO_FIELDS:
        if fid not in info_fields:
            invcf.add_info_to_header({'ID': fid, 'Description': desc, 'Type': typ, 'Number': num})
        elif info_fields[fid]['Type'] != typ or info_fields[fid]['Number'] != str(num):
            issues = True
            common.WARNING(_FIELD_ISSUE.format('info', fid))
        elif info_fields[fid]['Description'] != desc:
            common.WARNING("Overwriting the preexisting info {} field".format(fid))
    if issues:
        return 1

    invcf.add_filter_to_header({
        "ID": "NO_CALLS_REMAINING",
        "Description": "All calls at this locus were already nocalls or were individually "
                       "filtered before the locus level filters were applied."})
    try:
        locus_filters = BuildLocusFilters(args)
    except ValueError:
        return 1
    for f in locus_filters:
        if f.filter_name() not in old_filters:
            invcf.add_filter_to_header({"ID": f.filter_name(), "Description": f.description()})
        elif old_filters[f.filter_name()]['Description'] != f.description():
            common.WARNING("Using locus level filter " + f.filter_name() + "which has the same name as a FILTER "
                           "field that already exists in the input VCF. The filters DumpSTR writes to the output "
                           "with this name will possibly have different meanings than the filters with the name "
                           "that are already present.")
    call_filters = BuildCallFilters(args)

    # let the native reader decode what the call filters read (GT is always decoded)
    if hasattr(invcf, 'select_format'):
        from .. import vcfnative
        wanted = []
        for f in call_filters:
            wanted.extend(k for k, _ in f.planes())
        for cand in ('DP', 'LC'):
            if cand in format_fields:
                wanted.append(cand)
                break
        for key in dict.fromkeys(wanted):
            if key == '__minsupp':
                invcf.select_format('ALLREADS', vcfnative.KIND_MINSUPP, 1, alias='__minsupp')
            elif key == '__rc':
                invcf.select_format('RC', vcfnative.KIND_INT, 4, alias='__rc')
            elif key == '__repci':
                invcf.select_format('REPCI', vcfnative.KIND_INT_RANGES, 4, alias='__repci')
            elif key == '__ad':      # Number=R: one column per allele, up to AD_COLUMNS of them
                if 'AD' in format_fields and format_fields['AD']['Type'] == 'Integer':
                    invcf.select_format('AD', vcfnative.KIND_INT, _Run.AD_COLUMNS, alias='__ad')
            elif key in format_fields and format_fields[key]['Type'] in ('Integer', 'Float') \
                    and format_fields[key]['Number'].isdigit():
                invcf.select_format(key, ncol=int(format_fields[key]['Number']))

    suffix = '.vcf.gz' if args.zip else '.vcf'
    from .. import dist
    rank = dist.get_comm()[0]
    # in a sharded run only rank 0 writes the outputs; the other ranks get a scratch writer
    vcf_path = args.out + suffix if rank == 0 else args.out + '.rank%d.tmp' % rank + suffix
    outvcf = MakeWriter(vcf_path, invcf, " ".join(sys.argv))
    if outvcf is None:
        return 1
    run = _Run(args, invcf, call_filters, locus_filters, outvcf)

    n_samples = max(len(invcf.samples), 1)
    batch_loci = max(1, min(2048, BATCH_CELLS // n_samples))
    record_counter = 0
    batch = []
    use_batches = run.batch_path_ok(vcftype)
    if use_batches:
        from .. import runtime
        kinds = {k: (1 if h['Type'] == 'Integer' else 2 if h['Type'] == 'Float' else 4) for k, h in format_fields.items()}
        invcf.use_buffers(getattr(runtime.get_compute(), 'host_buffer', None), ring=2,
                          release=getattr(runtime.get_compute(), 'host_release', None))
        # Batch n + 1 is read while batch n is filtered and written (TRK_VCF_READ_AHEAD=0: off).  Rounds 3 and 4 (first half)
        # kept it off -- the phases it ran beside slowed down by what it hid: on the GPU boxes the container is granted
        # 16 CPUs' worth of time and the command line was bound by CPU seconds.  With the parse on the device the read is
        # the inflate alone and the overlap is real: 0.37 -> 0.30 s per GB (profiles/r04_notes.md section 15).
        # (--num-records shortens the last batch: no read beyond it.)
        if args.num_records is None and _knobs.env('TRK_VCF_READ_AHEAD', '1') == '1':
            invcf.read_ahead()
        # The sample columns are parsed on the device (trk_parse_samples, round 4; TRK_DEVICE_PARSE=0: on the host) -- scalar
        # Integer / Float planes only, so HipSTR's filter sets; the reader refuses otherwise and parses on the host as before
        device_parse = (_knobs.env('TRK_DEVICE_PARSE', '1') == '1' and hasattr(invcf, 'device_parse') and
                        getattr(runtime.get_compute(), 'eng', None) is not None and
                        invcf.device_parse(runtime.get_compute().eng))
    LAST_RUN.clear()
    LAST_RUN.update(path='batch' if use_batches else 'per-record', batches=0, fallback_batches=0)
    if use_batches:
        LAST_RUN['device_parse'] = bool(device_parse)
        # ... and the BGZF members are inflated there too (trk_inflate_blocks, round 5; TRK_DEVICE_INFLATE=0: by the reader's
        # threads): the compressed bytes cross PCIe, the host sees the newlines and the heads of the lines
        LAST_RUN['device_inflate'] = bool(device_parse and _knobs.env('TRK_DEVICE_INFLATE', DEVICE_INFLATE_DEFAULT['dumpSTR']) == '1' and
                                          hasattr(invcf, 'device_inflate') and invcf.device_inflate(runtime.get_compute().eng))
    last_rb = None
    while use_batches:
        if last_rb is not None:
            last_rb.release_device()
            last_rb = None
        want = batch_loci
        if args.num_records is not None:          # dumpSTR.py:1292-1293: the first num_records records
            want = min(batch_loci, args.num_records - record_counter)
            if want <= 0:
                break
        t0 = time.perf_counter()
        rb = last_rb = invcf.read_raw_batch(want)
        t1 = time.perf_counter()
        _tick('read_parse', t1 - t0)
        if rb.n == 0:
            break
        record_counter += rb.n
        hz = rb.harmonize(vcftype.name)
        _tick('harmonize', time.perf_counter() - t1)
        if args.verbose:
            for l in range(rb.n):
                common.MSG("Processing %s:%s" % rb.chrom_pos(l))
        LAST_RUN['batches'] += 1
        if hz.n_python == 0 and run.process_raw(rb, hz, kinds):
            continue
        LAST_RUN['fallback_batches'] += 1
        LAST_RUN['path'] = 'mixed'
        # a batch the native pieces do not cover: through the record objects, with the per-record loop's handling
        # of unparsable records
        try:
            recs, it, k = [], rb.iter_variants(), 0
            while True:
                try:
                    v = next(it)
                except StopIteration:
                    break
                except Exception:        # a line that does not parse: TRRecordHarmonizer.__next__'s message
                    raise ValueError("Unable to parse the " + str(record_counter - rb.n + k + 2) + "th tandem repeat in "
                                     "the provided VCF. Check that it is properly formatted.")
                recs.append(trh.HarmonizeRecord(vcftype, v))
                k += 1
            run.process(recs)
        except TypeError as te:
            if 'missing' in te.args[0] and 'mandatory' in te.args[0]:
                common.WARNING("Could not parse VCF.\n" + te.args[0])
                return 1
            raise te
        except ValueError as ve:
            if 'properly formatted' in ve.args[0]:
                common.WARNING("Could not parse VCF.\n" + ve.args[0])
                return 1
            raise ve
    if use_batches and last_rb is not None:
        last_rb.release_device()
    while not use_batches:
        try:
            record = next(harmonizer)
        except StopIteration:
            break
        except TypeError as te:
            message = te.args[0]
            if 'missing' in message and 'mandatory' in message:
                common.WARNING("Could not parse VCF.\n" + message)
                return 1
            raise te
        except ValueError as ve:
            message = ve.args[0]
            if 'properly formatted' in message:
                common.WARNING("Could not parse VCF.\n" + message)
                return 1
            raise ve
        if args.verbose:
            common.MSG("Processing %s:%s" % (record.chrom, record.pos))
        record_counter += 1
        if args.num_records is not None and record_counter > args.num_records:
            break
        batch.append(record)
        if len(batch) >= batch_loci:
            run.process(batch)
            batch = []
    run.process(batch)
    run.finish()

    invcf.close()
    outvcf.close()
    if run.rank != 0:
        os.remove(vcf_path)               # only rank 0's file holds the merged records
        return 0
    WriteSampLog(run.sample_info, invcf.samples, args.out + ".samplog.tab")
    WriteLocLog(run.loc_info, args.out + ".loclog.tab")
    if args.zip:
        # the reference shells out to `tabix` (dumpSTR.py:1347-1352); the index is written here
        from .. import tabix
        try:
            if getattr(outvcf, '_recs', None) is not None:
                outvcf.write_index()      # from the places the writer noted: the output is not read back
            else:
                tabix.build(args.out + suffix)
        except (OSError, ValueError) as e:
            common.WARNING("Tabix failed with returncode 1 (%s)" % e)
            return 1
    return 0


def run():  # pragma: no cover
    sys.exit(main(getargs()))


if __name__ == "__main__":  # pragma: no cover
    run()
