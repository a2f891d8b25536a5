"""Locus-level and call-level filters of dumpSTR with the reference's class
names, ``name`` / ``filter_name()`` strings and thresholds
(trtools/dumpSTR/filters.py).

In the reference every filter object evaluates numpy expressions over all
samples of one record.  Here a filter object is a *specification*: it names the
FORMAT planes it reads and the ``trk_call_filter`` opcode (include/trk.h) that
the ``k_call_filter`` kernel evaluates for every call of a batch.  String
FORMAT fields (ALLREADS/GB, RC, REPCI) are parsed once on the host into integer
planes; the predicate itself runs on the device.  ``value()`` reproduces the
number the reference puts behind the filter name in the FORMAT/FILTER text.
"""
import os

import numpy as np

from .. import _lib as L
from ..utils import common, utils

INT_MISSING = -2147483648


class FilterBase:
    """Interface of locus-level filters (filters.py:15-29)."""
    name = 'NotYetImplemented'

    def __call__(self, record):
        raise NotImplementedError

    def filter_name(self):
        raise NotImplementedError

    def description(self):
        return ''


# ---------------------------------------------------------------------------
# locus-level filters: decided by k_locus_filter from the batch statistics.
# `bit` is the TRK_LOCF_* bit; `__call__` keeps the per-record API.
# ---------------------------------------------------------------------------

def _record_stats(record):
    st, _ = record._device_stats()
    return st.locus_int[0, 0], st.locus_f64[0, 0]


def _on_device(record):
    """A TRRecord of this package has its statistics on the device; anything else that merely offers the TRRecord methods
    a filter reads (the duck-typed records of the reference's own filter tests, dumpSTR/tests/test_filters.py:68-84) is
    asked through those methods, as the reference's filters ask (filters.py:96-100, 138-142, 179-183)."""
    return hasattr(record, '_device_stats')


def _hwep_of(record, uselength):
    if not _on_device(record):
        return utils.GetHardyWeinbergBinomialTest(record.GetAlleleFreqs(uselength=uselength),
                                                  record.GetGenotypeCounts(uselength=uselength))
    I, F = _record_stats(record)
    status = I[L.LI_HWE_STATUS_LEN if uselength else L.LI_HWE_STATUS_STR]
    if status == L.HWE_VALUE_ERROR:
        raise ValueError("n must be a positive integer (no fully called genotype)")
    if status == L.HWE_INDEX_ERROR:
        raise IndexError("tuple index out of range")
    return F[L.LF_HWEP_LEN if uselength else L.LF_HWEP_STR]


def _het_of(record, uselength):
    if not _on_device(record):
        return utils.GetHeterozygosity(record.GetAlleleFreqs(uselength=uselength))
    _, F = _record_stats(record)
    return F[L.LF_HET_LEN if uselength else L.LF_HET_STR]


class Filter_MinLocusCallrate(FilterBase):
    """filters.py:35-64."""
    name = 'CALLRATE'
    bit = L.LOCF_CALLRATE

    def __init__(self, min_locus_callrate):
        self.threshold = min_locus_callrate

    def __call__(self, record):
        rate = record.GetCallRate()
        return rate if rate < self.threshold else None

    def filter_name(self):
        return self.name + str(self.threshold)


class Filter_MinLocusHWEP(FilterBase):
    """filters.py:66-106."""
    name = 'HWE'
    bit = L.LOCF_HWE

    def __init__(self, min_locus_hwep, uselength=False):
        self.threshold = min_locus_hwep
        self.uselength = uselength

    def __call__(self, record):
        hwep = _hwep_of(record, self.uselength)
        return hwep if hwep < self.threshold else None

    def filter_name(self):
        return self.name + str(self.threshold)


class Filter_MinLocusHet(FilterBase):
    """filters.py:108-147."""
    name = 'HETLOW'
    bit = L.LOCF_HETLOW

    def __init__(self, min_locus_het, uselength=False):
        self.threshold = min_locus_het
        self.uselength = uselength

    def __call__(self, record):
        het = _het_of(record, self.uselength)
        return het if het < self.threshold else None

    def filter_name(self):
        return self.name + str(self.threshold)


class Filter_MaxLocusHet(FilterBase):
    """filters.py:149-188."""
    name = 'HETHIGH'
    bit = L.LOCF_HETHIGH

    def __init__(self, max_locus_het, uselength=False):
        self.threshold = max_locus_het
        self.uselength = uselength

    def __call__(self, record):
        het = _het_of(record, self.uselength)
        return het if het > self.threshold else None

    def filter_name(self):
        return self.name + str(self.threshold)


class Filter_LocusHrun(FilterBase):
    """Penta/hexanucleotide STRs with a homopolymer run >= the period
    (filters.py:190-217).  Pure string work on the reference allele: host."""
    name = 'HRUN'
    bit = None   # evaluated on the host, passed to the device as an extern bit

    def __call__(self, record):
        seq = record.full_alleles[0] if record.HasFullStringGenotypes() else record.ref_allele
        hrun = utils.GetHomopolymerRun(seq)
        if "PERIOD" not in record.info:
            return None
        if record.info["PERIOD"] in [5, 6] and hrun >= record.info["PERIOD"]:
            return hrun
        return None

    def filter_name(self):
        return self.name


class _BedIndex:
    """Interval lookup over a (b)gzipped BED file (the reference uses pysam's
    tabix reader, filters.py:266-292)."""

    def __init__(self, filename):
        import gzip
        self.by_chrom = {}
        with gzip.open(filename, 'rt') as fh:
            for line in fh:
                if not line.strip() or line.startswith('#'):
                    continue
                f = line.rstrip('\n').split('\t')
                self.by_chrom.setdefault(f[0], []).append((int(f[1]), int(f[2])))
        for iv in self.by_chrom.values():
            iv.sort()

    def overlaps(self, chrom, start1, end1):
        """tabix-style region chrom:start1-end1 (1-based, inclusive)."""
        if chrom not in self.by_chrom:
            return False
        beg0, end0 = int(float(start1)) - 1, int(float(end1))
        return any(s < end0 and e > beg0 for s, e in self.by_chrom[chrom])

    def overlaps_many(self, chrom, start1, end1):
        """``overlaps`` for arrays of regions on one chromosome: some interval has s < end0 and e > beg0 <=> the
        largest end among the intervals that start before end0 lies beyond beg0 (starts sorted, running maximum of
        the ends)."""
        import numpy as np
        out = np.zeros(len(start1), dtype=bool)
        iv = self.by_chrom.get(chrom)
        if not iv:
            return out
        tab = self._arrays.get(chrom) if hasattr(self, '_arrays') else None
        if tab is None:
            if not hasattr(self, '_arrays'):
                self._arrays = {}
            starts = np.array([s for s, _ in iv], dtype=np.int64)
            ends = np.maximum.accumulate(np.array([e for _, e in iv], dtype=np.int64))
            tab = self._arrays[chrom] = (starts, ends)
        starts, ends = tab
        beg0 = np.trunc(np.asarray(start1, dtype=np.float64)).astype(np.int64) - 1
        end0 = np.trunc(np.asarray(end1, dtype=np.float64)).astype(np.int64)
        k = np.searchsorted(starts, end0, side='left')
        has = k > 0
        out[has] = ends[k[has] - 1] > beg0[has]
        return out


def create_region_filter(name, filename):
    """Locus filter flagging records that overlap a BED file (filters.py:219-300).
    Returns None (after a warning) when the file is unusable."""
    class Filter_Regions(FilterBase):
        bit = None

        def __init__(self, name, filename):
            self.threshold = ""
            self.name = name
            self.pass_checks = True
            self.regions = None
            self.LoadRegions(filename)

        def LoadRegions(self, filename):
            problem = None
            if not filename.endswith(".bed.gz") and not filename.endswith(".bed.bgz"):
                problem = "Make sure %s is bgzipped and indexed" % filename
            elif not os.path.isfile(filename):
                problem = "Could not find regions BED file %s" % filename
            elif not os.path.isfile(filename + ".tbi"):
                problem = "Could not find tabix index %s.tbi" % filename
            if problem:
                common.WARNING(problem)
                self.pass_checks = False
                return
            self.regions = _BedIndex(filename)

        def __call__(self, record):
            if self.regions is None:
                return None
            chrom = str(record.chrom)
            other = chrom.replace("chr", "") if "chr" in chrom else "chr" + chrom
            end = record.pos + record.ref_allele_length
            for c in (chrom, other):
                if self.regions.overlaps(c, record.pos, end):
                    return self.name
            return None

        def overlaps_batch(self, chroms, pos, end):
            """``__call__`` for a batch: chroms list of str, pos / end arrays (end = pos + ref_allele_length)."""
            import numpy as np
            out = np.zeros(len(chroms), dtype=bool)
            if self.regions is None:
                return out
            chroms = np.asarray(chroms, dtype=object)
            for c in set(chroms.tolist()):
                idx = np.flatnonzero(chroms == c)
                other = c.replace("chr", "") if "chr" in c else "chr" + c
                for cc in (c, other):
                    out[idx] |= self.regions.overlaps_many(cc, pos[idx], end[idx])
            return out

        def filter_name(self):
            return self.name

        def description(self):
            return 'Filter TRs overlapping this region'

    f = Filter_Regions(name, filename)
    return f if f.pass_checks else None


# ---------------------------------------------------------------------------
# call-level filters
# ---------------------------------------------------------------------------

class Reason:
    """Base of call-level filters (filters.py:306-324).

    ``planes()``  -> list of (key, builder) the filter needs; builder(record) returns
                     the ``[S, k]`` int32/float32 array of that record;
    ``spec(ix)``  -> trk_call_filter fields, ``ix`` maps plane key -> plane index;
    ``value(get, l)`` -> float64[S] the number the reference reports for a fired call."""
    name = ""

    def GetReason(self):
        return self.name

    def __call__(self, record):
        """Reference-style per-record evaluation (float64[S], nan = not filtered)."""
        from .dumpSTR import evaluate_call_filters
        mask, values = evaluate_call_filters([record], [self])
        out = np.full(record.GetNumSamples(), np.nan)
        hit = (mask[0] & 1).astype(bool)
        out[hit] = values[0][0][hit]
        return out


def _field(key):
    def build(record):
        return record.format[key]
    return (key, build)


class CallFilterMinValue(Reason):
    """filters.py:327-367."""

    def __init__(self, name, field, threshold):
        self.name = name + str(threshold)
        self.field = field
        self.threshold = threshold

    def planes(self):
        return [_field(self.field)]

    def spec(self, ix):
        return dict(op=L.F_LT, plane_a=ix[self.field], col_a=0, thr=self.threshold)

    def value(self, get, l):
        return get(self.field)[l][:, 0].astype(float)


class CallFilterMaxValue(CallFilterMinValue):
    """filters.py:369-409."""

    def spec(self, ix):
        return dict(op=L.F_GT, plane_a=ix[self.field], col_a=0, thr=self.threshold)


class _HipSTRRatio(Reason):
    numerator = None

    def __init__(self, threshold, rename=None):
        self.threshold = threshold
        if rename is not None:
            self.name = rename
        self.name += str(threshold)

    def planes(self):
        return [_field(self.numerator), _field('DP')]

    def spec(self, ix):
        return dict(op=L.F_RATIO_GT, plane_a=ix[self.numerator], col_a=0, plane_b=ix['DP'], col_b=0,
                    thr=self.threshold)

    def value(self, get, l):
        with np.errstate(divide='ignore', invalid='ignore'):
            return get(self.numerator)[l][:, 0] / get('DP')[l][:, 0]


class HipSTRCallFlankIndels(_HipSTRRatio):
    """filters.py:415-449."""
    name = "HipSTRCallFlankIndels"
    numerator = 'DFLANKINDEL'


class HipSTRCallStutter(_HipSTRRatio):
    """filters.py:451-484."""
    name = "HipSTRCallStutter"
    numerator = 'DSTUTTER'


def _min_supp_reads(record):
    """Host pre-parse of ALLREADS ('len|count;...') and GB ('a|b' / 'a/b') into the number
    the reference compares with the threshold: min over the call's alleles of the
    supporting read count, 0 when ALLREADS is absent/missing for the sample
    (filters.py:519-567).  int32 [S, 1]."""
    n = record.GetNumSamples()
    native = getattr(record.vcfrecord, '_fmt_cache', {}).get('__minsupp')
    if native is not None:           # decoded by the native reader (trk_vcf.h TRK_VCF_MINSUPP)
        return native
    out = np.zeros((n, 1), dtype=np.int32)
    if "ALLREADS" not in record.format:
        return out
    allreads = record.format["ALLREADS"]
    gb = record.format["GB"]
    called = record.GetCalledSamples()
    delim = None
    for i in np.nonzero(called)[0]:
        ar = str(allreads[i])
        if ar == '' or ar == '.':
            continue
        g = str(gb[i])
        if delim is None:
            if "/" in g:
                delim = "/"
            elif "|" in g:
                delim = "|"
            else:
                raise ValueError("Cant't identify phasing char ('|' or '/') in GB field")
        reads = {}
        for tok in ar.split(';'):
            k, v = tok.split('|')
            reads[int(k)] = int(v)
        out[i, 0] = min(reads.get(int(x), 0) for x in g.split(delim))
    return out


class HipSTRCallMinSuppReads(Reason):
    """filters.py:486-567."""
    name = "HipSTRMinSuppReads"

    def __init__(self, threshold, rename=None):
        self.threshold = threshold
        if rename is not None:
            self.name = rename
        self.name += str(threshold)

    def planes(self):
        return [('__minsupp', _min_supp_reads)]

    def spec(self, ix):
        return dict(op=L.F_CALLED_LT, plane_a=ix['__minsupp'], col_a=0, thr=self.threshold)

    def value(self, get, l):
        return get('__minsupp')[l][:, 0].astype(float)


class _GangSTRQexp(Reason):
    cols = ()

    def __init__(self, threshold):
        self.threshold = threshold
        self.name += str(threshold)

    def planes(self):
        return [_field('QEXP')]

    def spec(self, ix):
        if len(self.cols) == 1:
            return dict(op=L.F_CALLED_LT, plane_a=ix['QEXP'], col_a=self.cols[0], thr=self.threshold)
        return dict(op=L.F_CALLED_SUM_LT, plane_a=ix['QEXP'], col_a=self.cols[0], col_a2=self.cols[1],
                    thr=self.threshold)

    def value(self, get, l):
        q = get('QEXP')[l]
        v = q[:, self.cols[0]]
        if len(self.cols) == 2:
            v = v + q[:, self.cols[1]]          # float32 sum, as in the reference
        return v.astype(float)


class GangSTRCallExpansionProbHom(_GangSTRQexp):
    """filters.py:573-605."""
    name = "GangSTRCallExpansionProbHom"
    cols = (2,)


class GangSTRCallExpansionProbHet(_GangSTRQexp):
    """filters.py:607-639."""
    name = "GangSTRCallExpansionProbHet"
    cols = (1,)


class GangSTRCallExpansionProbTotal(_GangSTRQexp):
    """filters.py:641-674."""
    name = "GangSTRCallExpansionProbTotal"
    cols = (1, 2)


def _split_ints(arr, ncol, seps):
    """String FORMAT array -> int32 [S, ncol]; unparsable / missing entries -> INT_MISSING."""
    out = np.full((len(arr), ncol), INT_MISSING, dtype=np.int32)
    for i, s in enumerate(arr):
        s = str(s)
        for sep in seps[1:]:
            s = s.replace(sep, seps[0])
        toks = s.split(seps[0])
        if len(toks) != ncol:
            continue
        try:
            out[i] = [int(t) for t in toks]
        except ValueError:
            pass
    return out


def _native(record, key):
    return getattr(record.vcfrecord, '_fmt_cache', {}).get(key)


def _rc_plane(record):
    native = _native(record, '__rc')
    return native if native is not None else _split_ints(record.format['RC'], 4, [','])


class GangSTRCallSpanOnly(Reason):
    """filters.py:676-697."""
    name = "GangSTRCallSpanOnly"

    def __init__(self):
        pass

    def planes(self):
        return [('__rc', _rc_plane), _field('DP')]

    def spec(self, ix):
        return dict(op=L.F_CALLED_EQ, plane_a=ix['__rc'], col_a=1, plane_b=ix['DP'], col_b=0)

    def value(self, get, l):
        return get('__rc')[l][:, 1].astype(float)


class GangSTRCallSpanBoundOnly(Reason):
    """filters.py:699-722."""
    name = "GangSTRCallSpanBoundOnly"

    def __init__(self):
        pass

    def planes(self):
        return [('__rc', _rc_plane), _field('DP')]

    def spec(self, ix):
        return dict(op=L.F_CALLED_SUM_EQ, plane_a=ix['__rc'], col_a=1, col_a2=3, plane_b=ix['DP'], col_b=0)

    def value(self, get, l):
        rc = get('__rc')[l].astype(np.int64)
        return (rc[:, 1] + rc[:, 3]).astype(float)


def _repci_plane(record):
    native = _native(record, '__repci')
    if native is not None:
        return native
    ncol = 2 * record.format['REPCN'].shape[1]
    return _split_ints(record.format['REPCI'], ncol, [',', '-'])


class GangSTRCallBadCI(Reason):
    """filters.py:724-757."""
    name = "GangSTRCallBadCI"

    def __init__(self):
        pass

    def planes(self):
        return [_field('REPCN'), ('__repci', _repci_plane)]

    def spec(self, ix):
        return dict(op=L.F_CALLED_OUTSIDE_CI, plane_a=ix['REPCN'], plane_b=ix['__repci'])

    def value(self, get, l):
        ml = get('REPCN')[l]
        ci = get('__repci')[l]
        out = np.full(ml.shape[0], np.nan)
        for j in range(ml.shape[1] - 1, -1, -1):      # first offending haplotype wins
            bad = (ml[:, j] < ci[:, 2 * j]) | (ci[:, 2 * j + 1] < ml[:, j])
            out[bad] = ml[bad, j]
        return out


class PopSTRCallRequireSupport(Reason):
    """filters.py:835-867."""
    name = "PopSTRCallRequireSupport"

    def __init__(self, threshold):
        self.threshold = threshold
        self.name += str(threshold)

    def planes(self):
        # '__ad': the batch pipeline's fixed-width native decode of AD (Number=R); the per-record loop parses the
        # record's own list, as wide as its allele list
        return [('__ad', lambda record: record.format['AD'])]

    def spec(self, ix):
        return dict(op=L.F_AD_SUPPORT_LT, plane_a=ix['__ad'], thr=self.threshold)

    def value(self, get, l):
        raise NotImplementedError  # needs the genotype indices: handled in dumpSTR.evaluate_call_filters
