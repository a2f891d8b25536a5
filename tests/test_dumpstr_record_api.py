"""dumpSTR's per-record entry points under the reference's names -- ApplyCallFilters (dumpSTR.py:613-774),
ApplyLocusFilters (917-973), Check<Caller>Filters (101-394) -- driven the way the reference's own main loop
(dumpSTR.py:1270-1338) and its own filter tests (dumpSTR/tests/test_filters.py:60-200) drive them:

  * a record-by-record run over the synthetic HipSTR / GangSTR fixtures must end with the counters the REAL reference
    wrote (tests/golden/dumpstr_synth/*.samplog.tab, *.loclog.tab: tools/gen_golden_dumpstr.py) and put the same
    FORMAT/FILTER text, FILTER column and nulled genotypes on every record as the reference's output VCF holds;
  * the locus filters answer for duck-typed records (objects that only offer the TRRecord methods a filter reads) with
    the literal cases of the reference's tests.
On the CPU the statistics come through the oracle seam (tests/oracle_compute.py), on the GPU from libtrk."""
import argparse
import collections
import os
import sys

import numpy as np
import pytest

from helpers import GOLDEN

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
G = os.path.join(GOLDEN, 'dumpstr_synth')


def _record_loop(tmp_path, compute, name):
    import gen_golden_dumpstr as gg
    from trtools_amd import runtime, vcfio
    from trtools_amd.dumpSTR import dumpSTR
    from trtools_amd.utils import tr_harmonizer as trh
    caller, kw = gg.CASES[name]
    args = gg.make_args(str(tmp_path / name), os.path.join(G, 'synth_%s.vcf' % caller), caller, **kw)
    want = [ln.split('\t') for ln in open(os.path.join(G, name + '.vcf')).read().split('\n') if ln and not ln.startswith('#')]
    old = runtime.set_compute(compute)
    try:
        invcf = vcfio.VCFReader(args.vcf)
        samples = np.array(invcf.samples)
        call_filters = dumpSTR.BuildCallFilters(args)
        locus_filters = dumpSTR.BuildLocusFilters(args)
        sample_info = collections.OrderedDict([('numcalls', np.zeros(len(samples), dtype=int)),
                                               ('totaldp', np.zeros(len(samples), dtype=float))])
        for nm in dumpSTR.GetAllCallFilters(call_filters):
            sample_info[nm] = np.zeros(len(samples), dtype=int)
        loc_info = collections.OrderedDict([('totalcalls', 0), ('PASS', 0), ('NO_CALLS_REMAINING', 0)])
        for f in locus_filters:
            loc_info[f.filter_name()] = 0
        k = 0
        for record in trh.TRRecordHarmonizer(invcf, caller):
            before = record
            record = dumpSTR.ApplyCallFilters(record, call_filters, sample_info, samples)
            filtered = dumpSTR.ApplyLocusFilters(record, locus_filters, loc_info, args.drop_filtered)
            if filtered and args.drop_filtered:
                continue
            w = want[k]
            k += 1
            v = record.vcfrecord
            assert (v.CHROM, str(v.POS)) == (w[0], w[1])
            if not args.drop_filtered:
                assert (v.FILTER or 'PASS') == w[6], (name, k)
            keys = w[8].split(':')
            fi, gi = keys.index('FILTER'), keys.index('GT')
            got_text = [x.decode() if isinstance(x, bytes) else str(x) for x in np.asarray(v.format('FILTER'))]
            assert got_text == [col.split(':')[fi] for col in w[9:]], (name, k)
            # a filtered call has lost its genotype (dumpSTR.py:721-727); the record that came back reads the new one
            g = np.asarray(v.genotype.array())[:, :-1]
            for s, col in enumerate(w[9:]):
                if got_text[s] not in ('PASS', 'NOCALL'):
                    assert col.split(':')[gi] in ('.', './.', '.|.') and np.all(g[s] == -1), (name, k, s)
            assert np.array_equal(record.GetGenotypeIndicies()[:, :-1], g)
            assert record is before or record.vcfrecord is before.vcfrecord
        assert k == len(want)
        out = str(tmp_path / name)
        dumpSTR.WriteSampLog(sample_info, samples, out + '.samplog.tab')
        dumpSTR.WriteLocLog(loc_info, out + '.loclog.tab')
        for ext in ('.samplog.tab', '.loclog.tab'):
            assert open(out + ext).read() == open(os.path.join(G, name + ext)).read(), (name, ext)
    finally:
        runtime.set_compute(old)


@pytest.mark.parametrize("name", ['hipstr_all', 'hipstr_uselength_drop', 'gangstr_all'])
def test_record_loop_through_the_oracle_seam(tmp_path, name):
    from oracle_compute import OracleCompute
    _record_loop(tmp_path, OracleCompute(), name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ['hipstr_all', 'hipstr_uselength_drop', 'gangstr_all'])
def test_record_loop_on_the_device(tmp_path, name):
    from trtools_amd.compute import DeviceCompute
    _record_loop(tmp_path, DeviceCompute(), name)


# ---- duck-typed records: the cases of the reference's own locus-filter tests (test_filters.py:60-200) ----------------
class _Counts:
    """Counts everything, remembers nothing (the tests look at the return value only)."""

    def __getitem__(self, key):
        return 0

    def __setitem__(self, key, value):
        pass


class _Duck:
    def __init__(self, **methods):
        self.vcfrecord = argparse.Namespace(FILTER='')
        self.info, self.format = {}, {}
        for k, fn in methods.items():
            setattr(self, k, fn)

    def GetCalledSamples(self):
        return np.array([True, True, False])

    def GetNumSamples(self):
        return 3


def _locus_args(**kw):
    from test_dumpstr_cli import make_args
    return make_args('x', 'x', **kw)


def test_callrate_filter_on_a_duck_typed_record():
    from trtools_amd.dumpSTR.dumpSTR import ApplyLocusFilters, BuildLocusFilters
    rec = lambda: _Duck(GetCallRate=lambda: 0.5)
    r = rec()
    assert ApplyLocusFilters(r, BuildLocusFilters(_locus_args(min_locus_callrate=0.7)), _Counts(), False)
    assert r.vcfrecord.FILTER == 'CALLRATE0.7'
    r = rec()
    assert not ApplyLocusFilters(r, BuildLocusFilters(_locus_args(min_locus_callrate=0.3)), _Counts(), False)
    assert r.vcfrecord.FILTER == 'PASS'
    r = rec()
    assert ApplyLocusFilters(r, BuildLocusFilters(_locus_args(min_locus_callrate=0.7)), _Counts(), True)
    assert r.vcfrecord.FILTER == ''          # dropped records keep their column


def test_hwe_filter_on_a_duck_typed_record():
    # test_filters.py:106-152: ten genotypes over three sequences of two lengths; p = 0.21 by length, ~0.95 by sequence
    from trtools_amd.dumpSTR.dumpSTR import ApplyLocusFilters, BuildLocusFilters

    def gcounts(uselength=False):
        if uselength:
            return {(3, 3): 6, (3, 4): 2, (4, 4): 2}
        return {('ATATAT', 'ATATAT'): 2, ('ATATAT', 'ATAAAT'): 2, ('ATATAT', 'ATATATAT'): 1, ('ATAAAT', 'ATAAAT'): 2,
                ('ATAAAT', 'ATATATAT'): 1, ('ATATATAT', 'ATATATAT'): 2}

    def afreqs(uselength=False):
        return {3: .7, 4: .3} if uselength else {'ATATAT': .35, 'ATAAAT': .35, 'ATATATAT': .3}

    for thresh, passes, ul in ((0.05, True, True), (0.1, True, True), (0.3, False, True),
                               (0.05, True, False), (0.1, False, False), (0.3, False, False)):
        fs = BuildLocusFilters(_locus_args(min_locus_hwep=thresh, use_length=ul))
        assert passes != ApplyLocusFilters(_Duck(GetGenotypeCounts=gcounts, GetAlleleFreqs=afreqs), fs, _Counts(), False), (thresh, ul)


def test_het_filters_on_a_duck_typed_record():
    # test_filters.py:154-200: four alleles, two of them of one length
    from trtools_amd.dumpSTR.dumpSTR import ApplyLocusFilters, BuildLocusFilters

    def rec(c31, c32, c4, c5):
        tot = c31 + c32 + c4 + c5

        def afreqs(uselength=False):
            if uselength:
                return {3: (c31 + c32) / tot, 4: c4 / tot, 5: c5 / tot}
            return {'ATATAT': c31 / tot, 'ATAAAT': c32 / tot, 'ATATATAT': c4 / tot, 'ATATATATAT': c5 / tot}
        return _Duck(GetAlleleFreqs=afreqs)

    for freqs, thresh, higher, ul in (([.25] * 4, 0.7, True, False), ([.25] * 4, 0.7, False, True), ([.25] * 4, 0.8, False, False),
                                      ([.5, .5, 0, 0], 0.4, True, False), ([.5, .5, 0, 0], 0.4, False, True)):
        lo = BuildLocusFilters(_locus_args(min_locus_het=thresh, use_length=ul))
        hi = BuildLocusFilters(_locus_args(max_locus_het=thresh, use_length=ul))
        assert higher != ApplyLocusFilters(rec(*freqs), lo, _Counts(), False), (freqs, thresh, ul)
        assert higher == ApplyLocusFilters(rec(*freqs), hi, _Counts(), False), (freqs, thresh, ul)


def test_no_calls_remaining_and_counters():
    from trtools_amd.dumpSTR.dumpSTR import ApplyLocusFilters
    loc = collections.defaultdict(int)
    r = _Duck()
    r.GetCalledSamples = lambda: np.array([False, False, False])
    assert ApplyLocusFilters(r, [], loc, False) and r.vcfrecord.FILTER == 'NO_CALLS_REMAINING'
    r2 = _Duck()
    assert not ApplyLocusFilters(r2, [], loc, False) and r2.vcfrecord.FILTER == 'PASS'
    assert dict(loc) == {'NO_CALLS_REMAINING': 1, 'PASS': 1, 'totalcalls': 2}


def test_per_caller_checks():
    # the validators under the reference's names: ranges and min <= max, a WARNING and False otherwise
    from trtools_amd.dumpSTR import dumpSTR
    import gen_golden_dumpstr as gg
    a = gg.make_args('o', 'v', 'hipstr', hipstr_min_call_DP=10, hipstr_max_call_DP=50, hipstr_min_call_Q=0.9)
    assert dumpSTR.CheckHipSTRFilters({'DP', 'Q'}, a)
    a.hipstr_max_call_DP = 5
    assert not dumpSTR.CheckHipSTRFilters({'DP', 'Q'}, a)
    a = gg.make_args('o', 'v', 'hipstr', hipstr_max_call_stutter=1.5)
    assert not dumpSTR.CheckHipSTRFilters({'DP', 'DSTUTTER'}, a)
    assert dumpSTR.CheckGangSTRFilters({'DP', 'Q', 'QEXP'}, gg.make_args('o', 'v', 'gangstr', gangstr_expansion_prob_het=0.5))
    assert not dumpSTR.CheckGangSTRFilters({'DP'}, gg.make_args('o', 'v', 'gangstr', gangstr_min_call_DP=-1))
    assert not dumpSTR.CheckLongTRFilters({'DP'}, gg.make_args('o', 'v', 'longtr', longtr_min_call_DP=10, longtr_max_call_DP=2))
    assert not dumpSTR.CheckAdVNTRFilters({'ML'}, gg.make_args('o', 'v', 'advntr', advntr_min_ML=-0.1))
    assert not dumpSTR.CheckEHFilters({'LC'}, gg.make_args('o', 'v', 'eh', eh_min_call_LC=-3))
    assert dumpSTR.CheckPopSTRFilters({'DP', 'AD'}, gg.make_args('o', 'v', 'popstr', popstr_require_support=2))
    with pytest.raises(AssertionError):       # an option whose FORMAT field the VCF lacks: the reference asserts
        dumpSTR.CheckPopSTRFilters({'DP'}, gg.make_args('o', 'v', 'popstr', popstr_require_support=2))
