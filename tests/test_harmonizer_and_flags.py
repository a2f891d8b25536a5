"""(1) The harmoniser mirror (trtools_amd/utils/tr_harmonizer.py) against what the IMPORTED
reference produced for the records of every fixture VCF (tests/golden/harmonized_records.json.gz,
tools/gen_golden_dumpstr.py::gen_harmonized; reference tr_harmonizer.py:264-550, 693-773);
(2) statSTR flag combinations (use-length, region, precision, only-passing) against tables the
imported reference wrote (tests/golden/statstr_flags/)."""
import gzip
import json
import os
import sys

import pytest

from helpers import GOLDEN

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
DATA = os.path.join(GOLDEN, 'data')


def test_harmonized_records_match_reference():
    from trtools_amd.utils import tr_harmonizer as trh, utils
    gold = json.load(gzip.open(os.path.join(GOLDEN, 'harmonized_records.json.gz'), 'rt'))['files']
    total = 0
    for rel, g in gold.items():
        reader = utils.LoadSingleReader(os.path.join(DATA, rel), checkgz=False)
        vt = g['vcftype']
        assert trh.InferVCFType(reader, vt).name == vt
        want = iter(g['records'])
        for i, r in enumerate(trh.TRRecordHarmonizer(reader, vt)):
            if i % 7 and i > 40:
                continue
            got = [r.chrom, int(r.pos), int(r.end_pos), int(r.full_alleles_pos), int(r.full_alleles_end_pos),
                   r.record_id, r.motif, r.ref_allele, list(r.alt_alleles), float(r.ref_allele_length),
                   [float(x) for x in r.alt_allele_lengths], r.full_alleles is not None, r.quality_field,
                   str(r)[:200]]
            assert got == next(want), (rel, i)
            total += 1
    assert total > 5000


def test_type_inference_and_wrong_type_errors():
    """tr_harmonizer.py:180-244 and the mandatory-field checks :316-324, 349-354, 424-427."""
    from trtools_amd.utils import tr_harmonizer as trh, utils
    hip = utils.LoadSingleReader(os.path.join(DATA, 'many_samples.vcf.gz'), checkgz=False)
    assert trh.InferVCFType(hip) == trh.VcfTypes.hipstr
    with pytest.raises(TypeError):
        trh.InferVCFType(hip, 'gangstr')
    with pytest.raises(ValueError):
        trh._ToVCFType('nonsense')
    gang = utils.LoadSingleReader(os.path.join(DATA, 'dumpSTR', 'test_gangstr.vcf.gz'), checkgz=False)
    rec = next(gang)
    with pytest.raises(TypeError):
        trh.HarmonizeRecord('hipstr', rec)      # START/END/PERIOD missing
    with pytest.raises(TypeError):
        trh.HarmonizeRecord('advntr', rec)      # VID missing
    assert trh.MayHaveImpureRepeats('hipstr') and not trh.MayHaveImpureRepeats('gangstr')
    assert trh.HasLengthRefGenotype('eh') and trh.HasLengthAltGenotypes('popstr')


def _run_stat(tmp_path, compute, name):
    import gen_golden_dumpstr as gg
    from trtools_amd import runtime
    from trtools_amd.statSTR import statSTR
    old = runtime.set_compute(compute)
    try:
        out = str(tmp_path / name)
        assert statSTR.main(gg.stat_args(out, os.path.join(DATA, 'many_samples.vcf.gz'), **gg.STAT_CASES[name])) == 0
    finally:
        runtime.set_compute(old)
    got = open(out + '.tab').read()
    want = open(os.path.join(GOLDEN, 'statstr_flags', name + '.tab')).read()
    assert got == want, name
    assert got.count('\n') > 5


@pytest.mark.parametrize("name", ['uselength', 'region', 'precision7_few', 'only_passing'])
def test_statstr_flag_combinations_cpu(tmp_path, name):
    from oracle_compute import OracleCompute
    _run_stat(tmp_path, OracleCompute(), name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ['uselength', 'region', 'precision7_few', 'only_passing'])
def test_statstr_flag_combinations_gpu(tmp_path, name):
    from trtools_amd.compute import DeviceCompute
    _run_stat(tmp_path, DeviceCompute(), name)


def _api_checks(compute):
    """TRRecord.GetAlleleCounts / GetAlleleFreqs / GetMaxAllele with every kind of ``sample_index`` numpy accepts as a
    row index (reference tr_harmonizer.py:1400-1401, 1488-1489: ``gts[sample_index, :]``) against the oracle, and the
    per-record cache: the three getters of one (record, sample_index) cost ONE pass through the compute seam."""
    import numpy as np
    from oracle import trtools_oracle as orc
    from trtools_amd import runtime
    from trtools_amd.utils import tr_harmonizer as trh, utils

    class Counting:
        def __init__(self, inner):
            self.inner, self.calls = inner, 0

        def locus_stats(self, hb, nalleles_thresh=0.01):
            self.calls += 1
            return self.inner.locus_stats(hb, nalleles_thresh=nalleles_thresh)

    seam = Counting(compute)
    old = runtime.set_compute(seam)
    try:
        reader = utils.LoadSingleReader(os.path.join(GOLDEN, 'dumpstr_synth', 'synth_hipstr.vcf'), checkgz=False)
        rng = np.random.default_rng(8)
        n_checked = 0
        for i, rec in enumerate(trh.TRRecordHarmonizer(reader, 'hipstr')):
            if i % 9:
                continue
            S = rec.GetNumSamples()
            gt = rec.vcfrecord.genotype.array()[:, :-1]
            lens = [rec.ref_allele_length] + list(rec.alt_allele_lengths)
            for si in (None, rng.random(S) < 0.6, rng.permutation(S)[:S // 2], rng.integers(0, S, size=S + 7),
                       np.array([-1, 0, 0, 0, -2, 3]), [2, 2, 2]):
                before = seam.calls
                got = rec.GetAlleleCounts(sample_index=si)
                want = orc.get_allele_counts(gt, lens, None if si is None else np.asarray(si))
                assert {float(k): int(v) for k, v in got.items()} == {float(k): int(v) for k, v in want.items()}, (i, si)
                fr = rec.GetAlleleFreqs(sample_index=si)
                assert abs(sum(fr.values()) - 1.0) < 1e-12 or not fr
                mx = rec.GetMaxAllele(sample_index=si)
                wmx = orc.get_max_allele(gt, lens, None if si is None else np.asarray(si))
                assert (np.isnan(mx) and np.isnan(wmx)) or mx == wmx
                assert seam.calls == before + 1, "three getters of one (record, sample_index): one pass"
                n_checked += 1
        assert n_checked >= 24
    finally:
        runtime.set_compute(old)


def test_sample_index_kinds_and_stats_cache_cpu():
    from oracle_compute import OracleCompute
    _api_checks(OracleCompute())


@pytest.mark.gpu
def test_sample_index_kinds_and_stats_cache_gpu():
    from trtools_amd.compute import DeviceCompute
    _api_checks(DeviceCompute())


def _fixture_files():
    import os
    from helpers import GOLDEN
    D = os.path.join(GOLDEN, 'data')
    DD = os.path.join(D, 'dumpSTR')
    return [(os.path.join(DD, 'trio_chr21_hipstr.sorted.vcf.gz'), 'hipstr'),
            (os.path.join(DD, 'trio_chr21_gangstr.sorted.vcf.gz'), 'gangstr'),
            (os.path.join(DD, 'NA12878_chr21_advntr.sorted.vcf.gz'), 'advntr'),
            (os.path.join(DD, 'NA12878_chr21_eh.sorted.vcf.gz'), 'eh'),
            (os.path.join(DD, 'NA12878_chr21_popstr.sorted.vcf.gz'), 'popstr'),
            (os.path.join(DD, 'longtr_testfile.vcf.gz'), 'longtr')]


@pytest.mark.parametrize('path,vcftype', _fixture_files(), ids=[v for _, v in _fixture_files()])
def test_native_batch_harmoniser_equals_the_python_harmoniser(path, vcftype):
    """trk_vcf_harmonize (all six callers since round 3: ExpansionHunter / PopSTR alleles are fabricated from the
    motif as utils.FabricateAllele does) against the Python harmoniser record by record: allele lengths, the
    upper-cased sequences in class order, positions, the homopolymer run of the (possibly fabricated) reference
    allele, INFO PERIOD.  Records the native one declines (symbolic <DEL> alleles of LongTR) are skipped."""
    import os
    import numpy as np
    from trtools_amd import vcfnative
    from trtools_amd.utils import tr_harmonizer as trh, utils
    if not os.path.exists(path):
        pytest.skip("fixture absent")
    r = vcfnative.NativeVCFReader(path)
    vt = trh.VcfTypes[vcftype]
    n_checked = 0
    while n_checked < 600:
        rb = r.read_raw_batch(200)
        if rb.n == 0:
            break
        hz = rb.harmonize(vcftype)
        lens, strs = hz.lists()
        for l, rec in enumerate(rb.records()):
            if hz.status[l]:
                continue
            t = trh.HarmonizeRecord(vt, rec)
            want_len = [t.ref_allele_length] + list(t.alt_allele_lengths)
            want_str = [t.ref_allele] + list(t.alt_alleles)
            assert lens[l] == [float(x) for x in want_len], (l, lens[l], want_len)
            assert strs[l] == [str(x) for x in want_str], (l, strs[l][:3], want_str[:3])
            assert int(hz.pos[l]) == rec.POS
            seq = t.full_alleles[0] if t.HasFullStringGenotypes() else t.ref_allele
            assert int(hz.hrun[l]) == utils.GetHomopolymerRun(seq), l
            per = rec.INFO.get('PERIOD')
            assert (int(hz.period[l]) == per) if per is not None else (int(hz.period[l]) == -2147483648)
            n_checked += 1
    r.close()
    assert n_checked > 20
