"""The batch pipelines of the two command lines (native reader -> native batch harmoniser -> compute seam -> native
row / record writer, no Python object per record) against the per-record loops they replace: byte-identical outputs
on the reference's fixture VCFs and on the synthetic HipSTR / GangSTR files, for several argument sets each.  The
per-record loops are what the golden-file tests pin (tests/test_statstr_cli.py, test_dumpstr_cli.py, ...); this file
pins the batch pipelines to them.  CPU: oracle-backed compute stand-in; the GPU legs run the same comparison through
libtrk."""
import argparse
import os

import pytest

from helpers import GOLDEN, lab_env
from test_dumpstr_cli import make_args as dump_args

D = os.path.join(GOLDEN, 'data')
DD = os.path.join(D, 'dumpSTR')
SYN = os.path.join(GOLDEN, 'dumpstr_synth')

STAT_FILES = [
    (os.path.join(D, 'many_samples.vcf.gz'), 'hipstr'),
    (os.path.join(DD, 'trio_chr21_hipstr.sorted.vcf.gz'), 'hipstr'),
    (os.path.join(DD, 'trio_chr21_gangstr.sorted.vcf.gz'), 'gangstr'),
    (os.path.join(DD, 'test_gangstr.vcf.gz'), 'gangstr'),
    (os.path.join(DD, 'NA12878_chr21_advntr.sorted.vcf.gz'), 'advntr'),
    (os.path.join(DD, 'longtr_testfile.vcf.gz'), 'longtr'),
    (os.path.join(SYN, 'synth_hipstr.vcf'), 'hipstr'),
    (os.path.join(SYN, 'synth_gangstr.vcf'), 'gangstr'),
]


def _stat_args(vcf, out, vcftype, **kw):
    ns = argparse.Namespace(vcf=vcf, out=out, vcftype=vcftype, samples=None, sample_prefixes=None, plot_afreq=False,
                            region=None, thresh=True, afreq=True, acount=True, hwep=False, het=True, entropy=True,
                            mean=True, mode=True, var=True, numcalled=True, use_length=False, precision=4, nalleles=True,
                            nalleles_thresh=0.05, only_passing=False)
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


def _both(fn, env):
    outs = []
    for mode in ('1', '0'):
        os.environ[env] = mode
        try:
            outs.append(fn(mode))
        finally:
            del os.environ[env]
    return outs


def _run_stat(tmp_path, compute, path, vcftype, **kw):
    from trtools_amd import runtime
    from trtools_amd.statSTR import statSTR
    taken = []
    orig = statSTR._run_batches

    def spy(*a, **k):
        r = orig(*a, **k)
        taken.append(r)
        return r
    statSTR._run_batches = spy
    old = runtime.set_compute(compute)
    try:
        def go(mode):
            out = str(tmp_path / ('s' + mode))
            assert statSTR.main(_stat_args(path, out, vcftype, **kw)) == 0
            return open(out + '.tab').read()
        a, b = _both(go, 'TRK_STATSTR_BATCH')
    finally:
        runtime.set_compute(old)
        statSTR._run_batches = orig
    assert taken and taken[0] is not None and taken[0] > 0, "the batch pipeline did not run"
    assert a == b
    return a


# the oracle-backed stand-in takes a second per hundred records: the CPU legs use the small inputs, the GPU legs all
SMALL_STAT = [f for f in STAT_FILES if 'trio' not in f[0] and 'many_samples' not in f[0]]


@pytest.mark.parametrize('path,vcftype', SMALL_STAT, ids=[os.path.basename(p) for p, _ in SMALL_STAT])
def test_statstr_batch_pipeline_equals_per_record_loop(tmp_path, path, vcftype):
    from oracle_compute import OracleCompute
    hwep = vcftype in ('hipstr',) and 'trio' not in path
    t = _run_stat(tmp_path, OracleCompute(), path, vcftype, hwep=hwep)
    assert t.count('\n') > 5
    _run_stat(tmp_path, OracleCompute(), path, vcftype, use_length=True, precision=6, only_passing=True, acount=False)


def test_statstr_batch_pipeline_with_sample_groups(tmp_path):
    from oracle_compute import OracleCompute
    from trtools_amd import vcfio
    names = vcfio.VCFReader(os.path.join(SYN, 'synth_hipstr.vcf')).samples
    for i, sel in enumerate((names[::2], names[1::3])):
        (tmp_path / ('g%d.txt' % i)).write_text('\n'.join(sel) + '\n')
    _run_stat(tmp_path, OracleCompute(), os.path.join(SYN, 'synth_hipstr.vcf'), 'hipstr', hwep=True,
              samples=str(tmp_path / 'g0.txt') + ',' + str(tmp_path / 'g1.txt'), sample_prefixes='a,b')


DUMP_CASES = {
    'synth_hipstr_all': (os.path.join(SYN, 'synth_hipstr.vcf'),
                         dict(vcftype='hipstr', hipstr_max_call_flank_indel=0.05, hipstr_max_call_stutter=0.3,
                              hipstr_min_supp_reads=10, hipstr_min_call_DP=20, hipstr_max_call_DP=50,
                              hipstr_min_call_Q=0.9, use_length=True, min_locus_hwep=0.01, min_locus_callrate=0.5)),
    'hipstr_thresholds': (os.path.join(DD, 'trio_chr21_hipstr.sorted.vcf.gz'),
                          dict(vcftype='hipstr', hipstr_min_call_DP=20, hipstr_max_call_DP=60, hipstr_min_call_Q=0.9,
                               min_locus_callrate=0.7, min_locus_het=0.05, max_locus_het=0.6, num_records=None)),
    'hipstr_ratios_minsupp': (os.path.join(DD, 'trio_chr21_hipstr.sorted.vcf.gz'),
                              dict(vcftype='hipstr', hipstr_max_call_flank_indel=0.05, hipstr_max_call_stutter=0.3,
                                   hipstr_min_supp_reads=10, hipstr_min_call_DP=30, use_length=True,
                                   min_locus_hwep=0.01)),
    'hipstr_drop_filtered': (os.path.join(SYN, 'synth_hipstr.vcf'),
                             dict(vcftype='hipstr', hipstr_min_call_DP=25, hipstr_min_call_Q=0.95,
                                  min_locus_callrate=0.9, drop_filtered=True)),
    'gangstr_thresholds': (os.path.join(DD, 'trio_chr21_gangstr.sorted.vcf.gz'),
                           dict(vcftype='gangstr', gangstr_min_call_DP=10, gangstr_max_call_DP=100,
                                gangstr_min_call_Q=0.9, min_locus_callrate=0.6)),
    'synth_gangstr': (os.path.join(SYN, 'synth_gangstr.vcf'),
                      dict(vcftype='gangstr', gangstr_min_call_DP=15, gangstr_min_call_Q=0.92, max_locus_het=0.7)),
    # every GangSTR call filter (dumpSTR.py:819-836): expansion probabilities (single column and the float32 sum of two),
    # span-only / span+bound-only on the pre-parsed RC field, the confidence-interval filter on REPCN / REPCI
    'synth_gangstr_all': (os.path.join(SYN, 'synth_gangstr.vcf'),
                          dict(vcftype='gangstr', gangstr_min_call_DP=10, gangstr_max_call_DP=60, gangstr_min_call_Q=0.9,
                               gangstr_expansion_prob_het=0.3, gangstr_expansion_prob_hom=0.3,
                               gangstr_expansion_prob_total=0.6, gangstr_filter_span_only=True,
                               gangstr_filter_spanbound_only=True, gangstr_filter_badCI=True, min_locus_callrate=0.3)),
    'gangstr_all_trio': (os.path.join(DD, 'trio_chr21_gangstr.sorted.vcf.gz'),
                         dict(vcftype='gangstr', gangstr_expansion_prob_het=0.1, gangstr_expansion_prob_hom=0.1,
                              gangstr_expansion_prob_total=0.3, gangstr_filter_span_only=True,
                              gangstr_filter_spanbound_only=True, gangstr_filter_badCI=True, gangstr_min_call_DP=15)),
    'no_call_filters': (os.path.join(SYN, 'synth_hipstr.vcf'), dict(vcftype='hipstr', min_locus_callrate=0.95)),
    'longtr': (os.path.join(DD, 'longtr_testfile.vcf.gz'),
               dict(vcftype='longtr', longtr_min_call_DP=30, longtr_max_call_DP=200, longtr_min_call_Q=0.9,
                    use_length=True, min_locus_het=0.05)),
}


# inputs the native pieces decline (records without the mandatory INFO fields, a FORMAT/FILTER field from an earlier
# dumpSTR round): the batch goes through the record objects -- same outputs, just not through the batch pipeline
FALLBACK_CASES = set()      # (round 4: LongTR's symbolic '<DEL>' alleles are harmonised natively too)
BIG_CASES = {'hipstr_thresholds', 'hipstr_ratios_minsupp', 'gangstr_thresholds', 'gangstr_all_trio'}     # trio files: GPU legs only


def _run_dump(tmp_path, compute, name):
    from trtools_amd import runtime
    from trtools_amd.dumpSTR import dumpSTR
    path, kw = DUMP_CASES[name]
    taken = []
    orig = dumpSTR._Run.process_raw

    def spy(self, *a, **k):
        r = orig(self, *a, **k)
        taken.append(r)
        return r
    dumpSTR._Run.process_raw = spy
    old = runtime.set_compute(compute)
    try:
        def go(mode):
            out = str(tmp_path / ('d' + mode))
            assert dumpSTR.main(dump_args(out, path, **kw)) == 0
            return tuple(open(out + ext).read() for ext in ('.vcf', '.samplog.tab', '.loclog.tab'))
        a, b = _both(go, 'TRK_DUMPSTR_BATCH')
    finally:
        runtime.set_compute(old)
        dumpSTR._Run.process_raw = orig
    if name not in FALLBACK_CASES:
        assert taken and all(taken), "the batch pipeline did not handle every batch"
    for x, y, what in zip(a, b, ('vcf', 'samplog', 'loclog')):
        if x != y:
            la, lb = x.split('\n'), y.split('\n')
            i = next(i for i, (p, q) in enumerate(zip(la, lb)) if p != q)
            raise AssertionError("%s: %s differs at line %d:\n%s\n%s" % (name, what, i, la[i][:400], lb[i][:400]))
    return a


@pytest.mark.parametrize('name', sorted(set(DUMP_CASES) - BIG_CASES))
def test_dumpstr_batch_pipeline_equals_per_record_loop(tmp_path, name):
    from oracle_compute import OracleCompute
    vcf, _, loclog = _run_dump(tmp_path, OracleCompute(), name)
    assert vcf.count('\n') > 20 and 'PASS' in loclog


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(DUMP_CASES))
def test_dumpstr_batch_pipeline_gpu(tmp_path, name):
    from trtools_amd import runtime
    _run_dump(tmp_path, runtime.get_compute(), name)


@pytest.mark.gpu
@pytest.mark.parametrize('path,vcftype', STAT_FILES, ids=[os.path.basename(p) for p, _ in STAT_FILES])
def test_statstr_batch_pipeline_gpu(tmp_path, path, vcftype):
    from trtools_amd import runtime
    _run_stat(tmp_path, runtime.get_compute(), path, vcftype, hwep=('many_samples' in path or 'synth_hipstr' in path))


def _strip_format_key(src, dst, key, which):
    """Copy of the text VCF ``src`` in which the records ``which`` (0-based indices) lose FORMAT key ``key``."""
    import gzip
    op = gzip.open if src.endswith('.gz') else open
    n = -1
    with op(src, 'rt') as fin, open(dst, 'w') as fout:
        for line in fin:
            if line.startswith('#'):
                fout.write(line)
                continue
            n += 1
            if n in which:
                f = line.rstrip('\n').split('\t')
                keys = f[8].split(':')
                i = keys.index(key)
                f[8] = ':'.join(k for k in keys if k != key)
                f[9:] = [':'.join(v for j, v in enumerate(s.split(':')) if j != i) if ':' in s else s for s in f[9:]]
                line = '\t'.join(f) + '\n'
            fout.write(line)


@pytest.mark.parametrize('which', [{0}, {7}], ids=['first_record', 'later_record'])
def test_dumpstr_record_without_a_filtered_format_key_fails_as_the_reference_does(tmp_path, which):
    """a HipSTR record without DP under --hipstr-min-call-DP.  The reference raises KeyError at
    `record.format[self.field]` (filters.py:327-409); the batch pipeline used to fill the plane with the missing marker,
    null every call and flag the locus NO_CALLS_REMAINING.  It must decline the batch, and the per-record loop must
    then fail exactly as with TRK_DUMPSTR_BATCH=0."""
    from oracle_compute import OracleCompute
    from trtools_amd import runtime
    from trtools_amd.dumpSTR import dumpSTR
    src = os.path.join(SYN, 'synth_hipstr.vcf')
    vcf = str(tmp_path / 'nodp.vcf')
    _strip_format_key(src, vcf, 'DP', which)
    old = runtime.set_compute(OracleCompute())
    try:
        outcomes = []
        for mode in ('1', '0'):
            os.environ['TRK_DUMPSTR_BATCH'] = mode
            try:
                rc = dumpSTR.main(dump_args(str(tmp_path / ('o' + mode)), vcf, vcftype='hipstr', hipstr_min_call_DP=20))
                outcomes.append(('rc', rc))
            except Exception as e:      # noqa: BLE001 -- the two modes must fail the same way
                outcomes.append((type(e).__name__, str(e)))
            finally:
                del os.environ['TRK_DUMPSTR_BATCH']
    finally:
        runtime.set_compute(old)
    assert outcomes[0] == outcomes[1], outcomes
    assert outcomes[0][0] == 'KeyError' and 'DP' in outcomes[0][1], outcomes


def test_dumpstr_depth_field_is_judged_per_record(tmp_path):
    """dumpSTR.py:688-695 looks for DP (else LC) in EVERY record: a file in which some records lack DP and no filter
    reads it gives the same logs through the batch pipeline (which hands such a batch to the per-record loop) and
    through the loop itself."""
    from oracle_compute import OracleCompute
    from trtools_amd import runtime
    from trtools_amd.dumpSTR import dumpSTR
    src = os.path.join(SYN, 'synth_hipstr.vcf')
    vcf = str(tmp_path / 'somedp.vcf')
    _strip_format_key(src, vcf, 'DP', {3, 4, 11})
    old = runtime.set_compute(OracleCompute())
    try:
        def go(mode):
            out = str(tmp_path / ('p' + mode))
            assert dumpSTR.main(dump_args(out, vcf, vcftype='hipstr', hipstr_min_call_Q=0.9)) == 0
            return tuple(open(out + ext).read() for ext in ('.vcf', '.samplog.tab', '.loclog.tab'))
        a, b = _both(go, 'TRK_DUMPSTR_BATCH')
    finally:
        runtime.set_compute(old)
    assert a == b


def test_read_ahead_leaves_both_command_lines_outputs_alone(tmp_path, monkeypatch):
    """TRK_VCF_READ_AHEAD=1: the reader's worker thread reads batch n + 1 while batch n is processed (small batches
    here, so that several are in flight over the file); outputs equal the per-record loops' as without it."""
    from oracle_compute import OracleCompute
    from trtools_amd.statSTR import statSTR
    from trtools_amd.dumpSTR import dumpSTR
    monkeypatch.setenv('TRK_VCF_READ_AHEAD', '1')
    monkeypatch.setattr(statSTR, 'BATCH_CELLS', 40 * 7, raising=False)
    monkeypatch.setattr(dumpSTR, 'BATCH_CELLS', 40 * 7, raising=False)
    _run_stat(tmp_path, OracleCompute(), os.path.join(SYN, 'synth_hipstr.vcf'), 'hipstr', hwep=True)
    name = sorted(set(DUMP_CASES) - BIG_CASES - FALLBACK_CASES)[0]
    _run_dump(tmp_path, OracleCompute(), name)


def test_plot_afreq_keeps_the_table_on_the_batch_pipeline(tmp_path):
    """--plot-afreq (statSTR.py:603-607): the first eleven reported records are plotted from a short read of their
    own; the table comes from the batch pipeline and equals the one of a run without plots."""
    pytest.importorskip('matplotlib')
    import glob
    from oracle_compute import OracleCompute
    from trtools_amd import runtime
    from trtools_amd.statSTR import statSTR
    path = os.path.join(SYN, 'synth_hipstr.vcf')
    old = runtime.set_compute(OracleCompute())
    try:
        plain = str(tmp_path / 'plain')
        assert statSTR.main(_stat_args(path, plain, 'hipstr')) == 0
        out = str(tmp_path / 'plots')
        assert statSTR.main(_stat_args(path, out, 'hipstr', plot_afreq=True)) == 0
        assert statSTR.LAST_RUN['path'] == 'batch'
    finally:
        runtime.set_compute(old)
    assert open(out + '.tab').read() == open(plain + '.tab').read()
    n_records = open(plain + '.tab').read().count('\n') - 1
    assert len(glob.glob(out + '-*.pdf')) == min(11, n_records) > 0


def _flank_subset(tmp_path, n_flank=12, n_plain=12):
    """A text VCF with the header of the HipSTR trio file, ``n_flank`` of its records whose alleles carry flanking
    bases (INFO START != POS) and ``n_plain`` without, in file order; and a bgzipped + indexed BED file holding, for
    every flank record with START > POS, exactly the bases [POS, START) -- left of the harmonised record."""
    import gzip
    from trtools_amd import bgzf, tabix
    src = os.path.join(DD, 'trio_chr21_hipstr.sorted.vcf.gz')
    vcf = str(tmp_path / 'flanks.vcf')
    bed = str(tmp_path / 'flanks.bed.gz')
    flank_pos, iv, nf, npl = [], [], 0, 0
    with gzip.open(src, 'rt') as fin, open(vcf, 'w') as fout:
        for line in fin:
            if line.startswith('#'):
                fout.write(line)
                continue
            f = line.split('\t', 8)
            pos = int(f[1])
            start = int([t for t in f[7].split(';') if t.startswith('START=')][0][6:])
            if start > pos and nf < n_flank:
                nf += 1
                flank_pos.append(start)
                iv.append((f[0], pos - 1, start - 1))
                fout.write(line)
            elif start == pos and npl < n_plain:
                npl += 1
                fout.write(line)
            if nf == n_flank and npl == n_plain:
                break
    assert nf >= 4, "the fixture lost its flank-carrying records"
    with bgzf.BgzfWriter(bed) as w:
        w.write(''.join('%s\t%d\t%d\n' % t for t in iv).encode())
    open(bed + '.tbi', 'wb').close()       # the filter only requires the index to exist (filters.py:206-217)
    return vcf, bed, flank_pos


def _check_flank_regions(tmp_path, compute):
    from trtools_amd import runtime
    from trtools_amd.dumpSTR import dumpSTR
    vcf, bed, flank_pos = _flank_subset(tmp_path)
    old = runtime.set_compute(compute)
    try:
        def go(mode):
            out = str(tmp_path / ('r' + mode))
            assert dumpSTR.main(dump_args(out, vcf, vcftype='hipstr', filter_regions=bed, filter_regions_names='flank',
                                          hipstr_min_call_DP=5)) == 0
            assert (dumpSTR.LAST_RUN['path'] == 'batch') == (mode == '1')
            return tuple(open(out + ext).read() for ext in ('.vcf', '.samplog.tab', '.loclog.tab'))
        a, b = _both(go, 'TRK_DUMPSTR_BATCH')
    finally:
        runtime.set_compute(old)
    assert a == b
    # TRRecord.pos is INFO START for these records (tr_harmonizer.py:407): the bases left of START are not the record
    recs = [l.split('\t', 8) for l in a[0].split('\n') if l and not l.startswith('#')]
    hit = [r for r in recs if 'flank' in r[6]]
    starts = {int([t for t in r[7].split(';') if t.startswith('START=')][0][6:]): r for r in recs}
    assert all('flank' not in starts[p][6] for p in flank_pos if p in starts), [r[:7] for r in hit]


def test_dumpstr_region_filter_reads_the_harmonised_position(tmp_path):
    """--filter-regions in the batch pipeline tested [POS, POS + ref_len) where the reference and
    the per-record loop test [TRRecord.pos, ...) = INFO START for HipSTR records with flanking bases."""
    from oracle_compute import OracleCompute
    _check_flank_regions(tmp_path, OracleCompute())


@pytest.mark.gpu
def test_dumpstr_region_filter_reads_the_harmonised_position_gpu(tmp_path):
    from trtools_amd import runtime
    _check_flank_regions(tmp_path, runtime.get_compute())


# ---- round 4: native record heads + the sample columns written without decoding (trk_vcf_dumpstr_records) ----------
def _dump_variants(tmp_path, name, envs):
    """dumpSTR's batch pipeline on DUMP_CASES[name] under each environment of ``envs``: outputs + writer statistics."""
    from oracle_compute import OracleCompute
    from trtools_amd import runtime, vcfnative
    from trtools_amd.dumpSTR import dumpSTR
    path, kw = DUMP_CASES[name]
    old = runtime.set_compute(OracleCompute())
    outs = []
    try:
        for i, env in enumerate(envs):
            with lab_env(**env):
                before = vcfnative.dumpstr_writer_stats()
                out = str(tmp_path / ('v%d' % i))
                assert dumpSTR.main(dump_args(out, path, **kw)) == 0
                after = vcfnative.dumpstr_writer_stats()
                outs.append((tuple(open(out + ext).read() for ext in ('.vcf', '.samplog.tab', '.loclog.tab')),
                             {k: after[k] - before[k] for k in after}, dumpSTR.LAST_RUN['path']))
    finally:
        runtime.set_compute(old)
    return outs


@pytest.mark.parametrize('name', ['synth_hipstr_all', 'synth_gangstr_all', 'hipstr_drop_filtered', 'no_call_filters'])
def test_dumpstr_native_heads_and_undecoded_samples_equal_the_decode_path(tmp_path, name):
    """The three writers of the batch pipeline give the same bytes: (a) heads built natively + sample columns copied
    token by token (the default), (b) native heads + decode -> null -> format (TRK_FMT_FAST=0), (c) heads from Python
    (round 3's path, TRK_DUMPSTR_NATIVE_HEADS=0) -- and (a) really writes its records without decoding them."""
    a, b, c = _dump_variants(tmp_path, name, [{}, {'TRK_FMT_FAST': '0'}, {'TRK_DUMPSTR_NATIVE_HEADS': '0'}])
    assert a[2] == b[2] == c[2] == 'batch'
    assert a[0] == b[0] == c[0]
    n_out = sum(1 for ln in a[0][0].split('\n') if ln and not ln.startswith('#'))
    assert a[1]['fast'] == n_out and a[1]['decoded'] == 0, a[1]
    if name != 'hipstr_drop_filtered':
        assert n_out > 20
    assert b[1]['fast'] == 0 and b[1]['decoded'] == a[1]['fast']
    assert a[1]['caller_heads'] == 0 and c[1]['fast'] == 0


def _mutated_hipstr(tmp_path, seed, scalar=False):
    """synth_hipstr.vcf with its sample columns rewritten at random: numbers spelled the ways a decoder normalises
    ('007', '+5', '1.00', '0.50', '1e-3', seven digits), vector fields of equal and of ragged length, missing values,
    samples that stop early, a string field; plus INFO values of every declared type.  ``scalar``: no vector fields
    (the records the span writer's scalar tier takes, round 4): most records keep canonical numbers only, one in
    three gets the other spellings, tokens stop early or are a lone '.'."""
    import random
    rnd = random.Random(seed)
    src = os.path.join(SYN, 'synth_hipstr.vcf')
    dst = str(tmp_path / ('mut%d%s.vcf' % (seed, 's' if scalar else '')))
    if scalar:
        canon_i, canon_f = ['7', '12', '0', '-3', '31', '123456789', '.'], ['0.99', '1', '0.5', '0.0001', '0.00012345', '12.5', '999999', '-0', '.', '0.95', '123.456']
        odd_i, odd_f = ['007', '-0', '2147483000', '1234567890'], ['1.00', '0.50', '1e-3', '123456.7', '0.1234567', '0.0', '1E2', '1000000', '.5', '0.00001']
        with open(src) as fin, open(dst, 'w') as fout:
            for line in fin:
                if line.startswith('#'):
                    fout.write(line)
                    continue
                f = line.rstrip('\n').split('\t')
                keys = f[8].split(':')
                odd = rnd.random() < 0.33
                for i in range(9, len(f)):
                    t = f[i].split(':')
                    if len(t) != len(keys) or t[0] in ('.', './.', '.|.'):
                        continue
                    for k in ('DP', 'DSTUTTER', 'DFLANKINDEL'):
                        if rnd.random() < 0.4:
                            t[keys.index(k)] = rnd.choice(canon_i + (odd_i if odd and k != 'DP' else []))
                    if rnd.random() < 0.3:
                        t[keys.index('DP')] = rnd.choice(['11', '22', '45', '60'] + (['007', '033'] if odd else []))
                    if rnd.random() < 0.5:
                        t[keys.index('Q')] = rnd.choice(['0.99', '0.9', '0.5', '1', '0.951'] + (odd_f if odd else []))
                    r = rnd.random()
                    if r < 0.05:
                        t = t[:rnd.randint(1, len(t))]                  # trailing fields dropped
                    elif r < 0.08:
                        t = ['.']                                       # HipSTR's sample without a call
                    elif r < 0.09 and odd:
                        t = t + ['extra']                               # more fields than keys
                    f[i] = ':'.join(t)
                fout.write('\t'.join(f) + '\n')
        return dst
    ints = ['7', '007', '12', '0', '-3', '-0', '123456789', '2147483000', '.', '31']   # ('+5': the per-record path's)
    floats = ['0.99', '1.00', '0.50', '1', '0.5', '1e-3', '0.0001', '0.00012345', '123456.7', '0.1234567', '-0', '0.0', '.', '12.5',
              '1E2', '999999', '1000000', '0.95']
    extra_hdr = ['##FORMAT=<ID=PQ2,Number=2,Type=Float,Description="x">\n', '##FORMAT=<ID=AL,Number=.,Type=Integer,Description="x">\n',
                 '##FORMAT=<ID=TAG,Number=1,Type=String,Description="x">\n',
                 '##INFO=<ID=FQ,Number=1,Type=Float,Description="x">\n', '##INFO=<ID=FLG,Number=0,Type=Flag,Description="x">\n',
                 '##INFO=<ID=NV,Number=.,Type=Integer,Description="x">\n']
    with open(src) as fin, open(dst, 'w') as fout:
        for line in fin:
            if line.startswith('#CHROM'):
                fout.write(''.join(extra_hdr))
            if line.startswith('#'):
                fout.write(line)
                continue
            f = line.rstrip('\n').split('\t')
            f[7] += ';FQ=%s;NV=%s%s;OTHER=x,y' % (rnd.choice(['0.50', '1.25', '3', '1e-2', '.']),
                                                rnd.choice(['1,2', '007', '+4,.', '5']), rnd.choice(['', ';FLG', ';FLG=1']))
            keys = f[8].split(':')
            ragged_al, ragged_pq = rnd.random() < 0.3, rnd.random() < 0.2
            f[8] = ':'.join(keys + ['PQ2', 'AL', 'TAG'])
            for i in range(9, len(f)):
                t = f[i].split(':')
                if len(t) == len(keys) and t[0] not in ('.', './.', '.|.'):
                    if rnd.random() < 0.5:
                        t[keys.index('DP')] = rnd.choice(['11', '22', '45', '007', '033', '60'])
                    if rnd.random() < 0.5:
                        t[keys.index('Q')] = rnd.choice(['0.99', '1.00', '0.90', '0.5', '1', '0.951', '0.9512345'])
                pq = ','.join(rnd.choice(floats) for _ in range(rnd.choice([1, 2]) if ragged_pq else 2))
                al = ','.join(rnd.choice(ints) for _ in range(rnd.choice([1, 2, 3]) if ragged_al else 2))
                t += [pq, al, rnd.choice(['a', 'bc', '.', 'x|y;z'])]
                if rnd.random() < 0.08:
                    t = t[:rnd.randint(max(1, len(keys)), len(t))]         # trailing fields dropped
                f[i] = ':'.join(t)
            fout.write('\t'.join(f) + '\n')
    return dst


@pytest.mark.parametrize('seed,scalar', [(1, False), (2, False), (3, False), (4, False), (5, True), (6, True), (7, True), (8, True)])
def test_undecoded_sample_columns_on_rewritten_text(tmp_path, seed, scalar):
    """The span writer (fast_samples, and its scalar tier fast_samples_scalar on the records without vector fields)
    against decode -> null -> format on text it has to work for: every record comes out the same bytes whichever
    writer took it (scalar tier, general transducer alone, decode path, Python heads), and the native INFO rewrite
    equals vcfio.rewrite_info."""
    from oracle_compute import OracleCompute
    from trtools_amd import runtime, vcfnative
    from trtools_amd.dumpSTR import dumpSTR
    vcf = _mutated_hipstr(tmp_path, seed, scalar)
    old = runtime.set_compute(OracleCompute())
    outs, stats = [], []
    try:
        for i, env in enumerate([{}, {'TRK_FMT_FAST': '0'}, {'TRK_DUMPSTR_NATIVE_HEADS': '0'}, {'TRK_FMT_SCALAR': '0'}]):
            with lab_env(**env):
                before = vcfnative.dumpstr_writer_stats()
                out = str(tmp_path / ('w%d' % i))
                assert dumpSTR.main(dump_args(out, vcf, vcftype='hipstr', hipstr_min_call_DP=20, hipstr_max_call_DP=50,
                                              hipstr_min_call_Q=0.9, min_locus_callrate=0.2)) == 0
                assert dumpSTR.LAST_RUN['path'] == 'batch'
                after = vcfnative.dumpstr_writer_stats()
                outs.append(open(out + '.vcf').read())
                stats.append({k: after[k] - before[k] for k in after})
    finally:
        runtime.set_compute(old)
    for x, what in ((outs[1], 'decode path'), (outs[2], 'Python heads'), (outs[3], 'general transducer')):
        if x != outs[0]:
            la, lb = outs[0].split('\n'), x.split('\n')
            i = next(i for i, (p, q) in enumerate(zip(la, lb)) if p != q)
            fa, fb = la[i].split('\t'), lb[i].split('\t')
            j = next(j for j, (p, q) in enumerate(zip(fa, fb)) if p != q)
            raise AssertionError("line %d column %d differs from the %s:\n%s\n%s" % (i, j, what, fa[j][:200], fb[j][:200]))
    assert stats[0]['fast'] > 0, stats        # some records are taken by the span writer, the ragged ones by the decoder
    assert stats[0]['fast'] + stats[0]['decoded'] == stats[1]['decoded']


def test_writer_takes_the_sample_columns_from_the_caller_and_waits_for_them(tmp_path, monkeypatch):
    """trk_vcf_dumpstr2.dev_regions / dev_wait without a GPU: the columns of every output record are handed to the native
    writer as the device path would (here cut from a first run's own output), one record in five left to the writer
    (dev_flags), and the copy 'completes' inside dev_wait -- the buffer holds garbage until the writer calls it, which it
    must do after the heads and before the first byte is read.  Same bytes as the first run; a failing wait fails the
    batch (the pipeline falls back to its other writer)."""
    import ctypes as C
    import numpy as np
    from trtools_amd import vcfnative
    (ref, stats, path), = _dump_variants(tmp_path, 'synth_hipstr_all', [{}])
    cols = {}
    for ln in ref[0].split('\n'):
        if ln and not ln.startswith('#'):
            f = ln.split('\t')
            cols[(f[0], f[1])] = ('\t' + '\t'.join(f[9:])).encode()
    assert len(cols) > 20
    state = {'calls': 0, 'pending': [], 'fail': False, 'taken': 0}

    @C.CFUNCTYPE(C.c_int, C.c_void_p)
    def wait(_arg):
        state['calls'] += 1
        for buf, good in state['pending']:
            buf[:good.size] = good
        state['pending'] = []
        return 1 if state['fail'] else 0

    def fake_regions(self, prm, mask, cf_values, S, out_ring, dev_call=None, cf_plane_idx=None):
        n = self.n
        lo = np.ctypeslib.as_array(self.b.line_off, shape=(n,))
        le = np.ctypeslib.as_array(self.b.line_end, shape=(n,))
        parts, off, ln, fl = [], np.zeros(n, np.int64), np.zeros(n, np.uint32), np.ones(n, np.uint8)
        at = 0
        for l in range(n):
            head = C.string_at(self.b.text + int(lo[l]), min(int(le[l] - lo[l]), 200)).split(b'\t')
            c = cols.get((head[0].decode(), head[1].decode()))
            off[l] = at
            if c is not None and l % 5 != 4:
                parts.append(c)
                ln[l], fl[l] = len(c), 0
                at += len(c)
                state['taken'] += 1
        good = np.frombuffer(b''.join(parts) + b'\0', dtype=np.uint8).copy()
        buf = np.full(good.size, ord('#'), dtype=np.uint8)          # not there yet
        state['pending'].append((buf, good))
        return dict(buf=buf, off=off, len=ln, flags=fl, wait=(C.cast(wait, C.c_void_p).value, None), held=None)

    monkeypatch.setattr(vcfnative.RawBatch, '_device_regions', fake_regions)
    (got, stats2, path2), = _dump_variants(tmp_path, 'synth_hipstr_all', [{}])
    assert path2 == 'batch' and got == ref
    assert state['calls'] >= 1 and state['taken'] > 20 and not state['pending']
    state['fail'] = True
    (got3, _, path3), = _dump_variants(tmp_path, 'synth_hipstr_all', [{}])
    assert got3 == ref           # (the batch went to the writer that needs no columns)


def test_writer_lays_the_batch_out_and_the_caller_puts_the_columns_in_place(tmp_path, monkeypatch):
    """trk_vcf_dumpstr2.dev_emit (whole-record emit) without a GPU: the writer builds the heads, lays the batch out and
    calls back with every record's column offset; the callee here scribbles over the WHOLE block first (as one copy of a
    device buffer would) and then puts the columns (cut from a first run's own output) at the offsets it was given; one
    record in five is left to the writer (dev_flags).  Same bytes as the first run; a block that is too small is asked
    for again; a failing callback fails the batch (the pipeline falls back to its other writer)."""
    import ctypes as C
    import numpy as np
    from trtools_amd import vcfnative
    (ref, stats, path), = _dump_variants(tmp_path, 'synth_hipstr_all', [{}])
    cols = {}
    for ln in ref[0].split('\n'):
        if ln and not ln.startswith('#'):
            f = ln.split('\t')
            cols[(f[0], f[1])] = ('\t' + '\t'.join(f[9:])).encode()
    state = {'calls': 0, 'fail': False, 'taken': 0, 'cap': None, 'totals': []}

    def fake_regions(self, prm, mask, cf_values, S, out_ring, dev_call=None, cf_plane_idx=None):
        n = self.n
        lo = np.ctypeslib.as_array(self.b.line_off, shape=(n,))
        le = np.ctypeslib.as_array(self.b.line_end, shape=(n,))
        mine, ln, fl = {}, np.zeros(n, np.uint32), np.ones(n, np.uint8)
        for l in range(n):
            head = C.string_at(self.b.text + int(lo[l]), min(int(le[l] - lo[l]), 200)).split(b'\t')
            c = cols.get((head[0].decode(), head[1].decode()))
            if c is not None and l % 5 != 4:
                mine[l] = c
                ln[l], fl[l] = len(c), 0
                state['taken'] += 1

        def emit(_arg, rec_off, total, out):
            state['calls'] += 1
            state['totals'].append(int(total))
            if state['fail']:
                return 1
            C.memset(out, ord('#'), total)
            for l in range(n):
                if l in mine:
                    assert rec_off[l] > 0
                    C.memmove(out + rec_off[l], mine[l], len(mine[l]))
                else:
                    assert rec_off[l] == -1
            return 0
        cb = vcfnative._EMIT_FN(emit)
        cap = state['cap'] if state['cap'] is not None else int(le[n - 1] - lo[0]) * 2 + (1 << 16)
        return dict(buf=None, off=None, len=ln, flags=fl, wait=None, held=None, emit=cb, cap=cap, error=None)

    monkeypatch.setattr(vcfnative.RawBatch, '_device_regions', fake_regions)
    (got, stats2, path2), = _dump_variants(tmp_path, 'synth_hipstr_all', [{}])
    assert path2 == 'batch' and got == ref
    assert state['calls'] >= 1 and state['taken'] > 20
    # a block that is too small: the writer says how much it needs and is called again
    calls0 = state['calls']
    state['cap'] = 1000
    (got2, _, path2b), = _dump_variants(tmp_path, 'synth_hipstr_all', [{}])
    assert got2 == ref and path2b == 'batch' and state['calls'] > calls0
    state['cap'] = None
    state['fail'] = True
    (got3, _, path3), = _dump_variants(tmp_path, 'synth_hipstr_all', [{}])
    assert got3 == ref           # (the batch went to the writer that needs no columns)
