"""Randomised end-to-end runs: a random synthetic HipSTR / GangSTR VCF (any sample count) through the statSTR and
dumpSTR command-line mirrors, once with the device behind the compute seam and once with the oracle-backed seam the CPU
tests use -- the statistics table, the output VCF and both dumpSTR logs must be byte-identical."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from test_dumpstr_cli import make_args as dump_args
from test_statstr_cli import _args as stat_args

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def computes():
    from oracle_compute import OracleCompute
    from trtools_amd.compute import DeviceCompute
    return DeviceCompute(), OracleCompute()


def _both(computes, run):
    from trtools_amd import runtime
    outs = []
    for c in computes:
        old = runtime.set_compute(c)
        try:
            outs.append(run())
        finally:
            runtime.set_compute(old)
    return outs


@settings(max_examples=12, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 10**6), n_loci=st.integers(3, 40), S=st.integers(1, 70), caller=st.sampled_from(['hipstr', 'gangstr']),
       use_length=st.booleans(), groups=st.booleans())
def test_clis_device_equals_oracle_seam(tmp_path_factory, computes, seed, n_loci, S, caller, use_length, groups):
    from trtools_amd import synth
    from trtools_amd.dumpSTR import dumpSTR
    from trtools_amd.statSTR import statSTR
    rng = np.random.default_rng(seed)
    d = tmp_path_factory.mktemp('cli')
    vcf = str(d / 'in.vcf')
    loci = synth.make_loci(n_loci, S, seed=seed, pure_repeats=(caller == 'gangstr'))
    idx = np.arange(n_loci)
    rows = synth.cells_numpy(seed, loci, idx, S)
    extra = synth.gangstr_planes_numpy(seed, loci, idx, S, rows['gt'], rows['dp'], 0) if caller == 'gangstr' else None
    names = synth.render_vcf(vcf, loci, rows, caller=caller, extra=extra)
    # ---- statSTR ----
    kw = dict(vcf=vcf, vcftype=caller, use_length=use_length, nalleles_thresh=0.05)
    if groups and S >= 2:
        files = []
        for g in range(2):
            f = str(d / ('grp%d.txt' % g))
            pick = [n for n in names if rng.random() < 0.6] or [names[0]]
            open(f, 'w').write('\n'.join(pick) + '\n')
            files.append(f)
        kw['samples'] = ','.join(files)
        kw['sample_prefixes'] = 'a,b'

    def run_stat():
        out = str(d / 'stat')
        try:
            rc = statSTR.main(stat_args(out, **kw))
        except (ValueError, IndexError) as e:       # the reference's own failures (HWE test without a full genotype)
            return type(e).__name__ + ': ' + str(e), None
        return rc, open(out + '.tab').read() if rc == 0 else None
    (rc_a, tab_a), (rc_b, tab_b) = _both(computes, run_stat)
    assert rc_a == rc_b and tab_a == tab_b
    # ---- dumpSTR ----
    if caller == 'hipstr':
        f = dict(hipstr_min_call_DP=int(rng.integers(5, 25)), hipstr_max_call_DP=int(rng.integers(40, 200)),
                 hipstr_min_call_Q=float(rng.choice([0.8, 0.9, 0.95])), hipstr_max_call_stutter=0.15,
                 hipstr_max_call_flank_indel=0.15)
        if rng.random() < 0.5:
            f['hipstr_min_supp_reads'] = int(rng.integers(1, 12))
    else:
        f = dict(gangstr_min_call_DP=int(rng.integers(5, 25)), gangstr_max_call_DP=int(rng.integers(40, 200)),
                 gangstr_min_call_Q=0.9, gangstr_expansion_prob_het=0.05, gangstr_expansion_prob_total=0.2,
                 gangstr_filter_span_only=bool(rng.integers(0, 2)), gangstr_filter_spanbound_only=bool(rng.integers(0, 2)),
                 gangstr_filter_badCI=bool(rng.integers(0, 2)))
    f.update(vcftype=caller, use_length=use_length, min_locus_callrate=float(rng.choice([0.0, 0.5, 0.8])),
             min_locus_het=0.05, max_locus_het=0.9, drop_filtered=bool(rng.integers(0, 2)))
    if rng.random() < 0.5:
        f['filter_hrun'] = True

    def run_dump():
        out = str(d / 'dump')
        try:
            rc = dumpSTR.main(dump_args(out, vcf, **f))
        except (ValueError, IndexError) as e:       # e.g. the HWE filter on a locus without a full genotype
            return type(e).__name__, None
        return rc, tuple(open(out + ext).read() for ext in ('.vcf', '.samplog.tab', '.loclog.tab')) if rc == 0 else None
    (rc_a, out_a), (rc_b, out_b) = _both(computes, run_dump)
    assert rc_a == rc_b
    assert out_a == out_b


@settings(max_examples=15, deadline=None, derandomize=True, database=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 10**6), n_loci=st.integers(2, 30), S=st.integers(1, 50), use_length=st.booleans(),
       haploid_records=st.booleans())
def test_mixed_ploidy_device_equals_oracle_seam(tmp_path_factory, computes, seed, n_loci, S, use_length,
                                                haploid_records):
    """Haploid records and haploid samples (chrX-like) in one file: batches padded with -2, per-locus ploidy tables,
    row padding -- the statSTR table and the dumpSTR outputs (VCF, both logs) of the device seam equal the
    oracle-backed seam's."""
    from trtools_amd.statSTR import statSTR
    rng = np.random.default_rng(seed)
    d = tmp_path_factory.mktemp('mp')
    vcf = str(d / 'in.vcf')
    lines = ['##fileformat=VCFv4.1', '##command=HipSTR-v0.6.2 fuzz', '##INFO=<ID=START,Number=1,Type=Integer,Description="s">',
             '##INFO=<ID=END,Number=1,Type=Integer,Description="e">', '##INFO=<ID=PERIOD,Number=1,Type=Integer,Description="p">',
             '##FORMAT=<ID=GT,Number=1,Type=String,Description="g">', '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="d">',
             '##FORMAT=<ID=Q,Number=1,Type=Float,Description="q">', '##contig=<ID=chrX>',
             '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('s%d' % i for i in range(S))]
    for l in range(n_loci):
        pos = 1000 + 200 * l
        n_alt = int(rng.integers(0, 5))
        ref = 'AC' * int(rng.integers(3, 9))
        alts = ['AC' * int(rng.integers(1, 14)) + ('A' if rng.random() < 0.2 else '') for _ in range(n_alt)]
        alts = [a for a in dict.fromkeys(alts) if a != ref]
        A = 1 + len(alts)
        haploid_record = haploid_records and rng.random() < 0.35
        cols = []
        for s in range(S):
            fmt = ':%d:%.2f' % (int(rng.integers(1, 60)), rng.random())
            if rng.random() < 0.1:
                cols.append(('.' if haploid_record or rng.random() < 0.5 else './.') + ':.:.')
            elif haploid_record or rng.random() < 0.2:
                cols.append(str(int(rng.integers(0, A))) + fmt)
            else:
                a, b = int(rng.integers(0, A)), int(rng.integers(0, A))
                cols.append('%s|%s' % (a, '.' if rng.random() < 0.05 else b) + fmt)
        lines.append('\t'.join(['chrX', str(pos), '.', ref, ','.join(alts) or '.', '.', '.',
                                'START=%d;END=%d;PERIOD=2' % (pos, pos + len(ref) - 1), 'GT:DP:Q'] + cols))
    open(vcf, 'w').write('\n'.join(lines) + '\n')

    def run_stat(hwep):
        out = str(d / 'stat')
        try:
            rc = statSTR.main(stat_args(out, vcf=vcf, vcftype='hipstr', use_length=use_length, hwep=hwep, nalleles_thresh=0.05))
        except (ValueError, IndexError) as e:
            return type(e).__name__ + ': ' + str(e), None
        return rc, open(out + '.tab').read() if rc == 0 else None
    for hwep in (False, True):      # with --hwep haploid records make the reference raise: same error from both seams
        (rc_a, tab_a), (rc_b, tab_b) = _both(computes, lambda: run_stat(hwep))
        assert rc_a == rc_b and tab_a == tab_b
        if hwep is False:
            assert rc_a == 0

    from trtools_amd.dumpSTR import dumpSTR
    f = dict(vcftype='hipstr', use_length=use_length, hipstr_min_call_DP=int(rng.integers(5, 30)),
             hipstr_max_call_DP=int(rng.integers(35, 60)), hipstr_min_call_Q=float(rng.choice([0.3, 0.6, 0.9])),
             min_locus_callrate=float(rng.choice([0.0, 0.5])), min_locus_het=0.05, max_locus_het=0.95)

    def run_dump():
        out = str(d / 'dump')
        try:
            rc = dumpSTR.main(dump_args(out, vcf, **f))
        except (ValueError, IndexError) as e:
            return type(e).__name__ + ': ' + str(e), None
        return rc, tuple(open(out + ext).read() for ext in ('.vcf', '.samplog.tab', '.loclog.tab')) if rc == 0 else None
    # (dumpSTR recomputes HWEP for INFO: a haploid record makes the reference -- and both seams -- raise IndexError)
    (rc_a, out_a), (rc_b, out_b) = _both(computes, run_dump)
    assert rc_a == rc_b and out_a == out_b
    if not haploid_records and S >= 16:     # (a record all of whose samples came out haploid is a haploid record)
        assert rc_a == 0
