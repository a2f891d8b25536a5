"""Locus-sharded statSTR / dumpSTR (one process per GPU in production): two CPU processes over
gloo, each with the oracle-backed compute stand-in, must write exactly what one process writes
(and therefore what the reference writes: the single-process outputs are pinned to its goldens)."""
import argparse
import os
import pickle
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    from trtools_amd import dist as tdist, runtime
    from trtools_amd.statSTR import statSTR
    from trtools_amd.dumpSTR import dumpSTR
    from oracle_compute import OracleCompute
    from torch_comm import TorchComm
    import gen_golden_dumpstr as gg
    runtime.set_compute(OracleCompute())
    tdist.set_comm(TorchComm())
    statSTR.BATCH_CELLS = 50 * 64       # many small batches so that both ranks get work
    dumpSTR.BATCH_CELLS = 24 * 7
    sys.argv = ['dumpSTR', '--synthetic-golden', 'hipstr_all']
    vcf = os.path.join(GOLD, 'data', 'many_samples.vcf.gz')
    rc1 = statSTR.main(gg.stat_args(os.path.join(outdir, 'stat'), vcf, region='1:1000000-2000000', afreq=True, mean=True))
    caller, kw = gg.CASES['hipstr_all']
    rc2 = dumpSTR.main(gg.make_args(os.path.join(outdir, 'dump'), os.path.join(GOLD, 'dumpstr_synth', 'synth_hipstr.vcf'),
                                    caller, **kw))
    # associaTR: many small batches as well
    import contextlib
    import io
    import assoc_cases
    from trtools_amd.associaTR import associaTR as at
    at.BATCH_CELLS = 30 * 50
    with contextlib.redirect_stdout(io.StringIO()):
        at.main(assoc_cases.make_args(os.path.join(outdir, 'assoc.tsv'), **assoc_cases.CASES['two_trait_files'][0]))
    with open(os.path.join(outdir, 'rc%d' % rank), 'w') as fh:
        fh.write('%d %d' % (rc1, rc2))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_cli_outputs_equal_single_process(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path)
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(os.path.join(out, 'rc0')).read() == '0 0' and open(os.path.join(out, 'rc1')).read() == '0 0'
    # statSTR: the reference-generated table of the same arguments
    assert open(os.path.join(out, 'stat.tab')).read() == open(os.path.join(GOLD, 'statstr_flags', 'region.tab')).read()
    # dumpSTR: logs byte for byte, VCF line for line (INFO floats to 1e-6) against the reference-generated goldens
    for ext in ('.samplog.tab', '.loclog.tab'):
        assert open(os.path.join(out, 'dump' + ext)).read() == \
            open(os.path.join(GOLD, 'dumpstr_synth', 'hipstr_all' + ext)).read(), ext
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from assoc_compare import compare_tables
    assert compare_tables(os.path.join(out, 'assoc.tsv'), os.path.join(GOLD, 'associatr', 'two_trait_files.tsv'),
                          rtol=1e-9, p_rtol=0.0) > 900
    assert not [f for f in os.listdir(out) if f.endswith('.temp')]
    from vcf_compare import compare_vcfs
    assert compare_vcfs(os.path.join(out, 'dump.vcf'), os.path.join(GOLD, 'dumpstr_synth', 'hipstr_all.vcf')) == []


def _worker_sockets(rank, world, port, outdir):
    """Three ranks over the torch-free socket group; statSTR on a bgzipped file WITHOUT a region query: contiguous
    shards (trk_vcf_shard).  Every reader's block counters are recorded when it is closed."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from trtools_amd import dist as tdist, runtime, vcfnative
    from trtools_amd.statSTR import statSTR
    from trtools_amd.dumpSTR import dumpSTR
    from oracle_compute import OracleCompute
    import gen_golden_dumpstr as gg
    runtime.set_compute(OracleCompute())
    group = tdist.SocketGroup(rank, world, '127.0.0.1', port)
    tdist.set_comm(group)
    seen = []
    orig_close = vcfnative.NativeVCFReader.close

    def close(self):
        if getattr(self, 'shard_range', None) is not None and getattr(self, '_h', None):
            seen.append((self.shard_range, self.counters()))
        return orig_close(self)
    vcfnative.NativeVCFReader.close = close
    statSTR.BATCH_CELLS = 50 * 64
    dumpSTR.BATCH_CELLS = 3 * 40
    sys.argv = ['dumpSTR', '--synthetic-golden', 'hipstr_all']
    vcf = os.path.join(GOLD, 'data', 'many_samples.vcf.gz')
    rc1 = statSTR.main(gg.stat_args(os.path.join(outdir, 'stat'), vcf, afreq=True, mean=True, numcalled=True))
    trio = os.path.join(GOLD, 'data', 'dumpSTR', 'trio_chr21_hipstr.sorted.vcf.gz')
    rc2 = dumpSTR.main(gg.make_args(os.path.join(outdir, 'dump'), trio, 'hipstr', hipstr_min_call_DP=20,
                                    hipstr_max_call_DP=60, min_locus_callrate=0.7, drop_filtered=True))
    with open(os.path.join(outdir, 'res%d.pkl' % rank), 'wb') as fh:
        pickle.dump((rc1, rc2, seen), fh)
    group.barrier()
    group.close()


@pytest.mark.timeout(900)
def test_three_ranks_read_contiguous_shards(tmp_path):
    """each rank inflates its own share of the compressed file (not all of it) and the merged
    outputs are byte-identical to the single-process run."""
    import multiprocessing as mp
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    world = 3
    out = str(tmp_path / 'w')
    os.makedirs(out)
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_worker_sockets, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(800)
        assert p.exitcode == 0
    # the single-process run of the same command lines, here
    from trtools_amd import runtime
    from trtools_amd.statSTR import statSTR
    from trtools_amd.dumpSTR import dumpSTR
    from oracle_compute import OracleCompute
    import gen_golden_dumpstr as gg
    one = str(tmp_path / 'one')
    os.makedirs(one)
    old = runtime.set_compute(OracleCompute())
    argv = sys.argv
    try:
        sys.argv = ['dumpSTR', '--synthetic-golden', 'hipstr_all']
        vcf = os.path.join(GOLD, 'data', 'many_samples.vcf.gz')
        assert statSTR.main(gg.stat_args(os.path.join(one, 'stat'), vcf, afreq=True, mean=True, numcalled=True)) == 0
        trio = os.path.join(GOLD, 'data', 'dumpSTR', 'trio_chr21_hipstr.sorted.vcf.gz')
        assert dumpSTR.main(gg.make_args(os.path.join(one, 'dump'), trio, 'hipstr', hipstr_min_call_DP=20,
                                         hipstr_max_call_DP=60, min_locus_callrate=0.7, drop_filtered=True)) == 0
    finally:
        sys.argv = argv
        runtime.set_compute(old)
    for name in ('stat.tab', 'dump.vcf', 'dump.samplog.tab', 'dump.loclog.tab'):
        assert open(os.path.join(out, name)).read() == open(os.path.join(one, name)).read(), name
    n_readers = 0
    for r in range(world):
        rc1, rc2, seen = pickle.load(open(os.path.join(out, 'res%d.pkl' % r), 'rb'))
        assert (rc1, rc2) == (0, 0)
        for (b, e), c in seen:
            n_readers += 1
            # what the rank inflated: the compressed bytes of its own range plus at most two more blocks
            assert c['compressed'] <= (e - b) + 2 * 65536 + 64, (r, b, e, c)
    assert n_readers >= 2 * world
