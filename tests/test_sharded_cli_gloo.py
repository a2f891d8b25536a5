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
    import gen_golden_dumpstr as gg
    runtime.set_compute(OracleCompute())
    tdist.set_comm(tdist.TorchComm())
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
