"""The torch-free process group (trtools_amd.dist.SocketGroup): rendezvous, barrier and the small host collectives
bench.py's N > 1 path and the sharded command lines use, on 2 and 3 processes; and the same sharded dumpSTR batch as
tests/test_dist_gloo.py reduced through it (equal to the single-process result)."""
import multiprocessing as mp
import os
import pickle
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from test_dist_gloo import _free_port, _run_shard  # noqa: E402


def _worker(rank, world, port, outfile):
    from trtools_amd import dist as tdist
    g = tdist.SocketGroup(rank, world, '127.0.0.1', port)
    res = {}
    res['bcast'] = g.broadcast_bytes(b'x' * 128 if rank == 0 else b'')
    g.barrier()
    res['sum'] = g.allreduce_sum_i64(np.arange(6).reshape(2, 3) * (rank + 1))
    res['max'] = g.allreduce_max_f64(np.array([0.5 * (rank + 1), -1.0 * rank]))
    parts = g.allgather_bytes(np.full(rank + 2, rank, dtype=np.uint8))
    res['gather'] = [p.tolist() for p in parts]
    # empty payloads and a large one (crosses several recv() calls)
    big = np.arange(300000, dtype=np.int64) + rank
    res['big'] = int(g.allreduce_sum_i64(big).sum())
    res['empty'] = [p.size for p in g.allgather_bytes(np.zeros(0, dtype=np.uint8))]
    lo, hi = tdist.locus_shard(24, rank, world)
    info, loc, rows = _run_shard(lo, hi)
    res['info'] = dict(tdist.reduce_sample_info(info, g))
    res['loc'] = dict(tdist.reduce_loc_info(loc, g))
    res['rows'] = tdist.gather_rows(rows, g)
    g.barrier()
    g.close()
    with open(outfile % rank, 'wb') as fh:
        pickle.dump(res, fh)


@pytest.mark.timeout(300)
@pytest.mark.parametrize('world', [2, 3])
def test_socket_group_collectives_and_sharded_dumpstr(tmp_path, world):
    port = _free_port()
    out = str(tmp_path / 'r%d.pkl')
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    info1, loc1, rows1 = _run_shard(0, 24)
    for r in range(world):
        res = pickle.load(open(out % r, 'rb'))
        assert res['bcast'] == b'x' * 128
        tot = sum(range(1, world + 1))
        assert np.array_equal(res['sum'], np.arange(6).reshape(2, 3) * tot)
        assert res['max'].tolist() == [0.5 * world, 0.0]
        assert res['gather'] == [[q] * (q + 2) for q in range(world)]
        assert res['big'] == int(sum((np.arange(300000, dtype=np.int64) + q).sum() for q in range(world)))
        assert res['empty'] == [0] * world
        assert res['loc'] == dict(loc1)
        assert res['rows'] == rows1
        for k in info1:
            a, b = np.asarray(info1[k], dtype=float), np.asarray(res['info'][k], dtype=float)
            assert np.array_equal(np.isnan(a), np.isnan(b)), k
            assert np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]), k


def test_single_rank_group_is_a_no_op():
    from trtools_amd import dist as tdist
    g = tdist.SocketGroup(0, 1)
    g.barrier()
    assert g.broadcast_bytes(b'ab') == b'ab'
    assert g.allreduce_sum_i64(np.array([1, 2])).tolist() == [1, 2]
    assert [p.tolist() for p in g.allgather_bytes(np.array([7], dtype=np.uint8))] == [[7]]
    g.close()
