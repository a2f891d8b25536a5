"""World-size-2 test of the locus-sharded path on CPU (gloo): each rank runs the
dumpSTR batch on its contiguous locus shard (oracle-backed compute stand-in), the
counters are all-reduced and the per-locus rows all-gathered; rank 0 compares with
the single-process result over the whole call set."""
import collections
import os
import pickle
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(lo, hi):
    from trtools_amd.synth import make_loci, cells_numpy
    from trtools_amd.batch import HostBatch
    S, Lc = 37, 24
    loci = make_loci(Lc, S, seed=3)
    h = cells_numpy(3, loci, np.arange(lo, hi), S)
    hb = HostBatch(h['gt'], [2] * (hi - lo), loci.allele_lens[lo:hi], loci.allele_strs[lo:hi])
    return hb, [h['dp'], h['q']]


def _run_shard(lo, hi):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle_compute import OracleCompute
    from trtools_amd import _lib as L
    hb, planes = _batch(lo, hi)
    filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_LT, plane_a=1, thr=0.9)]
    spec = dict(min_callrate=0.9, min_hwep=None, min_het=0.1, max_het=None, use_length=False, n_extern=0)
    ch, st, bits, lc = OracleCompute().dumpstr_batch(hb, planes, filters, 0, spec)
    info = collections.OrderedDict()
    info['numcalls'] = ch.sample_counters[0]
    td = ch.totaldp.astype(float)
    td[ch.dp_missing > 0] = np.nan
    info['totaldp'] = td
    info['f0'], info['f1'] = ch.sample_counters[1], ch.sample_counters[2]
    loc = collections.OrderedDict([('totalcalls', int(lc[L.LC_TOTALCALLS])), ('PASS', int(lc[L.LC_PASS])),
                                   ('NO_CALLS_REMAINING', int(lc[L.LC_NO_CALLS])),
                                   ('CALLRATE', int(lc[L.LC_FILTER0 + 0])), ('HETLOW', int(lc[L.LC_FILTER0 + 2]))])
    rows = np.concatenate([st.locus_f64[0].reshape(hi - lo, -1),
                           bits.reshape(-1, 1).astype(np.float64)], axis=1).tobytes()
    return info, loc, rows


def _worker(rank, world, port, outfile):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from trtools_amd import dist as tdist
    from torch_comm import TorchComm
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    lo, hi = tdist.locus_shard(24, rank, world)
    info, loc, rows = _run_shard(lo, hi)
    comm = TorchComm()
    info = tdist.reduce_sample_info(info, comm)
    loc = tdist.reduce_loc_info(loc, comm)
    allrows = tdist.gather_rows(rows, comm)
    if rank == 0:
        with open(outfile, 'wb') as fh:
            pickle.dump((dict(info), dict(loc), allrows), fh)
    dist.barrier()
    dist.destroy_process_group()


def test_locus_shard_partition():
    from trtools_amd.dist import locus_shard
    for n in (0, 1, 7, 24, 100001):
        for w in (1, 2, 3, 8):
            cuts = [locus_shard(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_sharded_dumpstr_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / 'r0.pkl')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    info2, loc2, rows2 = pickle.load(open(out, 'rb'))
    info1, loc1, rows1 = _run_shard(0, 24)
    assert loc2 == dict(loc1)
    assert rows2 == rows1
    for k in info1:
        a, b = np.asarray(info1[k], dtype=float), np.asarray(info2[k], dtype=float)
        assert np.array_equal(np.isnan(a), np.isnan(b)), k
        assert np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)]), k


def _float_worker(rank, world, port, outfile):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from trtools_amd import dist as tdist
    from torch_comm import TorchComm
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    info = collections.OrderedDict()
    info['numcalls'] = np.array([3, 4, 5, 0]) * (rank + 1)
    # ExpansionHunter's LC is a Float field: per-shard depth sums carry fractions (and one sample is poisoned on rank 1)
    info['totaldp'] = np.array([10.25, 0.5, 7.125, 0.0]) + rank * np.array([0.5, 0.25, np.nan if rank else 0.0, 0.0])
    info['f0'] = np.array([1, 0, 2, 0])
    red = tdist.reduce_sample_info(info, TorchComm())
    if rank == 0:
        with open(outfile, 'wb') as fh:
            pickle.dump(dict(red), fh)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_float_depth_sums_keep_their_fraction_across_ranks(tmp_path):
    """reduce_sample_info cast every rank's float totaldp to int64 before the all-reduce, so a
    multi-rank dumpSTR run on ExpansionHunter input printed a different meanDP than the single-process run."""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'f.pkl')
    mp.spawn(_float_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    red = pickle.load(open(out, 'rb'))
    assert red['numcalls'].tolist() == [9, 12, 15, 0] and red['f0'].tolist() == [2, 0, 4, 0]
    td = red['totaldp']
    assert td[0] == 10.25 + 10.75 and td[1] == 0.5 + 0.75 and np.isnan(td[2]) and td[3] == 0.0
