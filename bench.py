#!/usr/bin/env python3
"""bench.py -- statSTR + dumpSTR hot path on synthetic many-sample call sets.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N ...`
  (one rank per GPU).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric / configs[3]): statSTR full statistics + dumpSTR
call- and locus-level filters on a HipSTR-shape call set of 100 000 loci x
10 000 samples per GPU, inputs resident in HBM before the timed region
(generated on the device by k_synth; its numpy twin regenerates rows on the
host for the parity spot-check and the CPU baseline).

One step =
  statSTR : trk_locus_stats(GT)                  (k_locus_count + k_locus_finalize + k_hwe_test)
  dumpSTR : trk_call_filters(GT, DP, Q)          (k_call_filter: min-DP, max-DP, min-Q -> masked GT',
                                                  filter mask, sample counters, and -- via the delta
                                                  outputs -- the allele/genotype counts of GT' obtained by
                                                  subtracting the masked calls from the counts of GT that
                                                  the statSTR half of the step already produced: no second
                                                  pass over the genotype tensor)
            trk_locus_finalize(counts of GT')    (k_locus_finalize + k_hwe_test)
            trk_locus_filters(callrate, HWE, het low/high)  (k_locus_filter)
  N > 1   : loci are sharded by rank (weak scaling: every rank owns 100k loci of an
            N x 100k-locus cohort).  The one real exchange of the path: dumpSTR's per-sample /
            per-filter counters and loc_info are sums over ALL loci -> RCCL all-reduce (< 1 MB);
            the per-locus filter decisions are all-gathered (RCCL, 0.4 MB per rank) for the rank
            that writes the cohort's FILTER column.  Statistic rows stay with the rank that owns
            the loci (each rank writes its slice of the table; rank order == locus order).
Queues: the two HBM-bound stream kernels (k_locus_count, k_call_filter) run on the context's queue 0; the
latency-bound rest (both finalisers, the locus filters, the RCCL exchange) on queue 1 beside the call-filter
kernel -- statSTR's finaliser of the step and dumpSTR's tail of the PREVIOUS step, whose outputs are double
buffered (Workload.step / flush).  All work of the K steps ends inside the timed region (flush + trk_sync).
torch is imported only for N > 1 (rendezvous, barrier, max-over-ranks), never for compute.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_CELL_CALL_FILTER = 20  # read GT 4 + DP 4 + Q 4, write GT' 4 + mask 4
BYTES_PER_CELL_COUNT = 4         # read GT 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--loci', type=int, default=100000)
    ap.add_argument('--samples', type=int, default=10000)
    ap.add_argument('--seed', type=int, default=20260928 + 3)
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the CPU baseline sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-check', action='store_true')
    ap.add_argument('--no-assoc', action='store_true', help='skip the associaTR scan timing (the "associatr_scan" extra)')
    return ap.parse_args()


class Workload:
    """Device-resident buffers + one step of the hot path."""

    def __init__(self, eng, args, rank, world):
        from trtools_amd.synth import SynthBatch
        from trtools_amd import _lib as L
        self.L = L
        self.eng = eng
        self.rank, self.world = rank, world
        self.n_loci, self.n_samples = args.loci, args.samples
        self.sb = SynthBatch(eng, args.loci, args.samples, seed=args.seed, planes=('dp', 'q'),
                             locus_base=rank * args.loci)
        self.planes = [self.sb.dev['dp'], self.sb.dev['q']]
        # dumpSTR --hipstr-min-call-DP 10 --hipstr-max-call-DP 1000 --hipstr-min-call-Q 0.9
        self.filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=1000),
                        dict(op=L.F_LT, plane_a=1, thr=0.9)]
        self.locus_args = dict(min_callrate=0.8, min_hwep=1e-4, min_het=0.05, max_het=0.95, use_length=False)
        b = self.sb.batch
        self.stats_a = [eng.alloc_stats(b) for _ in range(2)]     # statSTR rows (double buffered for the gather)
        self.stats_b = [eng.alloc_stats(b) for _ in range(2)]     # dumpSTR rows
        # everything a step hands to the next stage is double buffered (the tail of step n runs beside the head of
        # step n + 1); the masked genotypes and the mask are written and consumed on queue 0 only: one copy
        from trtools_amd.engine import CallResult
        co = eng.alloc_call_out(b, len(self.filters))
        S = self.n_samples
        self.call_outs = [co, CallResult(co.gt_out, co.filter_mask, eng.zeros((1 + len(self.filters), S), np.int64),
                                         eng.zeros((S,), np.int64), eng.zeros((S,), np.int64),
                                         eng.zeros((4,), np.int32), eng.zeros((S,), np.float64))]
        self.bits_ = [eng.empty((self.n_loci,), np.uint32) for _ in range(2)]
        self.loc_counters_ = [eng.zeros((L.TRK_LC_COLS,), np.int64) for _ in range(2)]
        self.gather = None
        if world > 1 or os.environ.get('TRK_FORCE_DIST'):
            self.gather = eng.empty((world, self.n_loci), np.uint32)
        self.step_no = 0
        self._pending = None
        self.overlap = os.environ.get('TRK_BENCH_OVERLAP', '1') != '0'

    # the buffers of the last completed step
    call_out = property(lambda self: self.call_outs[(self.step_no - 1) & 1])
    bits = property(lambda self: self.bits_[(self.step_no - 1) & 1])
    loc_counters = property(lambda self: self.loc_counters_[(self.step_no - 1) & 1])

    def step(self):
        """One statSTR + dumpSTR pass over the batch.  Queue 0 carries the HBM-bound stream kernels (count, call
        filters), queue 1 the latency-bound rest, placed beside the long call-filter kernel: statSTR's finaliser of
        this step and dumpSTR's finaliser + locus filters (+ the RCCL exchange) of the PREVIOUS step (its outputs
        are double buffered; ``flush`` runs the last one).  TRK_BENCH_OVERLAP=0 puts everything on queue 0, in
        step order."""
        eng = self.eng
        i = self.step_no & 1
        self.step_no += 1
        b = self.sb.batch
        out = self.call_outs[i]
        q1 = 1 if self.overlap else 0
        # counters are per step (each step is a complete statSTR + dumpSTR run)
        out.sample_counters.zero()
        out.sample_totaldp.zero()
        out.sample_dp_missing.zero()
        eng.locus_stats(b, out=self.stats_a[i], count_only=True)                    # statSTR: count
        self.stats_b[i].allele_count.copy_from(self.stats_a[i].allele_count)
        self.stats_b[i].locus_int.copy_from(self.stats_a[i].locus_int)
        eng.queue_wait(q1, 0)
        with eng.on_queue(q1):
            self._tail()                                                           # dumpSTR tail of the previous step
            eng.locus_finalize(b, self.stats_a[i])                                 # statSTR: 11 statistics per locus
        eng.call_filters(b, self.planes, self.filters, dp_plane=0, out=out, delta_stats=self.stats_b[i])
        # queue 0 goes on to the next step once queue 1 is through with what it holds now (the other buffer set)
        eng.queue_wait(0, q1)
        self._pending = i
        if not self.overlap:
            self._tail()

    def _tail(self):
        """dumpSTR after the call filters: statistics of the masked genotypes, locus filters, cohort-wide sums."""
        if self._pending is None:
            return
        eng, i = self.eng, self._pending
        self._pending = None
        out, bits, loc = self.call_outs[i], self.bits_[i], self.loc_counters_[i]
        loc.zero()
        eng.locus_finalize(self.sb.batch, self.stats_b[i])
        eng.locus_filters(self.n_loci, self.stats_b[i], bits_out=bits, counters=loc, **self.locus_args)
        if self.gather is not None:
            eng.allreduce_sum_i64(out.sample_counters)
            eng.allreduce_sum_i64(out.sample_totaldp)
            eng.allreduce_sum_i64(out.sample_dp_missing)
            eng.allreduce_sum_i64(loc)
            eng.allgather(bits, self.gather)

    def flush(self):
        """Enqueue the tail of the last step (call before the final synchronisation)."""
        q1 = 1 if self.overlap else 0
        self.eng.queue_wait(q1, 0)
        with self.eng.on_queue(q1):
            self._tail()


def parity_spot_check(wl, n_check=6):
    """Full-size run vs the oracle on a few regenerated rows + size-independent invariants."""
    from oracle import trtools_oracle as orc
    L = wl.L
    eng = wl.eng
    rng = np.random.default_rng(1)
    idx = np.sort(rng.choice(wl.n_loci, size=min(n_check, wl.n_loci), replace=False))
    host = wl.sb.host_rows(idx)
    i = (wl.step_no - 1) & 1
    cnt = wl.stats_a[i].allele_count.get()[0]
    li = wl.stats_a[i].locus_int.get()[0]
    lf = wl.stats_a[i].locus_f64.get()[0]
    off = wl.sb.tables[0]
    for r, l in enumerate(idx):
        o = orc.locus_stats(host['gt'][r], wl.sb.loci.allele_lens[l], wl.sb.loci.allele_strs[l], None,
                            use_length=False)
        assert np.array_equal(cnt[off[l]:off[l + 1]], o['index_counts']), ("allele counts", l)
        assert li[l, L.LI_N_CALLED] == o['numcalled'], ("numcalled", l)
        for col, key in ((L.LF_HET_STR, 'het'), (L.LF_MEAN, 'mean'), (L.LF_VAR, 'var'), (L.LF_HWEP_STR, 'hwep')):
            a, bb = lf[l, col], o[key]
            assert (np.isnan(a) and np.isnan(bb)) or abs(a - bb) <= 1e-9 * max(1.0, abs(bb)), (key, l, a, bb)
    # dumpSTR half: the delta-corrected counts must equal a recount of the masked genotypes
    lib_ = wl.stats_b[i].locus_int.get()[0]
    cntb = wl.stats_b[i].allele_count.get()[0]
    for r, l in enumerate(idx):
        l = int(l)
        g2 = wl.call_out.gt_out.get_rows(l, l + 1)[0]
        o2 = orc.locus_stats(g2, wl.sb.loci.allele_lens[l], wl.sb.loci.allele_strs[l], None, use_length=False)
        assert np.array_equal(cntb[off[l]:off[l + 1]], o2['index_counts']), ("masked allele counts", l)
        assert lib_[l, L.LI_N_CALLED] == o2['numcalled'], ("masked numcalled", l)
    # invariants over the whole shard
    nall = li[:, L.LI_N_ALLELES].astype(np.int64)
    seg = np.add.reduceat(cnt.astype(np.int64), off[:-1]) if wl.n_loci else np.zeros(0)
    assert np.array_equal(seg, nall), "sum of allele counts != N_ALLELES"
    assert np.all(li[:, L.LI_N_CALLED] <= wl.n_samples) and np.all(li[:, L.LI_N_BAD] == 0)
    cnts = wl.call_out.sample_counters.get()
    if wl.world == 1:
        lib = wl.stats_b[i].locus_int.get()[0]
        # every PASS call is a called sample of the masked matrix and vice versa
        assert int(cnts[0].sum()) == int(lib[:, L.LI_N_CALLED].sum()), "numcalls != called after masking"
        lc = wl.loc_counters.get()
        bits = wl.bits.get()
        assert lc[L.LC_PASS] == int(np.sum(bits == 0))
        assert lc[L.LC_TOTALCALLS] == int(lib[bits == 0, L.LI_N_CALLED].sum())
    return len(idx)


def assoc_extra(wl, args, iters=5):
    """SURVEY section 8 row f3 / BASELINE configs[4] on ONE GPU, outside the timed region of the headline
    metric: the associaTR scan (trk_assoc_scan) over this rank's resident genotype tensor, one seeded
    standard-normal trait, every sample in the regression set.  4 algorithmic bytes per call (the GT read).
    A handful of loci is checked against the associaTR oracle."""
    from trtools_amd.synth import pack_assoc_tables
    eng = wl.eng
    n_loci, n_samples = wl.n_loci, wl.n_samples
    alen, rcls = pack_assoc_tables(wl.sb.loci.allele_lens, 2)
    alen_d, rcls_d = eng.upload(alen, np.float64), eng.upload(rcls, np.uint16)
    rng = np.random.default_rng(args.seed + 77)
    y = rng.normal(size=n_samples)
    y = (y - y.mean()) / y.std()
    vec_d = eng.upload(y[None, :].copy(), np.float64)
    res = None
    eng.profile(True)
    for it in range(iters + 1):
        if it == 1:
            eng.sync()
            eng.profile_reset()
            t0 = time.perf_counter()
        res = eng.assoc_scan(wl.sb.batch, vec_d, alen_d, rcls_d, non_major_cutoff=20.0, out=res)
    eng.sync()
    wall = (time.perf_counter() - t0) / iters
    prof = eng.profile_get()
    eng.profile(False)
    n, ms = prof['k_assoc_scan']
    nf, msf = prof['k_assoc_finalize']
    scan_ms = ms / max(n, 1)
    cells = n_loci * n_samples
    out = {"workload": "associaTR linear-regression scan, %d loci x %d samples x 1 trait (BASELINE configs[4] on one GPU)"
                       % (n_loci, n_samples),
           "loci_per_s": n_loci / wall, "ms_per_pass": wall * 1e3,
           "kernels_ms": {"k_assoc_scan": scan_ms, "k_assoc_finalize": msf / max(nf, 1)},
           "roofline": {"bound": "hbm", "kernel": "k_assoc_scan", "bytes_per_cell": 4,
                        "achieved": cells * 4 / (scan_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": cells * 4 / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    if not args.no_check:
        from oracle import associatr_oracle as ao
        li, lf = res.locus_int.get(), res.locus_f64.get()
        idx = np.unique(np.linspace(0, n_loci - 1, 5).astype(int))
        rows = wl.sb.host_rows(idx)
        sf = np.ones(n_samples, dtype=bool)
        covars = np.ones((n_samples, 2))
        checked = 0
        for k, l in enumerate(idx):
            r = ao.scan_locus(rows['gt'][k], wl.sb.loci.allele_lens[l], sf, covars, y, 1.0, 20.0, 2)
            assert li[l, 0] == r['n_tested'], (l, li[l], r['n_tested'])
            if r['locus_filtered']:
                assert li[l, 1] != 0, (l, r['locus_filtered'])
                continue
            assert li[l, 1] == 0, (l, li[l])
            for col, key in ((0, 'pval'), (1, 'coef_std'), (2, 'se_std'), (3, 'rsquared')):
                assert abs(lf[l, col] - r[key]) <= 1e-9 * abs(r[key]) + 1e-12, (l, key, lf[l, col], r[key])
            checked += 1
        out["parity_loci_checked"] = int(len(idx))
        out["parity_loci_regressed"] = checked
        if not args.no_cpu_baseline:
            # the same scan through the oracle port (numpy + scipy, one locus and one OLS fit at a time like the
            # reference), 1 core, on a bounded sample of loci regenerated by the generator's numpy twin
            t0 = time.perf_counter()
            done = 0
            rng2 = np.random.default_rng(args.seed + 78)
            while time.perf_counter() - t0 < 3.0:
                pick = np.sort(rng2.choice(n_loci, size=8, replace=False))
                rows2 = wl.sb.host_rows(pick)
                for k, l in enumerate(pick):
                    ao.scan_locus(rows2['gt'][k], wl.sb.loci.allele_lens[l], sf, covars, y, 1.0, 20.0, 2)
                done += len(pick)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": done / dt, "unit": "loci/s", "cores": 1, "kind": "port",
                                   "sample": "%d random loci x %d samples through oracle/associatr_oracle.py "
                                             "(incl. regenerating the rows), %.1f s" % (done, n_samples, dt)}
    for d in (alen_d, rcls_d, vec_d, res.locus_int, res.locus_f64, res.allele_count):
        d.free()
    return out


def cpu_baseline(wl, budget_s):
    """The oracle (numpy/scipy port of the reference's per-locus algorithm) on a bounded
    sample of the same workload, one host core."""
    import collections
    from oracle import trtools_oracle as orc
    rng = np.random.default_rng(2)
    S = wl.n_samples
    done = 0
    t0 = time.perf_counter()
    info = collections.OrderedDict([('numcalls', np.zeros(S, dtype=int)), ('totaldp', np.zeros(S)),
                                    ('mindp', np.zeros(S, dtype=int)), ('maxdp', np.zeros(S, dtype=int)),
                                    ('minq', np.zeros(S, dtype=int))])
    loc = collections.defaultdict(int)
    gen_time = 0.0
    while True:
        l = int(rng.integers(0, wl.n_loci))
        tg = time.perf_counter()
        h = wl.sb.host_rows(np.array([l]))
        gen_time += time.perf_counter() - tg
        gt, dp, q = h['gt'][0], h['dp'][0].reshape(-1, 1), h['q'][0].reshape(-1, 1)
        lens, strs = wl.sb.loci.allele_lens[l], wl.sb.loci.allele_strs[l]
        orc.locus_stats(gt, lens, strs, None, use_length=False)            # statSTR, all 11 stats
        outs = [('mindp', orc.filt_min_value(dp, 10)), ('maxdp', orc.filt_max_value(dp, 1000)),
                ('minq', orc.filt_min_value(q, 0.9))]
        g2, _ = orc.apply_call_filters(gt, outs, info, dp=dp)              # dumpSTR call filters
        try:
            orc.apply_locus_filters(g2, lens, strs, loc, use_length=False, min_callrate=0.8, min_hwep=1e-4,
                                    min_het=0.05, max_het=0.95)
            orc.locus_info_fields(g2, lens, strs, False)
        except ValueError:
            pass
        done += 1
        if time.perf_counter() - t0 - gen_time >= budget_s:
            break
    el = time.perf_counter() - t0 - gen_time
    return dict(value=done / el, unit="loci/s", cores=1, kind="port",
                sample="%d random loci x %d samples of the same synthetic call set, statSTR (11 stats, "
                       "string alleles) + dumpSTR (3 call filters, 4 locus filters, INFO recompute) through "
                       "oracle/trtools_oracle.py (numpy+scipy, per-locus like the reference), %.1f s"
                       % (done, S, el),
                cells_per_s=done * S / el)


def cpu_baseline_c(wl, n_loci=3072, max_threads=32):
    """The C half of the oracle (oracle/oracle_c.c: counts, statistics, exact HWE test, the three threshold call
    filters, recount of the masked genotypes) on a bounded sample of the same call set: one core, then all cores
    with the loci split over threads ('single core and all cores').  A compiled, per-locus CPU
    implementation of the same step -- a stronger baseline than the numpy port, still only a reported number."""
    import threading
    from oracle import oracle_c
    n_loci = min(n_loci, wl.n_loci)
    idx = np.arange(n_loci)
    h = wl.sb.host_rows(idx)
    off_all = wl.sb.tables[0]
    off = (off_all[:n_loci + 1] - off_all[0]).astype(np.int32)
    lc, sc, cv = (np.ascontiguousarray(t[off_all[0]:off_all[n_loci]]) for t in wl.sb.tables[1:4])

    def work(lo, hi):
        o = (off[lo:hi + 1] - off[lo]).astype(np.int32)
        sl = slice(int(off[lo]), int(off[hi]))
        oracle_c.batch_stats(h['gt'][lo:hi], None, o, lc[sl], sc[sl], cv[sl])                      # statSTR
        g2 = oracle_c.call_filters_dpq(h['gt'][lo:hi], h['dp'][lo:hi], h['q'][lo:hi], 10, 1000, 0.9)[0]
        oracle_c.batch_stats(g2, None, o, lc[sl], sc[sl], cv[sl])                                  # dumpSTR on GT'

    oracle_c.load()
    out = {}
    for label, nt in (('one_core', 1), ('all_cores', max(1, min(max_threads, os.cpu_count() or 1, n_loci // 8)))):
        bounds = np.linspace(0, n_loci, nt + 1).astype(int)
        th = [threading.Thread(target=work, args=(int(bounds[i]), int(bounds[i + 1]))) for i in range(nt)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        out[label] = dict(value=n_loci / el, unit="loci/s", cores=nt, cells_per_s=n_loci * wl.n_samples / el)
    out['kind'] = "port (C restatement, oracle/oracle_c.c)"
    out['sample'] = "%d loci x %d samples of the same synthetic call set" % (n_loci, wl.n_samples)
    return out


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist = None
    use_dist = world > 1 or bool(os.environ.get('TRK_FORCE_DIST'))   # TRK_FORCE_DIST: exercise the
    if use_dist:                                                      # collective path on one rank
        import torch.distributed as dist  # rendezvous / barrier only
        dist.init_process_group(backend='gloo')
    from trtools_amd.engine import Engine
    eng = Engine(local_rank)
    if use_dist:
        uid = [eng.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        eng.comm_init(rank, world, uid[0])
    wl = Workload(eng, args, rank, world)

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        wl.step()
    wl.flush()
    barrier()
    eng.profile(True)
    eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    wl.flush()
    eng.sync()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        dist.barrier()
    prof = eng.profile_get()
    eng.profile(False)

    n_checked = 0
    if not args.no_check:
        n_checked = parity_spot_check(wl)
    if rank == 0:
        cells = wl.n_loci * wl.n_samples
        ms_step = elapsed / args.steps * 1e3
        loci_s = world * wl.n_loci * args.steps / elapsed
        kn, kms = prof['k_call_filter']
        cn, cms = prof['k_locus_count']
        fn_, fms = prof['k_locus_finalize']
        avg_cf = kms / max(kn, 1)
        avg_cnt = cms / max(cn, 1)
        achieved = cells * BYTES_PER_CELL_CALL_FILTER / (avg_cf * 1e-3) / 1e9 if kn else 0.0
        traffic = None
        tf = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get('k_call_filter_bytes_per_launch')
            except Exception:
                traffic = None
        out = {
            "metric": "loci/sec (and genotype-cells/sec) statSTR+dumpSTR, 100k loci x 10k samples",
            "value": loci_s, "unit": "loci/s",
            "cells_per_sec": loci_s * wl.n_samples,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i16", "data": "synthetic",
            "config": {"workload": "statSTR (11 stats) + dumpSTR (min-DP/max-DP/min-Q call filters, "
                                   "callrate/HWE/het-low/het-high locus filters) combined, HipSTR-shape, "
                                   "%d loci x %d samples per GPU (BASELINE configs[3])" % (wl.n_loci, wl.n_samples),
                       "n_loci_per_gpu": wl.n_loci, "n_samples": wl.n_samples, "ploidy": 2,
                       "max_alleles": int(np.max(np.diff(wl.sb.tables[0]))),
                       "sharding": "loci by rank; RCCL all-reduce of sample/locus counters + all-gather of the "
                                   "per-locus filter decisions" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "k_call_filter", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_cell": BYTES_PER_CELL_CALL_FILTER, "avg_launch_ms": avg_cf,
                         "launches": kn},
            "kernels_ms": {k: (v[1] / v[0] if v[0] else None) for k, v in prof.items()},
            "k_locus_count_roofline": {"achieved": cells * BYTES_PER_CELL_COUNT / (avg_cnt * 1e-3) / 1e9 if cn else 0.0,
                                       "unit": "GB/s", "bytes_per_cell": BYTES_PER_CELL_COUNT,
                                       "frac": (cells * BYTES_PER_CELL_COUNT / (avg_cnt * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                       if cn else 0.0},
            "queues": ("2: stream kernels on queue 0, finalisers / locus filters / RCCL exchange on queue 1 and "
                       "overlapped with them -- kernels_ms are per-launch averages under that contention "
                       "(queue 1 work runs beside k_call_filter)") if wl.overlap else "1",
            "parity_rows_checked": n_checked,
            "device": eng.arch,
        }
        if not args.no_assoc and world == 1:
            out["extras"] = {"associatr_scan": assoc_extra(wl, args)}
        if not args.no_cpu_baseline and world == 1:   # the CPU baselines are timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds)
            try:
                out.setdefault("extras", {})["cpu_baseline_c"] = cpu_baseline_c(wl)
            except Exception as e:      # the checker's C half is optional equipment of the box
                out.setdefault("extras", {})["cpu_baseline_c"] = {"error": str(e)[:200]}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == '__main__':
    main()
