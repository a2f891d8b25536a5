#!/usr/bin/env python3
"""bench.py -- statSTR + dumpSTR hot path on synthetic many-sample call sets.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by `python -m torch.distributed.run --nproc-per-node N ...`
  (one rank per GPU).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json metric / configs[3]): statSTR full statistics + dumpSTR
call- and locus-level filters on a HipSTR-shape call set of 100 000 loci x
10 000 samples, inputs resident in HBM before the timed region (generated on
the device by k_synth; its numpy twin regenerates rows on the host for the
CPU baseline).

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
  N > 1   : --scaling strong (default, BASELINE configs[3] as written): the 100 000-locus cohort is cut into
            contiguous locus shards, one per rank (12 500 loci each at N = 8; `value` = 100 000 loci x steps /
            time).  --scaling weak: every rank owns 100 000 loci of an N x 100k-locus cohort.
            The one real exchange of the path: dumpSTR's per-sample / per-filter counters and loc_info are sums
            over ALL loci (dumpSTR.py:1251-1268) and the per-locus filter decisions are needed by the rank that
            writes the cohort's FILTER column -> ONE grouped RCCL launch per step (trk_exchange: all-reduce of
            one packed int64 buffer + all-gather of the filter bits).  Statistic rows stay with the rank that
            owns the loci (rank order == locus order).
Queues: the two HBM-bound stream kernels (k_locus_count, k_call_filter) run on the context's queue 0; the
latency-bound rest beside the call-filter kernel on queues 1 and 2 -- dumpSTR's tail of the PREVIOUS step
(finaliser, locus filters, the RCCL exchange; its outputs are double -- for N > 1 triple -- buffered) on queue 1, statSTR's finaliser of
the step on queue 2 (Workload.step / flush).  All work of the K steps ends inside the timed region (flush + trk_sync).
torch is imported only for N > 1 (rendezvous, barrier, max-over-ranks), never for compute.

Parity inside the bench: after the timed region EVERY locus of the shard is checked against the compiled C
restatement of the reference's algorithm (oracle/fullsize.py: counts, 11 statistics, filter masks, masked
genotypes, sample_info, locus filter decisions, loc_info) -- `parity_rows_checked` == loci of the shard.

Extras (N = 1 only, each outside the timed region of the headline metric):
  strong_shard    the same step at 100000/N loci for N = 2, 4, 8 on this one GPU through a 1-rank RCCL
                  communicator: what one of N ranks does in the strong-scaled run, with the predicted scaling
                  efficiency (t_100k / N) / t_shard
  config1         BASELINE configs[1]: statSTR full statistics, 10k loci x 1k samples
  config2         BASELINE configs[2]: dumpSTR, GangSTR shape, nine call + four locus filters, 50k x 5k
  compact_outputs the call-filter pass with the opt-in one-byte mask and no masked-genotype plane (13 B per call)
  in_place_gt     the call-filter pass with the masked genotypes written into the batch's own tensor (sparse stores)
  end_to_end      statSTR's command line from a bgzipped text VCF to its table; a packed host batch through
                  upload + kernels + download (PCIe included)
  associatr_scan  BASELINE configs[4] on one GPU
  cpu_baseline_c  the compiled C restatement on one core and on all cores
"""
import argparse
import json
import os
import sys
import time
import warnings

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
    ap.add_argument('--loci', type=int, default=100000, help='loci of the cohort (strong) / per GPU (weak)')
    ap.add_argument('--samples', type=int, default=10000)
    ap.add_argument('--scaling', choices=('strong', 'weak'), default='strong')
    ap.add_argument('--seed', type=int, default=20260928 + 3)
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='budget of the CPU baseline sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-check', action='store_true')
    ap.add_argument('--no-assoc', action='store_true', help='skip the associaTR scan timing (the "associatr_scan" extra)')
    ap.add_argument('--no-extras', action='store_true', help='skip strong_shard / config1 / config2')
    return ap.parse_args()


FILTERS_DPQ = None


def filters_dpq():
    """dumpSTR --hipstr-min-call-DP 10 --hipstr-max-call-DP 1000 --hipstr-min-call-Q 0.9"""
    from trtools_amd import _lib as L
    return [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=1000),
            dict(op=L.F_LT, plane_a=1, thr=0.9)]


LOCUS_ARGS = dict(min_callrate=0.8, min_hwep=1e-4, min_het=0.05, max_het=0.95, use_length=False)


class Workload:
    """Device-resident buffers of one rank's shard + one step of the hot path."""

    def __init__(self, eng, seed, n_samples, loci, locus_base, world, use_comm, overlap=True, gather_loci=None,
                 pipeline_count=False, n_sets=None, ev_base=0, host_group=None):
        from trtools_amd.synth import SynthBatch
        from trtools_amd import _lib as L
        from trtools_amd.engine import CallResult
        self.L = L
        self.eng = eng
        self.world = world
        self.n_loci, self.n_real = len(loci.allele_lens), n_samples
        self.sb = SynthBatch(eng, self.n_loci, n_samples, seed=seed, planes=('dp', 'q'), locus_base=locus_base,
                             loci=loci)
        # Rows padded to a multiple of 32 samples (128 bytes) with no-call samples, as compute.DeviceCompute uploads
        # any cohort (trk_batch.n_pad_samples, trk_pad_rows): n_real = the cohort's samples (what every reported rate
        # counts), n_samples = the row length of every device array.  TRK_ROW_ALIGN=4: the dense layout.
        self.sb.pad_rows(max(4, int(os.environ.get('TRK_ROW_ALIGN', '32')) & ~3))
        self.n_samples = n_samples = self.sb.n_dev
        self.planes = [self.sb.dev['dp'], self.sb.dev['q']]
        self.filters = filters_dpq()
        self.locus_args = dict(LOCUS_ARGS)
        b = self.sb.batch
        # statSTR rows and, as the twin copy of the same count pass (TRK_STATS_TWIN), the counts dumpSTR's call
        # filters correct in place; one per buffer set (the tail runs one step behind)
        # Buffer sets of what a step hands on (small per-locus / per-sample arrays).  Two when the count pass runs in
        # order before the call filters; THREE when it is pipelined beside the previous step's call filters: with
        # two, count(n + 1) must wait for tail(n - 1) -- which runs beside call_filters(n) and, slowed by it, takes
        # as long -- so it started only when call_filters(n) ended (kernel timeline of the 12.5k-locus shard,
        # tools/timeline_probe.py: 125-148 us between two call-filter kernels, now the reduction kernel's ~20)
        self.NB = NB = int(n_sets or os.environ.get('TRK_BENCH_NSETS') or (3 if pipeline_count else 2))
        self.EV_COUNT, self.EV_CF, self.EV_TAIL, self.EV_FINA = (ev_base + k * NB for k in range(4))
        self.stats_a = [eng.alloc_stats(b, twin=True) for _ in range(NB)]
        self.stats_b = [(st.twin if getattr(st, 'twin', None) else eng.alloc_stats(b)) for st in self.stats_a]
        # everything a step hands to the next stage exists once per buffer set (the tail of step n runs beside the head of
        # step n + 1); the masked genotypes and the mask are written and consumed on queue 0 only: one copy.
        # Everything that is summed over the ranks lives back to back in ONE int64 buffer per step slot
        # (sample_info rows, totaldp, dp-missing, loc_info): one memset, one all-reduce.
        S, nf = n_samples, len(self.filters)
        S2 = (S + 1) & ~1                                         # 16-byte aligned segments
        self.sums_, self.call_outs, self.loc_counters_ = [], [], []
        # the two output planes of the call-filter pass, placed as every caller of Engine.alloc_call_out gets them
        # (trk_dev_alloc_pair: on this part the pass's two write streams run on one of two levels depending on where the
        # two planes landed; at most two spare planes during the search, TRK_PLACE_OUTPUTS=0: plain allocations)
        self.placement = None
        # (the product's own search, no private settings: at most two spare planes and two 16 GB steps, i.e. never more
        # than the working set again in transient memory; what it saw and took is in the line: roofline.placement_*)
        co0 = eng.alloc_call_out(b, len(self.filters))
        gt_out, mask = co0.gt_out, co0.filter_mask
        for x in (co0.sample_counters, co0.sample_totaldp, co0.sample_dp_missing, co0.error, co0.sample_totaldp_f64):
            x.free()
        if self.n_loci * S * 4 >= eng.PLACE_MIN_BYTES and os.environ.get('TRK_PLACE_OUTPUTS', '1') != '0':
            self.placement = dict(type(eng).last_placement or {})
        for _ in range(NB):
            sums = eng.zeros(((1 + nf) * S2 + 2 * S2 + L.TRK_LC_COLS,), np.int64)
            sc = sums.view(0, (1 + nf, S), np.int64) if S2 == S else None
            if sc is None:
                raise ValueError("an even number of samples is required")
            td = sums.view((1 + nf) * S2 * 8, (S,), np.int64)
            dm = sums.view((2 + nf) * S2 * 8, (S,), np.int64)
            loc = sums.view((3 + nf) * S2 * 8, (L.TRK_LC_COLS,), np.int64)
            self.sums_.append(sums)
            self.loc_counters_.append(loc)
            self.call_outs.append(CallResult(gt_out, mask, sc, td, dm, eng.zeros((4,), np.int32),
                                             eng.zeros((S,), np.float64)))
        # the all-gather moves equal-sized rows: the largest shard's size (shards differ by at most one locus)
        gl = max(self.n_loci, gather_loci or 0)
        self.bits_ = [eng.zeros((gl,), np.uint32) for _ in range(NB)]
        self.gather = eng.empty((world, gl), np.uint32) if use_comm else None
        self.step_no = 0
        self._pending = None
        self.overlap = overlap
        self.pipeline_count = pipeline_count
        # TRK_BENCH_COMM=host (rehearsal of the N > 1 control flow with several ranks on ONE device, where RCCL
        # refuses to build a communicator): the step's exchange goes through host copies over the socket group
        self.host_group = host_group

    # the buffers of the last completed step
    last = property(lambda self: (self.step_no - 1) % self.NB)     # buffer set of the last completed step
    call_out = property(lambda self: self.call_outs[self.last])
    bits = property(lambda self: self.bits_[self.last])
    loc_counters = property(lambda self: self.loc_counters_[self.last])
    # ordering points (trk_event_*): EV_COUNT / EV_CF / EV_TAIL / EV_FINA + i, one per buffer set i = step mod NB

    def step(self):
        """One statSTR + dumpSTR pass over the shard.
        queue 0  the call-filter pass (k_call_filter + k_cf_reduce), HBM bound
        queue 1  dumpSTR's tail of the PREVIOUS step: finaliser, locus filters, the RCCL exchange
        queue 2  statSTR's finaliser of this step
        queue 3  the count pass of this step when ``pipeline_count`` (N > 1 shards: the count of step n runs beside
                 the end of the call filters of step n - 1 -- it does not depend on them -- so that the ramp and the
                 tail of the two short stream kernels overlap); otherwise queue 0, in order before the call filters
        Every output a step hands on exists NB times; a queue that recycles a buffer set waits for the marks its
        consumers recorded NB steps ago (long past), never for work enqueued in this step, so queue 0 runs its
        stream kernels back to back.  TRK_BENCH_OVERLAP=0 puts everything on queue 0, in step order."""
        eng = self.eng
        i = self.step_no % self.NB
        self.step_no += 1
        b = self.sb.batch
        out = self.call_outs[i]
        q1, q2, qc = (1, 2, 3 if self.pipeline_count else 0) if self.overlap else (0, 0, 0)
        with eng.on_queue(qc):
            eng.event_wait(self.EV_TAIL + i)       # tail(n - NB) has consumed sums / stats_b of this buffer set
            eng.event_wait(self.EV_FINA + i)       # fin_a(n - NB) has consumed stats_a of this buffer set
            self.sums_[i].zero()                   # counters are per step (each step is a complete run)
            eng.locus_stats(b, out=self.stats_a[i], count_only=True)                # statSTR: count (+ twin copy)
            if not getattr(self.stats_a[i], 'twin', None):
                self.stats_b[i].allele_count.copy_from(self.stats_a[i].allele_count)
                self.stats_b[i].locus_int.copy_from(self.stats_a[i].locus_int)
            eng.event_record(self.EV_COUNT + i)
        with eng.on_queue(q1):
            # beside this step's call filters, not beside its count pass (the latency-bound finalisers cost the
            # short count kernel a third of its time, the long call-filter kernel ~3 %)
            eng.event_wait(self.EV_COUNT + i)
            self._tail()                                                           # dumpSTR tail of the previous step
        with eng.on_queue(q2):
            eng.event_wait(self.EV_COUNT + i)
            eng.locus_finalize(b, self.stats_a[i])                                 # statSTR: 11 statistics per locus
            eng.event_record(self.EV_FINA + i)
        eng.event_wait(self.EV_COUNT + i)
        eng.call_filters(b, self.planes, self.filters, dp_plane=0, out=out, delta_stats=self.stats_b[i])
        eng.event_record(self.EV_CF + i)
        self._pending = i
        if not self.overlap:
            self._tail()

    def _tail(self):
        """dumpSTR after the call filters: statistics of the masked genotypes, locus filters, cohort-wide sums
        (enqueued on the selected queue)."""
        if self._pending is None:
            return
        eng, i = self.eng, self._pending
        self._pending = None
        eng.event_wait(self.EV_CF + i)
        eng.locus_finalize(self.sb.batch, self.stats_b[i])
        eng.locus_filters(self.n_loci, self.stats_b[i], bits_out=self.bits_[i], counters=self.loc_counters_[i],
                          **self.locus_args)
        if self.gather is not None and self.host_group is not None:
            eng.sync()
            self.sums_[i].set(self.host_group.allreduce_sum_i64(self.sums_[i].get()))
            parts = self.host_group.allgather_bytes(self.bits_[i].get().view(np.uint8))
            self.gather.set(np.stack([np.frombuffer(p.tobytes(), dtype=np.uint32) for p in parts]))
        elif self.gather is not None:
            eng.exchange(self.sums_[i], self.bits_[i], self.gather)
        eng.event_record(self.EV_TAIL + i)

    def flush(self):
        """Enqueue the tail of the last step (call before the final synchronisation)."""
        with self.eng.on_queue(1 if self.overlap else 0):
            self._tail()

    def run(self, steps, warmup, barrier=None):
        """warmup untimed steps, then `steps` timed ones; returns (seconds, per-kernel profile)."""
        eng = self.eng
        for _ in range(warmup):
            self.step()
        self.flush()
        eng.sync()
        if barrier:
            barrier()
        eng.profile(True)
        eng.profile_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.flush()
        eng.sync()
        elapsed = time.perf_counter() - t0
        prof = eng.profile_get()
        eng.profile(False)
        return elapsed, prof

    def free(self):
        for st in self.stats_a + self.stats_b:
            for a in (st.allele_count, st.locus_int, st.locus_f64) + tuple(getattr(st, '_owners', ())):
                a.free()
        for a in self.sums_ + self.bits_ + [self.call_outs[0].gt_out, self.call_outs[0].filter_mask, self.gather]:
            if a is not None:
                a.free()
        for co in self.call_outs:
            co.error.free()
            co.sample_totaldp_f64.free()
        for a in list(self.sb.dev.values()) + list(self.sb.batch.arrays.values()):
            a.free()


def exhaustive_check(wl, single_rank_sums):
    """EVERY locus of the shard vs oracle_c on all host threads (oracle/fullsize.py).  The device generator's
    output is read back block by block and is the input of both sides."""
    from oracle import fullsize
    L = wl.L
    i = wl.last
    dev = dict(cnt_a=wl.stats_a[i].allele_count.get()[0], li_a=wl.stats_a[i].locus_int.get()[0],
               lf_a=wl.stats_a[i].locus_f64.get()[0], cnt_b=wl.stats_b[i].allele_count.get()[0],
               li_b=wl.stats_b[i].locus_int.get()[0], lf_b=wl.stats_b[i].locus_f64.get()[0],
               bits=wl.bits.get()[:wl.n_loci])
    if single_rank_sums:
        dev.update(sample_counters=wl.call_out.sample_counters.get(), totaldp=wl.call_out.sample_totaldp.get(),
                   dpmiss=wl.call_out.sample_dp_missing.get(), loc_counters=wl.loc_counters.get())
    gt_d, dp_d, q_d = wl.sb.dev['gt'], wl.sb.dev['dp'], wl.sb.dev['q']

    def fetch_inputs(lo, hi):
        return gt_d.get_rows(lo, hi), [dp_d.get_rows(lo, hi), q_d.get_rows(lo, hi)]

    def fetch_outputs(lo, hi):
        return wl.call_out.gt_out.get_rows(lo, hi), wl.call_out.filter_mask.get_rows(lo, hi)

    assert wl.call_out.error.get()[0] == 0
    t0 = time.perf_counter()
    r = fullsize.check_step(fetch_inputs, fetch_outputs, wl.n_loci, wl.n_samples, wl.sb.tables, wl.filters, 0,
                            wl.locus_args, dev, n_threads=max(1, fullsize.oracle_c.tuned_threads() // max(1, wl.world)),
                            n_pad=wl.sb.n_pad)
    r['seconds'] = time.perf_counter() - t0
    return r


def assoc_extra(wl, seed, no_check, no_cpu, iters=5, group=None, total_loci=None):
    """SURVEY section 8 row f3 / BASELINE configs[4], outside the timed region of the headline
    metric: the associaTR scan (trk_assoc_scan) over this rank's resident genotype tensor, one seeded
    standard-normal trait, every sample in the regression set.  4 algorithmic bytes per call (the GT read).
    A handful of loci is checked against the associaTR oracle.  With N > 1 every rank scans its locus shard
    (the scan has no exchange step: result rows stay with the rank that owns the loci, as the sharded associaTR
    command line writes them in rank order); barrier, `iters` timed passes, max over the ranks."""
    from trtools_amd.synth import pack_assoc_tables
    eng = wl.eng
    n_loci, n_samples = wl.n_loci, wl.n_samples
    alen, rcls = pack_assoc_tables(wl.sb.loci.allele_lens, 2)
    alen_d, rcls_d = eng.upload(alen, np.float64), eng.upload(rcls, np.uint16)
    rng = np.random.default_rng(seed + 77)
    y = rng.normal(size=n_samples)
    y = (y - y.mean()) / y.std()
    vec_d = eng.upload(y[None, :].copy(), np.float64)
    res = None
    eng.profile(True)
    for it in range(iters + 1):
        if it == 1:
            eng.sync()
            if group is not None:
                group.barrier()
            eng.profile_reset()
            t0 = time.perf_counter()
        res = eng.assoc_scan(wl.sb.batch, vec_d, alen_d, rcls_d, non_major_cutoff=20.0, out=res)
    eng.sync()
    wall = (time.perf_counter() - t0) / iters
    world = group.world if group is not None else 1
    if group is not None:
        wall = float(group.allreduce_max_f64(np.array([wall]))[0])
    total_loci = total_loci or n_loci
    prof = eng.profile_get()
    eng.profile(False)
    n, ms = prof['k_assoc_scan']
    nf, msf = prof['k_assoc_finalize']
    scan_ms = ms / max(n, 1)
    cells = n_loci * wl.n_real
    out = {"workload": "associaTR linear-regression scan, %d loci x %d samples x 1 trait (BASELINE configs[4] on %s)"
                       % (total_loci, wl.n_real,
                          "one GPU" if world == 1 else "%d GPUs, %d loci per GPU, no exchange step" % (world, n_loci)),
           "n_gpus": world, "loci_per_s": total_loci / wall, "ms_per_pass": wall * 1e3,
           "kernels_ms": {"k_assoc_scan": scan_ms, "k_assoc_finalize": msf / max(nf, 1)},
           "roofline": {"bound": "hbm", "kernel": "k_assoc_scan", "bytes_per_cell": 4,
                        "achieved": cells * 4 / (scan_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": cells * 4 / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    # THROUGHPUT of a scan over many batches: two passes in flight on two of the context's queues -- the per-locus
    # finaliser of one (a latency chain, 0.12 ms) runs beside the streaming scan of the other.  Identical outputs.
    try:
        res2 = None
        for it in range(2 * iters + 2):
            if it == 2:
                eng.sync()
                t0 = time.perf_counter()
            with eng.on_queue(it % 2):
                if it % 2:
                    res2 = eng.assoc_scan(wl.sb.batch, vec_d, alen_d, rcls_d, non_major_cutoff=20.0, out=res2)
                else:
                    res = eng.assoc_scan(wl.sb.batch, vec_d, alen_d, rcls_d, non_major_cutoff=20.0, out=res)
        eng.sync()
        w2 = (time.perf_counter() - t0) / (2 * iters)      # (this rank's; no collective inside an optional extra)
        same = (np.array_equal(res2.locus_int.get(), res.locus_int.get()) and
                np.array_equal(res2.locus_f64.get(), res.locus_f64.get(), equal_nan=True))
        out["two_queues"] = {"what": "the same pass, two batches in flight on two queues (throughput, not latency; rank 0's "
                                     "time, every rank scanning its shard)",
                             "ms_per_pass": w2 * 1e3, "loci_per_s": total_loci / w2, "identical_outputs": bool(same)}
        for d in (res2.locus_int, res2.locus_f64, res2.allele_count):
            d.free()
    except Exception as e:      # an extra, never a reason to lose the line
        out["two_queues"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if not no_check:
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
        # ... and EVERY locus of the shard against the compiled restatement (oracle_c.c orc_assoc_locus)
        from oracle import fullsize
        gt_d = wl.sb.dev['gt']
        x1 = np.zeros((n_samples, 2))
        x1[:, 1] = 1.0
        t0 = time.perf_counter()
        ra = fullsize.check_assoc(lambda lo, hi: gt_d.get_rows(lo, hi), n_loci, n_samples, wl.sb.tables[0], alen, x1, y,
                                  li, lf, non_major_cutoff=20.0,
                                  n_threads=max(1, fullsize.oracle_c.tuned_threads() // max(1, world)))
        out["parity_loci_checked"] = int(ra['loci'])
        out["parity_loci_regressed"] = int(ra['regressed'])
        out["parity"] = {"checker": "oracle/oracle_c.c orc_assoc_locus on %d host threads, every locus; %d loci also "
                                    "against oracle/associatr_oracle.py" % (ra['threads'], len(idx)),
                         "worst_rel": ra['worst_rel'], "seconds": time.perf_counter() - t0}
        if not no_cpu:
            # the same scan through the oracle port (numpy + scipy, one locus and one OLS fit at a time like the
            # reference), 1 core, on a bounded sample of loci regenerated by the generator's numpy twin
            t0 = time.perf_counter()
            done = 0
            rng2 = np.random.default_rng(seed + 78)
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


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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
    return dict(value=done / el, unit="loci/s", cores=1, kind="port", cpu=cpu_model(),
                sample="%d random loci x %d samples of the same synthetic call set, statSTR (11 stats, "
                       "string alleles) + dumpSTR (3 call filters, 4 locus filters, INFO recompute) through "
                       "oracle/trtools_oracle.py (numpy+scipy, per-locus like the reference), %.1f s"
                       % (done, S, el),
                cells_per_s=done * S / el)


def cpu_baseline_c(wl, n_loci=16384):
    """The C half of the oracle (oracle/oracle_c.c: counts, statistics, exact HWE test, the three threshold call
    filters, recount of the masked genotypes) on a bounded sample of the same call set read back from the device:
    one core, then all cores with the loci split over OpenMP threads inside the library ('single core
    and all cores').  A compiled, per-locus CPU implementation of the same step -- a stronger baseline than the
    numpy port, still only a reported number."""
    from oracle import oracle_c
    n_loci = min(n_loci, wl.n_loci)
    gt = wl.sb.dev['gt'].get_rows(0, n_loci)
    planes = [wl.sb.dev['dp'].get_rows(0, n_loci), wl.sb.dev['q'].get_rows(0, n_loci)]
    off_all = wl.sb.tables[0]
    off = (off_all[:n_loci + 1] - off_all[0]).astype(np.int32)
    lc, sc, cv = (np.ascontiguousarray(t[off_all[0]:off_all[n_loci]]) for t in wl.sb.tables[1:4])
    oracle_c.load()
    out = {}
    for label, nt in (('one_core', 1), ('all_cores', oracle_c.tuned_threads())):
        n = n_loci if nt > 1 else max(64, n_loci // 16)
        o = (off[:n + 1]).astype(np.int32)
        sl = slice(0, int(off[n]))
        g, pl = np.ascontiguousarray(gt[:n]), [np.ascontiguousarray(p[:n]) for p in planes]
        # outputs allocated (and touched) before the clock starts: the region times the C code, not numpy's page faults
        st = (np.zeros(int(off[n]), dtype=np.int32), np.zeros((n, 8), dtype=np.int32), np.zeros((n, 10)))
        co = (np.zeros_like(g), np.zeros((n, wl.n_samples), dtype=np.uint32))
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            oracle_c.batch_stats(g, None, o, lc[sl], sc[sl], cv[sl], n_threads=nt, out=st)                  # statSTR
            g2 = oracle_c.call_filters(g, pl, wl.filters, dp_plane=0, n_threads=nt, out=co)[0]
            oracle_c.batch_stats(g2, None, o, lc[sl], sc[sl], cv[sl], n_threads=nt, out=st)                 # dumpSTR on GT'
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        out[label] = dict(value=n / best, unit="loci/s", cores=nt, cells_per_s=n * wl.n_real / best,
                          sample="%d loci x %d samples, %.2f s (best of 2)" % (n, wl.n_real, best))
    out['kind'] = "port (C restatement, oracle/oracle_c.c, OpenMP over loci inside the library)"
    out['cpu'] = cpu_model()
    out['cores_visible'] = oracle_c.n_cores()
    out['note'] = ("all_cores uses the OpenMP team size that a calibration run found fastest on this host "
                   "(oracle_c.tuned_threads), not necessarily every visible hardware thread")
    return out


def strong_shard_extra(eng, args, loci, t_full_ms, steps):
    """What one of N ranks does in the strong-scaled run, measured on this one GPU: loci [0, 100000/N) of the same
    cohort, the exchange through a 1-rank RCCL communicator (the collective kernels launch and run; only the wire
    is missing).  predicted_efficiency = (t_100k / N) / t_shard."""
    out = {}
    cells_full = args.loci * args.samples
    for n in (2, 4, 8):
        hi = args.loci // n
        wl = Workload(eng, args.seed, args.samples, loci.slice(0, hi), 0, 1, use_comm=True, pipeline_count=True)
        el, prof = wl.run(steps, 3)
        ms = el / steps * 1e3
        kn, kms = prof['k_call_filter']
        cf = kms / max(kn, 1)
        cells = hi * args.samples
        out["N=%d" % n] = {"loci": hi, "ms_per_step": ms, "k_call_filter_ms": cf,
                           "k_call_filter_frac": cells * BYTES_PER_CELL_CALL_FILTER / (cf * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "kernels_ms": {k: (v[1] / v[0] if v[0] else None) for k, v in prof.items() if v[0]},
                           "predicted_efficiency": (t_full_ms / n) / ms,
                           "predicted_loci_per_s_at_N": args.loci / (ms * 1e-3),
                           # the shard's two output planes (hi x S x 4 B each): placed from 256 MB on (Engine.PLACE_MIN_BYTES)
                           "output_planes_bytes_each": hi * wl.n_samples * 4,
                           "output_planes_placed": (wl.placement or {}).get('placed') if wl.placement is not None else
                                                   "not searched (planes below 256 MB: no placement classes to tell apart)",
                           "placement_probe_ms": (wl.placement or {}).get('probe_ms'),
                           "placement_reserved": (wl.placement or {}).get('reserved'),
                           "placement_peak_extra_bytes": (wl.placement or {}).get('peak_extra_bytes')}
        wl.free()
    out["note"] = ("one GPU, 1-rank RCCL communicator: every kernel and collective launch of a rank of the N-GPU "
                   "strong-scaled run, without the xGMI wire time (< 1 MB per step)")
    return out


def compact_outputs_extra(wl, iters=8):
    """The opt-in output set for callers that rebuild their records from the mask (this repository's dumpSTR): a
    one-byte mask per call, no masked-genotype plane (trk_call_out.filter_mask8, gt_out = NULL) -- 12 B read + 1 B
    written per call instead of 12 + 8.  Same filters, same counters and delta outputs; NOT the headline contract
    (SURVEY.md 8d prices the path at 20 B per call), reported beside it."""
    eng = wl.eng
    b = wl.sb.batch
    i = wl.last
    ref_mask = wl.call_out.filter_mask.get_rows(0, 2048)
    ref_counters = wl.call_out.sample_counters.get()
    out = eng.alloc_call_out(b, len(wl.filters), want_gt=False, want_mask=False, want_mask8=True)
    st = eng.alloc_stats(b)
    eng.profile(True)
    for it in range(iters + 2):
        if it == 2:
            eng.sync()
            eng.profile_reset()
        for a in (out.sample_counters, out.sample_totaldp, out.sample_dp_missing):
            a.zero()
        st.allele_count.copy_from(wl.stats_a[i].allele_count)
        st.locus_int.copy_from(wl.stats_a[i].locus_int)
        eng.call_filters(b, wl.planes, wl.filters, dp_plane=0, out=out, delta_stats=st)
    eng.sync()
    n, ms = eng.profile_get()['k_call_filter']
    eng.profile(False)
    ms /= n
    m8 = out.filter_mask8.get_rows(0, 2048)
    want = ((ref_mask & np.uint32(0x7f)) | ((ref_mask >> np.uint32(24)) & np.uint32(0x80))).astype(np.uint8)
    assert np.array_equal(m8, want), "compact mask differs from the 32-bit mask"
    assert np.array_equal(out.sample_counters.get(), ref_counters), "counters differ with the compact outputs"
    assert np.array_equal(st.allele_count.get(), wl.stats_b[i].allele_count.get())
    cells = wl.n_loci * wl.n_real
    res = {"workload": "the call-filter pass of the headline step with trk_call_out.filter_mask8 only (no gt_out, no "
                       "32-bit mask): 13 B per call", "k_call_filter_ms": ms, "bytes_per_cell": 13,
           "achieved_GBs": cells * 13 / (ms * 1e-3) / 1e9, "frac": cells * 13 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "checked": "mask bytes of 2048 loci, every sample counter and every allele count equal the 20 B run's"}
    for a in (out.filter_mask8, out.sample_counters, out.sample_totaldp, out.sample_dp_missing, out.error,
              out.sample_totaldp_f64, st.allele_count, st.locus_int, st.locus_f64):
        a.free()
    return res


def in_place_extra(wl, iters=6):
    """The call-filter pass of the headline step with the masked genotypes written IN PLACE (trk_call_out.gt_out ==
    trk_batch.gt, as dumpSTR.py:721-727 updates its record): 12 B read + the 4 B mask + only the 16-byte chunks that hold
    a filtered call.  One big write stream: the pass no longer depends on which allocations its two output planes are
    (profiles/r03_notes.md section 22).  Opt-in; the headline keeps SURVEY.md 8d's two full output planes (20 B)."""
    eng = wl.eng
    b = wl.sb.batch
    i = wl.last
    gt = b.arrays['gt']
    keep = eng.empty(gt.shape, gt.dtype)
    keep.copy_from(gt)
    ref_rows = 2048
    ref_mask = wl.call_out.filter_mask.get_rows(0, ref_rows)
    ref_gt = wl.call_out.gt_out.get_rows(0, ref_rows)
    ref_counters = wl.call_out.sample_counters.get()
    out = eng.alloc_call_out(b, len(wl.filters), in_place=True)
    st = eng.alloc_stats(b)
    eng.profile(True)
    try:
        for it in range(iters + 2):
            if it == 2:
                eng.sync()
                eng.profile_reset()
            gt.copy_from(keep)
            for a in (out.sample_counters, out.sample_totaldp, out.sample_dp_missing):
                a.zero()
            st.allele_count.copy_from(wl.stats_a[i].allele_count)
            st.locus_int.copy_from(wl.stats_a[i].locus_int)
            eng.call_filters(b, wl.planes, wl.filters, dp_plane=0, out=out, delta_stats=st)
        eng.sync()
        n, ms = eng.profile_get()['k_call_filter']
        eng.profile(False)
        ms /= n
        got_gt = gt.get_rows(0, ref_rows)
        assert np.array_equal(got_gt, ref_gt), "in-place genotypes differ from the two-plane run's masked genotypes"
        assert np.array_equal(out.filter_mask.get_rows(0, ref_rows), ref_mask), "mask differs with in-place genotypes"
        assert np.array_equal(out.sample_counters.get(), ref_counters), "counters differ with in-place genotypes"
        assert np.array_equal(st.allele_count.get(), wl.stats_b[i].allele_count.get())
    finally:
        gt.copy_from(keep)
        eng.sync()
    # chunks (4 calls) that hold a call which is made and filtered: the ones the pass stores
    filt = ((ref_mask & np.uint32(0x7fffffff)) != 0) & ((ref_mask >> np.uint32(31)) == 0)
    S4 = filt.shape[1] // 4 * 4
    touched = float(filt[:, :S4].reshape(filt.shape[0], -1, 4).any(axis=2).mean())
    bpc = 12 + 4 + 4 * touched
    cells = wl.n_loci * wl.n_real
    res = {"workload": "the call-filter pass of the headline step with gt_out == gt (in place): 12 B read, the 4 B mask "
                       "and %.1f %% of the genotype chunks written" % (100 * touched),
           "k_call_filter_ms": ms, "bytes_per_cell": round(bpc, 3), "chunks_written_frac": touched,
           "achieved_GBs": cells * bpc / (ms * 1e-3) / 1e9, "frac": cells * bpc / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "checked": "genotypes and mask words of %d loci, every sample counter and every allele count equal the 20 B run's"
                      % ref_rows}
    for a in (keep, out.filter_mask, out.sample_counters, out.sample_totaldp, out.sample_dp_missing, out.error,
              out.sample_totaldp_f64, st.allele_count, st.locus_int, st.locus_f64):
        a.free()
    return res


def qc_reduce_extra(wl, no_check, iters=8):
    """qcSTR's per-sample / per-locus call counts and quality sums of the headline cohort in one
    pass (trk_qc_reduce: genotypes + the Q plane, 8 B per call; three launches: scan + two small finishers)."""
    eng = wl.eng
    b = wl.sb.batch
    q = wl.sb.dev['q']
    res = eng.qc_reduce(b, q)
    eng.sync()
    eng.timer_start(0)
    for _ in range(iters):
        for k, v in res.items():
            if k != '_keep':
                v.free()
        res = eng.qc_reduce(b, q)
    eng.timer_stop(0)
    ms = eng.timer_ms(0) / iters
    cells = wl.n_loci * wl.n_real
    out = {"workload": "qcSTR reductions (calls + quality sums per sample and per locus), %d loci x %d samples"
                       % (wl.n_loci, wl.n_real), "ms_per_pass": ms, "bytes_per_cell": 8,
           "achieved_GBs": cells * 8 / (ms * 1e-3) / 1e9, "frac": cells * 8 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    if not no_check:
        from oracle import trtools_oracle as orc
        n = 64
        gt = wl.sb.dev['gt'].get_rows(0, n).reshape(n, wl.n_samples, 2)
        qh = q.get_rows(0, n).reshape(n, wl.n_samples)
        lc, ls, ln = res['locus_calls'].get()[:n], res['locus_qual_sum'].get()[:n], res['locus_qual_n'].get()[:n]
        for l in range(n):
            calls, qq, _ = orc.qc_record(gt[l], qh[l].reshape(-1, 1))
            assert int(calls.sum()) == int(lc[l]) and int(ln[l]) == wl.n_samples, "qc reduce: locus %d counts" % l
            assert abs(float(qq.astype(np.float64).sum()) - float(ls[l])) <= 1e-9 * max(1.0, abs(float(ls[l])))
        assert int(res['sample_calls'].get().sum()) == int(res['locus_calls'].get().sum())
        out["parity"] = "%d loci against oracle/trtools_oracle.qc_record; sum over samples == sum over loci" % n
    for k, v in res.items():
        if k != '_keep':
            v.free()
    return out


def config1_extra(eng, no_check, iters=50):
    """BASELINE configs[1]: statSTR full statistics on a synthetic HipSTR-shape call set, 10k loci x 1k samples."""
    from trtools_amd.synth import SynthBatch
    from trtools_amd import _lib as L
    Lc, S = 10000, 1000
    sb = SynthBatch(eng, Lc, S, seed=20260928 + 1, planes=())
    res = eng.alloc_stats(sb.batch)
    # the pass as a command line runs it (no event brackets between its launches) ...
    for it in range(iters + 3):
        if it == 3:
            eng.sync()
            t0 = time.perf_counter()
        eng.locus_stats(sb.batch, out=res)
    eng.sync()
    w = (time.perf_counter() - t0) / iters
    # ... and once more with HIP events around each launch for the per-kernel figures (the brackets cost a few us)
    eng.profile(True)
    for it in range(iters + 3):
        if it == 3:
            eng.sync()
            eng.profile_reset()
            t0 = time.perf_counter()
        eng.locus_stats(sb.batch, out=res)
    eng.sync()
    w_prof = (time.perf_counter() - t0) / iters
    pg = eng.profile_get()
    eng.profile(False)
    cnt_ms = pg['k_locus_count'][1] / pg['k_locus_count'][0]
    fin_ms = pg['k_locus_finalize'][1] / pg['k_locus_finalize'][0]
    fused = os.environ.get('TRK_FUSED_STATS') != '0'
    out = {"workload": "statSTR full statistics, HipSTR shape, %d loci x %d samples (BASELINE configs[1])" % (Lc, S),
           "ms_per_pass": w * 1e3, "loci_per_s": Lc / w, "calls_per_s": Lc * S / w,
           "ms_per_pass_with_event_brackets": w_prof * 1e3,
           "launches_per_pass": ("2: k_locus_count_v3<4,4,true> (count + finaliser), k_hwe_test_slots" if fused else
                                 "5: count, counter reset, finaliser, HWE tests, HWE serial remainder"),
           "kernels_ms": ({"k_locus_count_v3<fin>": cnt_ms, "k_hwe_test_slots": fin_ms} if fused else
                          {"k_locus_count": cnt_ms, "k_locus_finalize+k_hwe_test": fin_ms}),
           "roofline": {"bound": "hbm", "kernel": ("k_locus_count_v3<4,4,true> (the finaliser is its epilogue: the time "
                                                   "is count + finaliser)" if fused else "k_locus_count"),
                        "bytes_per_cell": 4,
                        "achieved": Lc * S * 4 / (cnt_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": Lc * S * 4 / (cnt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "note": "40 MB per launch: 5 us at the HBM peak -- launch-latency regime; the long-stream "
                                "figure for this row length is extras.short_rows"}}
    # THROUGHPUT of the same pass when a command line keeps several batches in flight: the pass is a chain of
    # short latency-bound kernels (count + finaliser 26 us, HWE tests 20-23) that leave most of the chip idle;
    # four independent batches round-robin over the context's four in-order queues overlap them
    NQ = 4
    sets = [res] + [eng.alloc_stats(sb.batch) for _ in range(NQ - 1)]
    for it in range(iters + NQ):
        if it == NQ:
            eng.sync()
            t0 = time.perf_counter()
        with eng.on_queue(it % NQ):
            eng.locus_stats(sb.batch, out=sets[it % NQ])
    eng.sync()
    wq = (time.perf_counter() - t0) / iters
    out["four_queues"] = {"what": "the same pass, four batches in flight on the context's four queues (throughput, not "
                                  "latency)", "ms_per_pass": wq * 1e3, "loci_per_s": Lc / wq}
    for st in sets[1:]:
        if not no_check:
            assert np.array_equal(st.locus_f64.get(), res.locus_f64.get(), equal_nan=True)
            assert np.array_equal(st.allele_count.get(), res.allele_count.get())
        for a in (st.allele_count, st.locus_int, st.locus_f64):
            a.free()
    if not no_check:
        from oracle import fullsize, oracle_c
        off, lc, sc, cv = sb.tables
        cnt_o, oi, of = oracle_c.batch_stats(sb.dev['gt'].get(), None, off, lc, sc, cv, n_threads=oracle_c.n_cores())
        fullsize._cmp_stats('config1', 0, res.allele_count.get()[0], res.locus_int.get()[0], res.locus_f64.get()[0],
                            cnt_o, oi, of, S)
        out["parity_rows_checked"] = Lc
    # the same row length as a long stream (1M loci x 1k samples = 4 GB): the short-row rate of the count kernel
    for a in (res.allele_count, res.locus_int, res.locus_f64):
        a.free()
    for a in list(sb.dev.values()) + list(sb.batch.arrays.values()):
        a.free()
    return out


def short_rows_extra(eng, iters=5):
    """The count kernel on 1000-sample rows as a long stream (400k loci x 1k samples = 1.6 GB per launch)."""
    from trtools_amd.synth import SynthBatch, make_loci
    Lc, S = 400000, 1000
    base = make_loci(10000, S, 20260928 + 1)
    # tile the 10k-locus table 40 times (the per-call hash still differs by locus)
    import copy
    loci = copy.copy(base)
    reps = Lc // 10000
    loci.motifs = base.motifs * reps
    loci.allele_strs = base.allele_strs * reps
    loci.allele_lens = base.allele_lens * reps
    nA = int(base.allele_off[-1])
    loci.allele_off = np.concatenate([base.allele_off[:-1] + r * nA for r in range(reps)] +
                                     [np.array([reps * nA])]).astype(np.int32)
    loci.cdf24 = np.tile(base.cdf24, reps)
    loci.miss_thr16 = np.tile(base.miss_thr16, reps)
    loci.inbreed_thr16 = np.tile(base.inbreed_thr16, reps)
    sb = SynthBatch(eng, Lc, S, seed=20260928 + 11, planes=(), loci=loci)
    res = eng.alloc_stats(sb.batch)
    eng.profile(True)
    for it in range(iters + 2):
        if it == 2:
            eng.sync()
            eng.profile_reset()
        eng.locus_stats(sb.batch, out=res, count_only=True)
    eng.sync()
    pg = eng.profile_get()
    eng.profile(False)
    ms = pg['k_locus_count'][1] / pg['k_locus_count'][0]
    out = {"workload": "k_locus_count on %d loci x %d samples (1000-sample rows as a long stream)" % (Lc, S),
           "k_locus_count_ms": ms, "achieved_GBs": Lc * S * 4 / (ms * 1e-3) / 1e9,
           "frac": Lc * S * 4 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    for a in (res.allele_count, res.locus_int, res.locus_f64):
        a.free()
    for a in list(sb.dev.values()) + list(sb.batch.arrays.values()):
        a.free()
    return out


def gangstr_filters():
    """BuildCallFilters order for GangSTR (dumpSTR.py:819-836): min/max DP, min Q, expansion-prob het / hom / total,
    span-only, span+bound-only, bad CI."""
    from trtools_amd import _lib as L
    return [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_GT, plane_a=0, thr=60), dict(op=L.F_LT, plane_a=1, thr=0.9),
            dict(op=L.F_CALLED_LT, plane_a=2, col_a=1, thr=0.05), dict(op=L.F_CALLED_LT, plane_a=2, col_a=2, thr=0.05),
            dict(op=L.F_CALLED_SUM_LT, plane_a=2, col_a=1, col_a2=2, thr=0.2),
            dict(op=L.F_CALLED_EQ, plane_a=3, col_a=1, plane_b=0, col_b=0),
            dict(op=L.F_CALLED_SUM_EQ, plane_a=3, col_a=1, col_a2=3, plane_b=0, col_b=0),
            dict(op=L.F_CALLED_OUTSIDE_CI, plane_a=4, plane_b=5)]


def config2_extra(eng, no_check, iters=5):
    """BASELINE configs[2]: dumpSTR call + locus filters on a synthetic GangSTR-shape call set, 50k loci x 5k samples,
    the nine GangSTR call filters + four locus filters.  FORMAT planes are handed over planar ([k, L, S], what
    Engine.upload_plane produces from cyvcf2-shaped [L, S, k] arrays): every column streams as 16-byte vectors and
    unused columns (QEXP[0], RC[0], RC[2]) are never read -- 4 (GT) + 4 + 4 + 8 + 8 + 8 + 16 = 52 B read,
    8 B written per call."""
    from trtools_amd.synth import SynthBatch
    from trtools_amd import _lib as L
    Lc, S = 50000, 5000
    sb = SynthBatch(eng, Lc, S, seed=20260928 + 2, planes=('dp', 'q'), pure_repeats=True)
    sb.add_gangstr_planes()
    # rows on 128-byte boundaries (24 no-call padding samples: 5024 per device row), as compute.DeviceCompute uploads
    # a cohort -- a 5000-sample row is 20 000 bytes, three rows of four start inside a cache line (planar pass
    # 3.16 -> 2.91 ms, same box); every rate counts the 5000 real samples
    sb.pad_rows(max(4, int(os.environ.get('TRK_ROW_ALIGN', '32')) & ~3))
    names = ['dp', 'q', 'qexp', 'rc', 'repcn', 'repci']
    inter = [sb.dev[n] for n in names]
    planes = [eng.planarize(p) for p in inter]
    filters = gangstr_filters()
    locus_args = dict(min_callrate=0.8, min_hwep=1e-3, min_het=0.05, max_het=0.9, use_length=False)
    st = eng.alloc_stats(sb.batch)
    out_c = eng.alloc_call_out(sb.batch, len(filters))
    bits = eng.empty((Lc,), np.uint32)
    loc = eng.zeros((L.TRK_LC_COLS,), np.int64)
    eng.profile(True)
    for it in range(iters + 1):
        if it == 1:
            eng.sync()
            eng.profile_reset()
            t0 = time.perf_counter()
        for a in (out_c.sample_counters, out_c.sample_totaldp, out_c.sample_dp_missing, loc):
            a.zero()
        eng.locus_stats(sb.batch, out=st, count_only=True)
        eng.call_filters(sb.batch, planes, filters, dp_plane=0, out=out_c, delta_stats=st)
        eng.locus_finalize(sb.batch, st)
        eng.locus_filters(Lc, st, bits_out=bits, counters=loc, **locus_args)
    eng.sync()
    w = (time.perf_counter() - t0) / iters
    pg = eng.profile_get()
    eng.profile(False)
    cf = pg['k_call_filter'][1] / pg['k_call_filter'][0]
    moved = 4 + 4 + 4 + 8 + 8 + 8 + 16 + 8   # GT, DP, Q, QEXP[1:3], RC[1], RC[3], REPCN, REPCI; GT' + mask
    out = {"workload": "dumpSTR, GangSTR shape, 9 call filters + 4 locus filters, %d loci x %d samples "
                       "(BASELINE configs[2]), FORMAT planes planar, %d samples per device row" % (Lc, S, sb.n_dev),
           "ms_per_pass": w * 1e3, "loci_per_s": Lc / w, "calls_per_s": Lc * S / w,
           "kernels_ms": {k: (v[1] / v[0]) for k, v in pg.items() if v[0]},
           "roofline": {"bound": "hbm", "kernel": "k_call_filter_gs<true,true,511>", "bytes_per_cell_moved": moved,
                        "bytes_per_cell_nominal": 72,
                        "achieved": Lc * S * moved / (cf * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": Lc * S * moved / (cf * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    if not no_check:
        from oracle import fullsize
        dev = dict(cnt_a=None, cnt_b=st.allele_count.get()[0], li_b=st.locus_int.get()[0], lf_b=st.locus_f64.get()[0],
                   bits=bits.get(), sample_counters=out_c.sample_counters.get(), totaldp=out_c.sample_totaldp.get(),
                   dpmiss=out_c.sample_dp_missing.get(), loc_counters=loc.get())
        assert out_c.error.get()[0] == 0

        def fetch_inputs(lo, hi):
            return sb.dev['gt'].get_rows(lo, hi), [p.get_rows(lo, hi) for p in inter]

        def fetch_outputs(lo, hi):
            return out_c.gt_out.get_rows(lo, hi), out_c.filter_mask.get_rows(lo, hi)

        t0 = time.perf_counter()
        r = fullsize.check_step(fetch_inputs, fetch_outputs, Lc, sb.n_dev, sb.tables, filters, 0, locus_args, dev,
                                block=2048, n_pad=sb.n_pad)
        out["parity_rows_checked"] = r['loci']
        out["parity_calls_bit_for_bit"] = r['calls_bit_for_bit']
        out["parity_worst_float_rel"] = r['worst_float_rel']
        out["parity_seconds"] = time.perf_counter() - t0
    for a in (planes + [st.allele_count, st.locus_int, st.locus_f64, out_c.gt_out, out_c.filter_mask,
                        out_c.sample_counters, out_c.sample_totaldp, out_c.sample_dp_missing, out_c.error,
                        out_c.sample_totaldp_f64, bits, loc] + list(sb.dev.values()) + list(sb.batch.arrays.values())):
        a.free()
    return out


def end_to_end_extra(eng, seed):
    """SURVEY.md 8(d) "report both kernel-only and end-to-end": (a) statSTR's command line from a bgzipped text VCF
    (17 000 loci x 5000 samples = 1.02 GB of text, GT:DP:Q, written here) to its table of 11 statistics -- native reader, native batch
    harmoniser, upload from pinned staging, kernels, download, native row formatter; (b) a packed host batch
    (4096 loci x 10000 samples) through upload + statistics + download, i.e. the device path with PCIe included,
    from pageable and from pinned host memory.  Never part of `value`."""
    import tempfile
    from trtools_amd import runtime, synth
    from trtools_amd.batch import HostBatch
    from trtools_amd.bgzf import BgzfWriter
    from trtools_amd.compute import DeviceCompute
    from trtools_amd.statSTR import statSTR
    out = {}
    comp = DeviceCompute(engine=eng)
    old = runtime.set_compute(comp)
    try:
        # 17 000 loci x 5 000 samples = 1.02 GB of text (the end-to-end rows at >= 1 GB): 1000 distinct
        # records are rendered (the sample columns of a record are ~60 KB of text) and written 17 times over with their
        # own POS / ID / START / END -- the command lines read, harmonise, count and write every record as any other
        L0, TILES, S = 1000, int(os.environ.get('TRK_E2E_TILES', '17')), 5000
        Lc = L0 * TILES
        tmp = tempfile.mkdtemp(prefix='trk_e2e_')
        path = os.path.join(tmp, 'synth.vcf.gz')
        loci = synth.make_loci(L0, S, seed=5)
        recs = []
        for l0 in range(0, L0, 64):
            idx = np.arange(l0, min(L0, l0 + 64))
            rows = synth.cells_numpy(5, loci, idx, S)
            for r, l in enumerate(idx):
                strs = loci.allele_strs[l]
                g0 = np.where(rows['gt'][r, :, 0] < 0, '.', rows['gt'][r, :, 0].astype(str))
                g1 = np.where(rows['gt'][r, :, 1] < 0, '.', rows['gt'][r, :, 1].astype(str))
                dp = np.where(rows['dp'][r] == -2147483648, '.', rows['dp'][r].astype(str))
                q = np.where(np.isnan(rows['q'][r]), '.', np.char.mod('%g', rows['q'][r]))
                cols = np.char.add(np.char.add(np.char.add(np.char.add(g0, '|'), g1), ':'),
                                   np.char.add(np.char.add(dp, ':'), q))
                recs.append((strs[0], ','.join(strs[1:]) or '.', len(loci.motifs[l]), '\tGT:DP:Q\t' + '\t'.join(cols) + '\n'))
        with BgzfWriter(path, level=1) as fh:
            fh.write('##fileformat=VCFv4.1\n##command=HipSTR-v0.6.2 --synthetic\n')
            for k in ('START', 'END', 'PERIOD'):
                fh.write('##INFO=<ID=%s,Number=1,Type=Integer,Description="%s">\n' % (k, k))
            for k, t in (('GT', 'String'), ('DP', 'Integer'), ('Q', 'Float')):
                fh.write('##FORMAT=<ID=%s,Number=1,Type=%s,Description="%s">\n' % (k, t, k))
            fh.write('#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('S%05d' % i for i in range(S)) + '\n')
            for t in range(TILES):
                block = []
                for l, (ref, alts, period, tail) in enumerate(recs):
                    gl = t * L0 + l
                    pos = 1000 + 500 * gl
                    block.append('chr1\t%d\tSTR_%d\t%s\t%s\t.\t.\tSTART=%d;END=%d;PERIOD=%d%s'
                                 % (pos, gl, ref, alts, pos, pos + len(ref) - 1, period, tail))
                fh.write(''.join(block))
        del recs

        def clear_outputs(prefix):
            # (truncating the previous run's 1.5 GB output is 0.15 s of open(): not the command line's time)
            for f in os.listdir(tmp):
                if f.startswith(prefix + '.'):
                    os.remove(os.path.join(tmp, f))
        ns = argparse.Namespace(vcf=path, out=os.path.join(tmp, 'stat'), vcftype='hipstr', samples=None,
                                sample_prefixes=None, plot_afreq=False, region=None, thresh=True, afreq=True,
                                acount=True, hwep=True, het=True, entropy=True, mean=True, mode=True, var=True,
                                numcalled=True, use_length=False, precision=4, nalleles=True, nalleles_thresh=0.01,
                                only_passing=False)
        best = None
        for _ in range(3):
            clear_outputs('stat')
            t0 = time.perf_counter()
            rc = statSTR.main(ns)
            el = time.perf_counter() - t0
            assert rc == 0
            best = el if best is None else min(best, el)
        out["statstr_cli_text_vcf_to_table"] = {
            "workload": "statSTR (11 statistics) on a bgzipped text VCF, %d loci x %d samples, GT:DP:Q (%.0f MB "
                        "compressed), to its .tab file" % (Lc, S, os.path.getsize(path) / 1e6),
            "seconds": best, "loci_per_s": Lc / best, "calls_per_s": Lc * S / best,
            "rows": sum(1 for _ in open(ns.out + '.tab')) - 1, "path": dict(statSTR.LAST_RUN)}
        # dumpSTR on the same file: three call filters + four locus filters, output VCF (~180 MB) and the two logs
        from trtools_amd.dumpSTR import dumpSTR
        argv = sys.argv
        sys.argv = ['dumpSTR', '--vcf', path, '--out', os.path.join(tmp, 'dump'), '--vcftype', 'hipstr',
                    '--hipstr-min-call-DP', '10', '--hipstr-max-call-DP', '55', '--hipstr-min-call-Q', '0.9',
                    '--min-locus-callrate', '0.8', '--min-locus-hwep', '0.001', '--min-locus-het', '0.05',
                    '--max-locus-het', '0.9']
        try:
            dargs = dumpSTR.getargs()
        finally:
            sys.argv = argv
        best = None
        for _ in range(3):
            clear_outputs('dump')
            t0 = time.perf_counter()
            rc = dumpSTR.main(dargs)
            el = time.perf_counter() - t0
            assert rc == 0
            best = el if best is None else min(best, el)
        out["dumpstr_cli_text_vcf_to_vcf"] = {
            "workload": "dumpSTR (min/max call DP, min call Q; call rate, HWE, het low/high) on the same file, to an "
                        "output VCF of %.0f MB + sample and locus logs" % (os.path.getsize(os.path.join(tmp, 'dump.vcf')) / 1e6),
            "seconds": best, "loci_per_s": Lc / best, "calls_per_s": Lc * S / best, "path": dict(dumpSTR.LAST_RUN)}
        # round 6: the same command line with --zip -- the output as BGZF members made by libtrk (trk_bgzf_compress) and the
        # tabix index from the places the writer noted (rounds 1-5: zlib members on a thread pool of the interpreter and a
        # scan of the finished file: 7.7 s)
        zargs = argparse.Namespace(**vars(dargs))
        zargs.zip, zargs.out = True, os.path.join(tmp, 'zdump')
        best = None
        for _ in range(2):
            clear_outputs('zdump')
            t0 = time.perf_counter()
            rc = dumpSTR.main(zargs)
            el = time.perf_counter() - t0
            assert rc == 0
            best = el if best is None else min(best, el)
        out["dumpstr_cli_zip"] = {
            "workload": "the same dumpSTR run with --zip: a bgzipped output VCF of %.0f MB + its .tbi"
                        % (os.path.getsize(zargs.out + '.vcf.gz') / 1e6),
            "seconds": best, "loci_per_s": Lc / best, "calls_per_s": Lc * S / best,
            "index_bytes": os.path.getsize(zargs.out + '.vcf.gz.tbi')}
        clear_outputs('zdump')
        # ... and with the members DEFLATED ON THE DEVICE (TRK_DEVICE_DEFLATE=1: trk_deflate_bgzf -- one wave per 16 KB member;
        # files ~20 % larger than level 6's, a seventh of its CPU seconds)
        os.environ['TRK_DEVICE_DEFLATE'] = '1'
        try:
            best = None
            for _ in range(2):
                clear_outputs('zdump')
                t0 = time.perf_counter()
                rc = dumpSTR.main(zargs)
                el = time.perf_counter() - t0
                assert rc == 0
                best = el if best is None else min(best, el)
            out["dumpstr_cli_zip_device_deflate"] = {
                "workload": "the same --zip run with the members made on the device: an output of %.0f MB + its .tbi"
                            % (os.path.getsize(zargs.out + '.vcf.gz') / 1e6),
                "seconds": best, "loci_per_s": Lc / best, "calls_per_s": Lc * S / best}
        finally:
            del os.environ['TRK_DEVICE_DEFLATE']
        clear_outputs('zdump')
        # round 6: associaTR's command line (BASELINE configs[4]'s caller) on the same file, one trait: the batch pipeline
        # (native reader -> batch harmoniser -> device parse / inflate -> one scan per batch -> rows)
        from trtools_amd.associaTR import associaTR
        import contextlib
        import io
        tr_path = os.path.join(tmp, 'traits.npy')
        np.save(tr_path, np.random.default_rng(5).normal(size=(S, 2)))
        sys.argv = ['associaTR', os.path.join(tmp, 'assoc.tsv'), path, 'pheno', tr_path, '--same-samples', '--vcftype', 'hipstr']
        try:
            aargs = associaTR.getargs()
        finally:
            sys.argv = argv
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
                warnings.simplefilter('ignore')
                associaTR.main(aargs)
            el = time.perf_counter() - t0
            best = el if best is None else min(best, el)
        out["associatr_cli_text_vcf_to_table"] = {
            "workload": "associaTR (one trait, no covariates) on the same file, to its table",
            "seconds": best, "loci_per_s": Lc / best, "calls_per_s": Lc * S / best,
            "rows": sum(1 for _ in open(os.path.join(tmp, 'assoc.tsv'))) - 1, "path": dict(associaTR.LAST_RUN)}
        # BASELINE configs[2] through text at reduced scale: a GangSTR-shape file (GT:DP:Q:REPCN:REPCI:RC:QEXP), the
        # nine GangSTR call filters + four locus filters, output VCF + logs
        Lg, Sg = 400, 2000
        gpath = os.path.join(tmp, 'gangstr.vcf')
        gl = synth.make_loci(Lg, Sg, seed=7, pure_repeats=True)
        gidx = np.arange(Lg)
        grows = synth.cells_numpy(7, gl, gidx, Sg)
        synth.render_vcf(gpath, gl, grows, caller='gangstr',
                         extra=synth.gangstr_planes_numpy(7, gl, gidx, Sg, grows['gt'], grows['dp'], 0))
        sys.argv = ['dumpSTR', '--vcf', gpath, '--out', os.path.join(tmp, 'gdump'), '--vcftype', 'gangstr',
                    '--gangstr-min-call-DP', '10', '--gangstr-max-call-DP', '60', '--gangstr-min-call-Q', '0.9',
                    '--gangstr-expansion-prob-het', '0.05', '--gangstr-expansion-prob-hom', '0.05',
                    '--gangstr-expansion-prob-total', '0.2', '--gangstr-filter-span-only',
                    '--gangstr-filter-spanbound-only', '--gangstr-filter-badCI', '--min-locus-callrate', '0.8',
                    '--min-locus-hwep', '0.001', '--min-locus-het', '0.05', '--max-locus-het', '0.9']
        try:
            gargs = dumpSTR.getargs()
        finally:
            sys.argv = argv
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            rc = dumpSTR.main(gargs)
            el = time.perf_counter() - t0
            assert rc == 0
            best = el if best is None else min(best, el)
        out["dumpstr_cli_gangstr_nine_filters"] = {
            "workload": "dumpSTR with the nine GangSTR call filters + four locus filters (BASELINE configs[2] at reduced "
                        "scale) on a GangSTR-shape text VCF, %d loci x %d samples (%.0f MB), to an output VCF + logs"
                        % (Lg, Sg, os.path.getsize(gpath) / 1e6),
            "seconds": best, "loci_per_s": Lg / best, "calls_per_s": Lg * Sg / best, "path": dict(dumpSTR.LAST_RUN)}
        for f in os.listdir(tmp):
            os.remove(os.path.join(tmp, f))
        os.rmdir(tmp)
        # (b) packed batch -> results, PCIe included
        Lb, Sb = 4096, 10000
        lo2 = synth.make_loci(Lb, Sb, seed + 9)
        rows = synth.cells_numpy(seed + 9, lo2, np.arange(Lb), Sb)
        off, lc, sc, cv = synth.pack_alleles(lo2.allele_lens, lo2.allele_strs)
        res = {}
        for kind in ('pageable', 'pinned'):
            if kind == 'pinned':
                gt = eng.host_buffer(rows['gt'].nbytes)[:rows['gt'].nbytes].view(np.int16).reshape(rows['gt'].shape)
                gt[...] = rows['gt']
            else:
                gt = rows['gt']
            hb = HostBatch.from_tables(gt, np.full(Lb, 2, dtype=np.uint8), off, lc, sc, cv)
            comp.locus_stats(hb)
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                comp.locus_stats(hb)
                el = time.perf_counter() - t0
                best = el if best is None else min(best, el)
            res[kind] = {"seconds": best, "loci_per_s": Lb / best, "calls_per_s": Lb * Sb / best,
                         "host_to_device_GBs": rows['gt'].nbytes / best / 1e9}
        res["workload"] = ("statSTR statistics of one packed host batch, %d loci x %d samples (%.0f MB of genotypes): "
                           "upload + kernels + download of the per-locus results" % (Lb, Sb, rows['gt'].nbytes / 1e6))
        out["packed_batch_incl_pcie"] = res
    finally:
        runtime.set_compute(old)
    return out


def box_stream_probe(eng, wl):
    """The box's own yardstick (it overwrites GT' / mask: call it after everything that reads them): the bare 12 B-in /
    8 B-out stream of the call-filter pass's shape on the same planes (trk_stream_probe), and the clocks the runtime
    reports."""
    try:
        pms = eng.stream_probe(wl.sb.dev['gt'], wl.sb.dev['dp'], wl.sb.dev['q'], wl.call_out.gt_out,
                               wl.call_out.filter_mask, wl.n_loci, wl.n_samples, reps=5)
        box_probe = {"what": "k_stream_probe<3,2>: three 16 B/lane nontemporal input streams, two output streams, "
                             "the call-filter pass's tiling and launch geometry (cf_geometry), no arithmetic",
                     "avg_launch_ms": pms,
                     "achieved": wl.n_loci * wl.n_real * BYTES_PER_CELL_CALL_FILTER / (pms * 1e-3) / 1e9,
                     "unit": "GB/s"}
        box_probe["frac"] = box_probe["achieved"] / HBM_PEAK_GBS
        box_probe.update(eng.device_clocks())
    except Exception as e:      # an aid, never a reason to lose the line
        box_probe = {"error": str(e)[:200]}
    return box_probe


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON: native libraries print there too (RCCL's version banner at
    # communicator creation), so file descriptor 1 is pointed at stderr for the run and the line goes to a copy
    # of the original descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    group = None
    use_dist = world > 1 or bool(os.environ.get('TRK_FORCE_DIST'))   # TRK_FORCE_DIST: exercise the
    # collective path on one rank.  TRK_BENCH_COMM=host: the step's exchange through host copies over the socket
    # group instead of RCCL; TRK_BENCH_SHARE_DEVICE=1: every rank on device 0 -- together they rehearse the whole
    # N > 1 control flow (sharding, barrier, max-over-ranks, per-rank every-locus check, cohort-sum assertion) with
    # several processes on the ONE GPU a builder can reach (RCCL refuses two ranks of one communicator on a device)
    comm_mode = os.environ.get('TRK_BENCH_COMM', 'rccl')
    share_device = bool(os.environ.get('TRK_BENCH_SHARE_DEVICE'))
    from trtools_amd.engine import Engine
    from trtools_amd.synth import make_loci
    from trtools_amd.dist import locus_shard, SocketGroup
    if use_dist:
        group = SocketGroup(rank, world)     # rendezvous / barrier / small host reductions over TCP: no torch
    eng = Engine(0 if share_device else local_rank)
    comm_note = None
    if use_dist and comm_mode != 'host':
        # RCCL, one rank per GPU.  If the communicator cannot be built on ANY rank (the ranks agree over the socket
        # group), the run goes on with the step's exchange through host copies and says so in the line -- a scaling
        # figure with the wire named is worth more than none.
        err, uid = '', b''
        if rank == 0:
            try:
                uid = eng.comm_unique_id()
            except Exception as e:
                err = "%s: %s" % (type(e).__name__, e)
        uid = group.broadcast_bytes(uid)     # (every rank takes part whatever happened on rank 0)
        if uid:
            try:
                eng.comm_init(rank, world, uid)
            except Exception as e:
                err = "%s: %s" % (type(e).__name__, e)
        elif not err:
            err = "no communicator id from rank 0"
        n_bad = int(group.allreduce_sum_i64(np.array([1 if err else 0], dtype=np.int64))[0])
        if n_bad:
            errs = group.allgather_bytes(np.frombuffer(err.encode()[:400], dtype=np.uint8))
            first = next((bytes(b.tobytes()).decode(errors='replace') for b in errs if len(b)), '')
            comm_mode = 'host'
            comm_note = "RCCL communicator not built on %d of %d ranks (%s): exchange through host copies" % (
                n_bad, world, first)
            print(comm_note, file=sys.stderr)
    # the cohort's per-locus tables (every rank builds the same ones from the seed)
    loci = make_loci(args.loci, args.samples, args.seed)
    if args.scaling == 'strong':
        lo, hi = locus_shard(args.loci, rank, world)
        my_loci, locus_base, total_loci = (loci.slice(lo, hi) if world > 1 else loci), lo, args.loci
    else:
        my_loci, locus_base, total_loci = loci, rank * args.loci, args.loci * world
    wl = Workload(eng, args.seed, args.samples, my_loci, locus_base, world, use_comm=use_dist,
                  overlap=os.environ.get('TRK_BENCH_OVERLAP', '1') != '0', gather_loci=-(-args.loci // world),
                  pipeline_count=(world > 1) or bool(os.environ.get('TRK_BENCH_PIPE_COUNT')),
                  host_group=group if (use_dist and comm_mode == 'host') else None)

    def barrier():
        if group is not None:
            group.barrier()

    elapsed, prof = wl.run(args.steps, args.warmup, barrier)
    if group is not None:
        elapsed = float(group.allreduce_max_f64(np.array([elapsed]))[0])
        group.barrier()

    check, probe = None, None
    if not args.no_check:
        check = exhaustive_check(wl, single_rank_sums=(world == 1))
        if group is not None:
            # cohort-wide sums: the oracle's per-shard sums added over the ranks must equal what the exchange produced
            c, td, dm, loc = check['sums']
            packed = group.allreduce_sum_i64(np.concatenate([c.reshape(-1), td, dm, loc]).astype(np.int64))
            got = np.concatenate([wl.call_out.sample_counters.get().reshape(-1), wl.call_out.sample_totaldp.get(),
                                  wl.call_out.sample_dp_missing.get(), wl.loc_counters.get()])
            assert np.array_equal(packed, got), "cohort-wide sums (all-reduce over the ranks) differ from the oracle's"
            check['loci_all_ranks'] = int(group.allreduce_sum_i64(np.array([check['loci']]))[0])
            # every rank's filter decisions arrived, in rank order == locus order
            gathered = wl.gather.get()
            assert np.array_equal(gathered[rank][:wl.n_loci], wl.bits.get()[:wl.n_loci]), \
                "all-gathered filter bits: own row differs"
            mine = np.zeros(gathered.shape[1], dtype=np.uint32)
            mine[:wl.n_loci] = wl.bits.get()[:wl.n_loci]
            rows_all = group.allgather_bytes(mine.view(np.uint8))
            for r in range(world):
                nr = locus_shard(args.loci, r, world)
                nr = (nr[1] - nr[0]) if args.scaling == 'strong' else args.loci
                assert np.array_equal(np.frombuffer(rows_all[r].tobytes(), dtype=np.uint32)[:nr], gathered[r][:nr]), \
                    "all-gathered filter bits: row of rank %d differs from what that rank computed" % r
    if rank == 0:
        cells = wl.n_loci * wl.n_real
        ms_step = elapsed / args.steps * 1e3
        loci_s = total_loci * args.steps / elapsed
        kn, kms = prof['k_call_filter']
        cn, cms = prof['k_locus_count']
        avg_cf = kms / max(kn, 1)
        avg_cnt = cms / max(cn, 1)
        achieved = cells * BYTES_PER_CELL_CALL_FILTER / (avg_cf * 1e-3) / 1e9 if kn else 0.0
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(tf) and wl.n_loci == 100000 and wl.n_real == 10000:
            try:
                pj = json.load(open(tf))
                traffic = pj.get('k_call_filter_bytes_per_launch')
                traffic_src = ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                               "bench command (%s), guide's gfx950 corrections applied; a recorded figure, NOT "
                               "measured in this run" % pj.get('source', 'see profiles/README.md'))
            except Exception:
                traffic = None
        out = {
            "metric": "loci/sec (and genotype-cells/sec) statSTR+dumpSTR, 100k loci x 10k samples",
            "value": loci_s, "unit": "loci/s",
            "cells_per_sec": loci_s * wl.n_real,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "i16", "data": "synthetic",
            "config": {"workload": "statSTR (11 stats) + dumpSTR (min-DP/max-DP/min-Q call filters, "
                                   "callrate/HWE/het-low/het-high locus filters) combined, HipSTR-shape, "
                                   "%d loci x %d samples (BASELINE configs[3]), %s" %
                                   (total_loci, wl.n_real,
                                    "one GPU" if world == 1 else
                                    ("locus-sharded over %d GPUs, %d loci per GPU" % (world, wl.n_loci))),
                       "n_loci_total": total_loci, "n_loci_per_gpu": wl.n_loci, "n_samples": wl.n_real, "ploidy": 2,
                       "row_layout": ("%d samples per device row: %d no-call padding samples so that every row starts "
                                      "on a 128-byte boundary (trk_batch.n_pad_samples; rates count the %d real samples)"
                                      % (wl.n_samples, wl.sb.n_pad, wl.n_real)) if wl.sb.n_pad else "dense",
                       "max_alleles": int(np.max(np.diff(wl.sb.tables[0]))),
                       "sharding": ("contiguous locus shards by rank; per step ONE grouped RCCL launch: all-reduce of "
                                    "the packed sample_info / totaldp / loc_info counters + all-gather of the "
                                    "per-locus filter decisions" if comm_mode != 'host' else
                                    "contiguous locus shards by rank; the step's exchange through HOST copies over the "
                                    "socket group (%s)" % (comm_note or "TRK_BENCH_COMM=host")) if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "k_call_filter_v4 (HIP events around that kernel alone; the 0.015 ms "
                                                   "k_cf_reduce that follows it is kernels_ms.k_cf_reduce)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "bytes_per_cell": BYTES_PER_CELL_CALL_FILTER, "avg_launch_ms": avg_cf,
                         "launches": kn, "box_stream_probe": None, "frac_of_box_stream": None,
                         # flat keys (the driver's parser drops nested objects of this one)
                         "placement_probe_ms": (wl.placement or {}).get('probe_ms'),
                         "placement_placed": (wl.placement or {}).get('placed'),
                         "placement_jumps": (wl.placement or {}).get('jumps'),
                         "placement_seconds": (wl.placement or {}).get('seconds'),
                         "placement_peak_extra_bytes": (wl.placement or {}).get('peak_extra_bytes'),
                         "placement_reserved": (wl.placement or {}).get('reserved'),
                         "reservation": type(eng).last_reservation,
                         "placement_note": "Engine() reserves the pair as the process's first two device allocations "
                                           "(trk_reserve_pair, the product default: 2 x 4 GiB; 'reservation' = what that "
                                           "saw) and trk_dev_alloc_pair lends it out (placement_reserved); without a "
                                           "reservation the call times the masked-genotype plane with each candidate "
                                           "mask plane (write-only probe, ms), at most two spare planes"},
            "kernels_ms": {k: (v[1] / v[0] if v[0] else None) for k, v in prof.items()},
            "k_locus_count_roofline": {"achieved": cells * BYTES_PER_CELL_COUNT / (avg_cnt * 1e-3) / 1e9 if cn else 0.0,
                                       "unit": "GB/s", "bytes_per_cell": BYTES_PER_CELL_COUNT,
                                       "frac": (cells * BYTES_PER_CELL_COUNT / (avg_cnt * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                       if cn else 0.0},
            "queues": ("3: stream kernels on queue 0, dumpSTR's finaliser / locus filters / RCCL exchange on queue 1, "
                       "statSTR's finaliser on queue 2, both overlapped with the call-filter kernel -- kernels_ms "
                       "are per-launch averages under that contention") if wl.overlap else "1",
            "parity_rows_checked": (check.get('loci_all_ranks', check['loci']) if check else 0),
            "device": eng.arch,
        }
        if check:
            out["parity"] = {"checker": "oracle/oracle_c.c on %d host threads (oracle/fullsize.py)" % check['threads'],
                             "loci": check['loci'], "calls": check['calls'],
                             "calls_bit_for_bit_gt_and_mask": check['calls_bit_for_bit'],
                             "worst_float_rel": check['worst_float_rel'], "seconds": check['seconds']}
    if world > 1 and not args.no_assoc:
        # BASELINE configs[4] ("associaTR ... 8 x MI355X"): every rank scans its shard, max over the ranks
        a = assoc_extra(wl, args.seed, args.no_check, True, group=group, total_loci=total_loci)
        if rank == 0:
            out.setdefault("extras", {})["associatr_scan"] = a
    if rank == 0 and world == 1:
        extras = out.setdefault("extras", {})
        if not args.no_assoc:
            extras["associatr_scan"] = assoc_extra(wl, args.seed, args.no_check, args.no_cpu_baseline)
        if not args.no_cpu_baseline:   # the CPU baselines are timed at N = 1 only
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds)
            try:
                extras["cpu_baseline_c"] = cpu_baseline_c(wl)
            except Exception as e:      # the checker's C half is optional equipment of the box
                extras["cpu_baseline_c"] = {"error": str(e)[:200]}
        if not args.no_extras:
            extras["compact_outputs"] = compact_outputs_extra(wl)
            extras["in_place_gt"] = in_place_extra(wl)
            extras["qc_reduce"] = qc_reduce_extra(wl, args.no_check)
            probe = box_stream_probe(eng, wl)
            wl.free()
            if not use_dist:
                uid = eng.comm_unique_id()
                eng.comm_init(0, 1, uid)
            extras["strong_shard"] = strong_shard_extra(eng, args, loci, ms_step, max(args.steps, 20))
            extras["config1"] = config1_extra(eng, args.no_check)
            extras["short_rows"] = short_rows_extra(eng)
            extras["config2"] = config2_extra(eng, args.no_check)
            extras["end_to_end"] = end_to_end_extra(eng, args.seed)
    if rank == 0:
        if probe is None:
            probe = box_stream_probe(eng, wl)
        out["roofline"]["box_stream_probe"] = probe
        if probe.get("achieved"):
            out["roofline"]["frac_of_box_stream"] = out["roofline"]["achieved"] / probe["achieved"]
            # flat copies (the driver's parser keeps scalars of this object, not nested ones)
            out["roofline"]["box_stream_ms"] = probe.get("avg_launch_ms")
            out["roofline"]["box_stream_frac_of_peak"] = probe.get("frac")
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if group is not None:
        group.barrier()
        group.close()
    eng.close()


if __name__ == '__main__':
    main()
