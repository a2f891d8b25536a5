#!/usr/bin/env python3
"""trk_parse_samples on the 1 GB probe file (tools/e2e_probe.py): per batch of the native reader the batch's text goes
to the device, the kernel is timed with HIP events, its arrays are compared with the reader's for the whole batch.
usage: parse_probe.py /tmp/e2e/synth_17000x5000.vcf.gz [--planes DP Q] [--iters 5]"""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd import vcfnative
from test_gpu_parse import device_inputs

ap = argparse.ArgumentParser()
ap.add_argument('vcf')
ap.add_argument('--planes', nargs='*', default=['DP', 'Q'])
ap.add_argument('--iters', type=int, default=5)
a = ap.parse_args()
eng = Engine(0)
r = vcfnative.NativeVCFReader(a.vcf)
for k in a.planes:
    r.select_format(k)
kinds = ['f' if r.format_types[k][0] == 'Float' else 'i' for k in a.planes]
S = len(r.samples)
tot_ms = tot_bytes = tot_calls = tot_out = 0
flagged = nrec = 0
t_h2d = 0.0
while True:
    rb = r.read_raw_batch()
    if rb.n == 0:
        break
    text, so, le, gi, pidx = device_inputs(rb, a.planes)
    host = np.frombuffer(text, dtype=np.uint8)
    t = time.time()
    td = eng.empty((host.shape[0] + 32,), np.uint8)
    eng._chk(eng.lib.trk_memcpy_h2d(eng.ctx, td.ptr, host.ctypes.data, host.shape[0]))
    eng.sync()
    t_h2d += time.time() - t
    so_d, le_d, gi_d = eng.upload(so, np.int64), eng.upload(le, np.int64), eng.upload(gi, np.int8)
    pl = [(eng.upload(p, np.int8), k) for p, k in zip(pidx, kinds)]
    out = eng.parse_samples(td, so_d, le_d, S, rb.gt.shape[2], gi_d, planes=pl)      # warm-up + the result that is checked
    eng.sync()
    eng.timer_start(0)
    for _ in range(a.iters):
        o2 = eng.parse_samples(td, so_d, le_d, S, rb.gt.shape[2], gi_d, planes=pl)
        for x in [o2['gt'], o2['locus_ploidy'], o2['flags']] + o2['planes']:
            x.free()
    eng.timer_stop(0)
    ms = eng.timer_ms(0) / a.iters
    fl = out['flags'].get()
    take = fl == 0
    assert np.array_equal(out['gt'].get()[take], rb.gt[take])
    for k, arr in zip(a.planes, out['planes']):
        got, want = arr.get(), rb.planes[k][:, :, 0]
        assert np.array_equal(got[take].view(np.uint32), want[take].view(np.uint32)), k
    assert np.array_equal(out['locus_ploidy'].get()[take], rb.locus_ploidy[take])
    region = int((le - so).sum())
    outb = rb.n * S * (rb.gt.shape[2] * 2 + 4 * len(a.planes))
    print("batch of %d records: %.1f MB of sample text, kernel %.3f ms = %.0f GB/s of text (+ %.0f MB written: %.0f GB/s moved), "
          "%d flagged; every unflagged record equals the reader's arrays" % (rb.n, region / 1e6, ms, region / ms / 1e6, outb / 1e6,
                                                                            (region + outb) / ms / 1e6, int((~take).sum())), flush=True)
    tot_ms += ms; tot_bytes += region; tot_calls += rb.n * S; tot_out += outb
    flagged += int((~take).sum()); nrec += rb.n
    for x in [td, so_d, le_d, gi_d, out['gt'], out['locus_ploidy'], out['flags']] + out['planes'] + [p for p, _ in pl]:
        x.free()
print("TOTAL %d records, %d calls: kernel %.2f ms for %.0f MB of text = %.0f GB/s (%.0f GB/s with the arrays written; %.2e calls/s), "
      "%d records flagged; pageable host -> device copy of the text %.3f s = %.1f GB/s" % (
          nrec, tot_calls, tot_ms, tot_bytes / 1e6, tot_bytes / tot_ms / 1e6, (tot_bytes + tot_out) / tot_ms / 1e6,
          tot_calls / (tot_ms * 1e-3), flagged, t_h2d, tot_bytes / t_h2d / 1e9))
