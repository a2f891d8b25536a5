#!/usr/bin/env python3
"""trk_inflate_blocks on every BGZF member of a file: members per launch, ms per launch, GB/s of text, against zlib on one
host thread.  usage: inflate_probe.py file.vcf.gz [members per launch ...]"""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine

path = sys.argv[1]
sizes = [int(x) for x in sys.argv[2:]] or [256, 1024, 4096, 0]
raw = open(path, 'rb').read()
mem, p = [], 0
while p + 18 <= len(raw):
    xlen = raw[p + 10] | (raw[p + 11] << 8)
    bsize = (raw[p + 16] | (raw[p + 17] << 8)) + 1
    isize = struct.unpack('<I', raw[p + bsize - 4:p + bsize])[0]
    mem.append((p + 12 + xlen, bsize - 12 - xlen - 8, isize))
    p += bsize
n_text = sum(m[2] for m in mem)
print("%s: %d members, %.1f MB compressed, %.1f MB of text" % (path, len(mem), len(raw) / 1e6, n_text / 1e6), flush=True)
t = time.perf_counter()
k = min(len(mem), 400)
for o, n, isz in mem[:k]:
    zlib.decompress(raw[o:o + n], -15)
dt = time.perf_counter() - t
print("zlib, one thread: %.1f MB/s of text" % (sum(m[2] for m in mem[:k]) / dt / 1e6), flush=True)
eng = Engine(0, reserve_pair_gb=0)
comp = eng.upload(np.frombuffer(raw + bytes(64), np.uint8), np.uint8)
for per in sizes:
    per = per or len(mem)
    ms_tot, n_launch = 0.0, 0
    flagged = 0
    for rep in range(2):
        for a in range(0, len(mem), per):
            part = mem[a:a + per]
            in_off = np.array([m[0] for m in part], np.int64)
            in_len = np.array([m[1] for m in part], np.int32)
            out_len = np.array([m[2] for m in part], np.int32)
            out_off = np.zeros(len(part), np.int64)
            out_off[1:] = np.cumsum(out_len[:-1])
            eng.sync()
            eng.timer_start(0)
            text, fl = eng.inflate_blocks(comp, in_off, in_len, out_off, out_len)
            eng.timer_stop(0)
            if rep:
                ms_tot += eng.timer_ms(0)
                n_launch += 1
                flagged += int((fl != 0).sum())
            text.free()
    print("%6d members per launch: %8.3f ms per launch, %7.1f GB/s of text (whole file %.1f ms), %d flagged"
          % (per, ms_tot / n_launch, n_text / (ms_tot * 1e-3) / 1e9, ms_tot, flagged), flush=True)
eng.close()
