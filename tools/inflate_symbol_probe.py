#!/usr/bin/env python3
"""What a SYMBOL costs k_inflate_bgzf: the same text as BGZF members of three makes -- zlib's default (what bgzip writes),
Z_HUFFMAN_ONLY (every symbol a literal: 65 280 per member) and Z_RLE (literals and distance-1 matches) -- a launch of
1024 members each (four per CU: one member's latency, nothing shares a SIMD), milliseconds per member.
usage: inflate_symbol_probe.py file.vcf.gz"""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')
import os, struct, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
import gzip

N = 1024
text = b''
with gzip.open(sys.argv[1], 'rb') as fh:      # (BGZF is a chain of gzip members)
    text = fh.read(N * 65280 + 200000)
text = text[-N * 65280:]                       # (past the header: sample columns)
eng = Engine(0, reserve_pair_gb=0)


for name, strategy, level in (("default level 6", zlib.Z_DEFAULT_STRATEGY, 6), ("default level 1", zlib.Z_DEFAULT_STRATEGY, 1),
                              ("huffman only", zlib.Z_HUFFMAN_ONLY, 6), ("rle", zlib.Z_RLE, 6), ("stored", zlib.Z_DEFAULT_STRATEGY, 0)):
    parts, in_off, in_len, out_len = [], [], [], []
    at = 0
    for k in range(N):
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        d = c.compress(text[k * 65280:(k + 1) * 65280]) + c.flush()
        parts.append(d)
        in_off.append(at)
        in_len.append(len(d))
        out_len.append(65280)
        at += len(d)
    raw = b''.join(parts)
    comp = eng.upload(np.frombuffer(raw + bytes(64), np.uint8), np.uint8)
    in_off, in_len, out_len = np.array(in_off, np.int64), np.array(in_len, np.int32), np.array(out_len, np.int32)
    out_off = np.arange(N, dtype=np.int64) * 65280
    ms = []
    for rep in range(3):
        eng.sync()
        eng.timer_start(0)
        t, fl = eng.inflate_blocks(comp, in_off, in_len, out_off, out_len)
        eng.timer_stop(0)
        ms.append(eng.timer_ms(0))
        if rep == 0:
            got = bytes(t.get()[:N * 65280])
            assert got == text and not fl.any(), name
        t.free()
    comp.free()
    print("%-16s %6.1f KB per member compressed, %7.3f ms per member (best of 3: %s)"
          % (name, len(raw) / N / 1e3, min(ms), ' '.join('%.3f' % x for x in ms)), flush=True)
eng.close()
