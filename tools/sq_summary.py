#!/usr/bin/env python3
"""Reads the text tools/sq_counters.sh wrote and prints, per kernel, the instruction mix relative to the wave-cycles
(SQ_WAVE_CYCLES counts 4-clock quads summed over waves; with W waves resident per SIMD a pipe that is busy all the
time shows 1/W)."""
import collections, re, sys
d = collections.defaultdict(dict)
for ln in open(sys.argv[1]):
    m = re.match(r'(\S+)\s+(SQ_\w+|GRBM_\w+)\s+avg\s+(\S+)\s+n\s+(\d+)', ln)
    if m:
        d[m.group(1)][m.group(2)] = float(m.group(3))
want = sys.argv[2:] or None
for k, v in d.items():
    if 'SQ_WAVE_CYCLES' not in v or (want and not any(w in k for w in want)):
        continue
    wc, w = v['SQ_WAVE_CYCLES'], v['SQ_WAVES']
    print("%s\n   waves %d  wave_cycles/wave %.0f" % (k, w, wc / w))
    for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_SMEM',
              'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_ANY',
              'SQ_WAIT_ANY', 'SQ_WAIT_INST_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE'):
        if c in v:
            print("   %-22s %.4g   per wave-cycle %.3f" % (c, v[c], v[c] / wc))
