"""A slow and a fast pair of output planes; the mask plane moved SKEW bytes into its (larger) allocation: which skews
turn a slow pair fast, which a fast pair slow?  (The period and the width of the bad window of whatever the two
lock-step write streams collide on.)"""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys
os.environ['TRK_POOL_GB'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
eng = Engine(0)
L, S = 100000, 10016
SLACK = 256 << 20
class At:
    def __init__(self, ptr): self.ptr = ptr
ins = [eng.empty((L, S), np.uint32) for _ in range(3)]
for a in ins: a.zero()
planes = [eng.empty((L * S + SLACK // 4,), np.uint32) for _ in range(7)]
def probe(a, b, skew=0):
    return eng.stream_probe(ins[0], ins[1], ins[2], At(a.ptr), At(b.ptr + skew), L, S, reps=3)
pairs = [(i, j, probe(planes[i], planes[j])) for i in range(7) for j in range(i + 1, 7)]
lo = min(p[2] for p in pairs)
slow = [p for p in pairs if p[2] > 1.08 * lo]
fast = [p for p in pairs if p[2] <= 1.03 * lo]
print("pairs: %d fast (%.2f ms), %d slow: %s" % (len(fast), lo, len(slow), [(i, j, round(t, 2)) for i, j, t in slow]))
skews = [0] + [1 << k for k in range(7, 28)] + [3 << 20, 5 << 20, 6 << 20, 12 << 20, 24 << 20, 48 << 20, 96 << 20, 192 << 20]
for name, sel in (("slow", slow[:2]), ("fast", fast[:2])):
    for i, j, t in sel:
        print("%s pair (%d, %d):" % (name, i, j))
        print("   " + "  ".join("%s:%.2f" % ((("%dK" % (s >> 10)) if s < (1 << 20) else ("%dM" % (s >> 20))) if s >= 1024 else str(s),
                                               probe(planes[i], planes[j], s)) for s in sorted(skews)), flush=True)
