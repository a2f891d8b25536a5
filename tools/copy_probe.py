import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
eng = Engine(0)
n = 1 << 30   # 4 GiB of int32
a = eng.empty((n,), np.int32); b = eng.empty((n,), np.int32)
a.zero(); eng.sync()
for _ in range(2): b.copy_from(a)
eng.sync()
eng.timer_start(0)
for _ in range(10): b.copy_from(a)
eng.timer_stop(0)
ms = eng.timer_ms(0) / 10
print("hipMemcpyAsync D2D 4 GiB: %.3f ms -> %.0f GB/s (read+write)" % (ms, 2 * a.nbytes / ms / 1e6))
eng.timer_start(1)
for _ in range(10): a.zero()
eng.timer_stop(1)
ms = eng.timer_ms(1) / 10
print("hipMemsetAsync 4 GiB: %.3f ms -> %.0f GB/s (write)" % (ms, a.nbytes / ms / 1e6))
