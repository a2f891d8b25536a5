import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys, time
sys.path.insert(0, '/root/repo')
import bench
from trtools_amd.engine import Engine
from trtools_amd.synth import make_loci
eng = Engine(0)
uid = eng.comm_unique_id(); eng.comm_init(0, 1, uid)
loci = make_loci(100000, 10000, 20260931)
for n in (12500, 25000):
    w = bench.Workload(eng, 20260931, 10000, loci.slice(0, n), 0, 1, use_comm=True, pipeline_count=True, gather_loci=n)
    for _ in range(5): w.step()
    w.flush(); eng.sync()
    K = 40
    t0 = time.perf_counter()
    for _ in range(K): w.step()
    t1 = time.perf_counter()
    w.flush(); eng.sync()
    t2 = time.perf_counter()
    print("%d loci: host enqueue %.3f ms/step, total %.3f ms/step" % (n, (t1 - t0) / K * 1e3, (t2 - t0) / K * 1e3), flush=True)
    w.free()
