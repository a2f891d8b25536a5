import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import sys; sys.path.insert(0,'/root/repo')
import bench, json
from trtools_amd.engine import Engine
eng = Engine(0)
r = bench.config1_extra(eng, False)
print(json.dumps({k: r[k] for k in r if k not in ("roofline", "workload")}))
