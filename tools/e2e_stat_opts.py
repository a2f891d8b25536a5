"""statSTR's command line (tools/e2e_stat_only.py) with library options set first: e2e_stat_opts.py file KEY=VALUE ..."""
import os, sys
os.environ.setdefault('TRK_LAB', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trtools_amd import _lib as L
for kv in sys.argv[2:]:
    L.set_option(*kv.split('=', 1))
sys.argv = sys.argv[:2]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'e2e_stat_only.py')).read())
