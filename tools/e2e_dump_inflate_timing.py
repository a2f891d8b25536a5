"""dumpSTR's command line with the device inflate and the hook's per-run timing (library option TRK_INFLATE_TIMING)."""
import os, sys
os.environ.setdefault('TRK_LAB', '1')
os.environ['TRK_DEVICE_INFLATE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trtools_amd import _lib as L
L.set_option('TRK_INFLATE_TIMING', '1')
for kv in sys.argv[2:]:
    L.set_option(*kv.split('=', 1))
sys.argv = sys.argv[:2]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'e2e_dump_only.py')).read())
