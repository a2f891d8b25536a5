"""trk_deflate_bgzf on a block of dumpSTR-like text: seconds per call (upload, kernels, download, CRCs), the ratio, and
libtrk's host compressor (trk_bgzf_compress, libdeflate level 6 / 1) on the same bytes."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import ctypes as C, os, sys, time, zlib, gzip
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd import bgzf
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 150
from test_vcfnative_hook import _synthetic
base = _synthetic(400, 5000, seed=1)
text = (base * (mb * 1000000 // len(base) + 1))[:mb * 1000000]
eng = Engine(0, reserve_pair_gb=0)
for rep in range(4):
    t = time.time(); out = eng.deflate_bgzf(text); dt = time.time() - t
    print("device: %d MB in %.1f ms = %.1f GB/s  ratio %.3f" % (mb, dt * 1e3, len(text) / dt / 1e9, len(out) / len(text)), flush=True)
assert gzip.decompress(bytes(out)) == text
# from a pinned buffer (what the record writer's output block is)
pin = eng.host_buffer(len(text))
pin[:len(text)] = np.frombuffer(text, dtype=np.uint8)
for rep in range(3):
    t = time.time(); out = eng.deflate_bgzf(None, address=pin.ctypes.data, nbytes=len(text)); dt = time.time() - t
    print("device, pinned text: %.1f ms = %.1f GB/s" % (dt * 1e3, len(text) / dt / 1e9), flush=True)
lib = bgzf._native_lib()
buf = bytearray(lib.trk_bgzf_bound(len(text)))
dst = (C.c_char * len(buf)).from_buffer(buf)
for lvl in (6, 1):
    for rep in range(2):
        got = C.c_size_t(); t = time.time()
        lib.trk_bgzf_compress(C.c_char_p(text), len(text), lvl, 0, dst, len(buf), C.byref(got)); dt = time.time() - t
    print("host libdeflate level %d: %.1f ms = %.2f GB/s  ratio %.3f" % (lvl, dt * 1e3, len(text) / dt / 1e9, got.value / len(text)), flush=True)
