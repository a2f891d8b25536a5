"""Environment settings of the package.

SUPPORTED settings -- the table in README.md ("Environment") -- are read with ``env``:

    TRK_DEVICE            HIP device of this process (default: LOCAL_RANK, else 0)
    TRK_VCF_THREADS       inflate / parse threads of a native reader (libtrk; default: twice the CPU grant, 8 ... 64)
    TRK_FMT_THREADS       formatter threads of the native writers (libtrk; default: twice the CPU grant, 8 ... 32)
    TRK_VCF_READ_AHEAD    1 (default): batch n + 1 is read on a helper thread while batch n is worked on
    TRK_DEVICE_INFLATE    1: BGZF blocks are inflated on the GPU (the file crosses PCIe compressed); default: 1 for statSTR,
                          0 for dumpSTR (DEVICE_INFLATE_DEFAULT below)
    TRK_DEVICE_PARSE      1 (default): the sample columns are parsed on the GPU
    TRK_DEVICE_FORMAT     1 (default): dumpSTR's sample columns are written on the GPU
    TRK_PLACE_OUTPUTS     1 (default): big output plane pairs of the call-filter pass are placed (trk_dev_alloc_pair)
    TRK_RESERVE_PAIR_GB   GiB per plane an Engine reserves for that pair at start-up (default 4 on >= 64 GB devices
                          for API engines, 0 for the command lines)
    TRK_POOL_GB           device memory the engine's buffer pool may hold (default 8)
    TRK_ZIP_LEVEL         level of dumpSTR --zip's host compressor (libdeflate 1 ... 9, 0: stored; default 6)
    TRK_DEVICE_DEFLATE    1: dumpSTR --zip's blocks are deflated on the GPU (trk_deflate_bgzf); default 0

libtrk itself reads three of them with getenv -- TRK_VCF_THREADS, TRK_FMT_THREADS and TRK_VCF_BUF_CACHE_MB (megabytes of
text buffers a process keeps for its next reader, default 4096, 0: none; read once, when the first reader opens) -- and
nothing else.

Everything else the package ever looked up in the environment -- forced code paths of the parity tests, A/B switches of
tools/ -- is a LAB knob: read with ``lab`` and honoured only when TRK_LAB=1 is set (tests/conftest.py and the tools
set it).  The library's own switches are options of include/trk_test.h (``_lib.set_option``); its lab build
(`make -C trtools_amd/csrc lab`) reads them from the environment too.
"""
import os

SUPPORTED = ('TRK_DEVICE', 'TRK_VCF_THREADS', 'TRK_FMT_THREADS', 'TRK_VCF_READ_AHEAD', 'TRK_DEVICE_INFLATE',
             'TRK_DEVICE_PARSE', 'TRK_DEVICE_FORMAT', 'TRK_PLACE_OUTPUTS', 'TRK_RESERVE_PAIR_GB', 'TRK_POOL_GB',
             'TRK_ZIP_LEVEL', 'TRK_DEVICE_DEFLATE')


# The command lines' defaults for TRK_DEVICE_INFLATE (profiles/r05_notes.md section 6, r05_e2e_cpu_seconds.txt; 1.02 GB):
#   statSTR  '1': 0.076-0.088 s against 0.095-0.105 with the host's inflater threads on the same box, at 0.5 CPU-seconds
#                 instead of 1.6-1.7 (two runs of 4096 members in flight behind the reader, kernels on a lowest-priority queue);
#   dumpSTR  '0': its own kernels and copies share the device with the inflate kernel -- 0.22-0.24 s against 0.173,
#                 although at 0.76 CPU-seconds instead of 1.8-2.0: TRK_DEVICE_INFLATE=1 where host cores are the scarce thing.
DEVICE_INFLATE_DEFAULT = {'statSTR': '1', 'dumpSTR': '0'}


def env(name, default=None):
    """A supported setting."""
    assert name in SUPPORTED, name
    return os.environ.get(name, default)


def lab(name, default=None):
    """A lab knob: ``default`` unless the process runs with TRK_LAB=1."""
    if os.environ.get('TRK_LAB') != '1':
        return default
    return os.environ.get(name, default)
