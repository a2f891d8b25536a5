"""Process-wide compute object (one Engine per process per GPU)."""
from . import _knobs
import os

_compute = None


def get_compute():
    """The DeviceCompute of this process; created on first use.  Raises (no CPU
    fallback) when libtrk.so is missing or no MI355X is visible."""
    global _compute
    if _compute is None:
        from .compute import DeviceCompute
        # (the command lines and the TRRecord API work batch by batch: output planes far below the size at which their
        # placement shows -- no reserved pair, unless TRK_RESERVE_PAIR_GB asks for one)
        env = _knobs.env('TRK_RESERVE_PAIR_GB')
        _compute = DeviceCompute(device=int(_knobs.env('TRK_DEVICE', os.environ.get('LOCAL_RANK', '0'))),
                                 reserve_pair_gb=float(env) if env else 0.0)
    return _compute


def set_compute(obj):
    """Install a compute object (tests inject an oracle-backed checker here)."""
    global _compute
    old = _compute
    _compute = obj
    return old
