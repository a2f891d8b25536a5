"""Process-wide compute object (one Engine per process per GPU)."""
import os

_compute = None


def get_compute():
    """The DeviceCompute of this process; created on first use.  Raises (no CPU
    fallback) when libtrk.so is missing or no MI355X is visible."""
    global _compute
    if _compute is None:
        from .compute import DeviceCompute
        _compute = DeviceCompute(device=int(os.environ.get('TRK_DEVICE', os.environ.get('LOCAL_RANK', '0'))))
    return _compute


def set_compute(obj):
    """Install a compute object (tests inject an oracle-backed checker here)."""
    global _compute
    old = _compute
    _compute = obj
    return old
