"""Shared test helpers (JSON golden decoding, float comparison)."""
import json
import math
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def unjf(x):
    if x is None:
        return None
    if isinstance(x, str):
        return {'nan': math.nan, 'inf': math.inf, '-inf': -math.inf}[x]
    return float(x)


def close(a, b, rtol=1e-9, atol=1e-12):
    """Float parity bar of BASELINE.json north_star: within 1e-9."""
    a, b = float(a), float(b)
    if math.isnan(a) or math.isnan(b):
        return math.isnan(a) and math.isnan(b)
    if math.isinf(a) or math.isinf(b):
        return a == b
    return abs(a - b) <= atol + rtol * max(abs(a), abs(b))


class lab_env:
    """``with lab_env(TRK_FMT_FAST='0', ...):`` -- switches for the block, in BOTH places a switch can live: the
    package's lab knobs (environment, honoured under TRK_LAB=1: tests/conftest.py) and the library's options
    (include/trk_test.h: trk_test_set_option).  Restored afterwards."""

    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        from trtools_amd import _lib as L
        self.old_env = {k: os.environ.get(k) for k in self.kv}
        self.opt = L.options(**self.kv)
        self.opt.__enter__()
        os.environ.update(self.kv)
        return self

    def __exit__(self, *exc):
        self.opt.__exit__(*exc)
        for k, v in self.old_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        return False
