import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
# the suite forces code paths and compares them: the package's lab knobs are honoured (trtools_amd/_knobs.py); the
# library's own switches are set through helpers.lab_env / _lib.set_option (include/trk_test.h)
os.environ.setdefault('TRK_LAB', '1')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
