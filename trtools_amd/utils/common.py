"""stderr messaging helpers (same surface as the reference's
trtools/utils/common.py:7-36)."""
import sys


def WARNING(msg):
    """Print a warning line on stderr."""
    sys.stderr.write(msg.strip() + "\n")


def MSG(msg, debug=False):
    """Print a status line on stderr, only when ``debug`` is set."""
    if debug:
        sys.stderr.write(msg.strip() + "\n")
