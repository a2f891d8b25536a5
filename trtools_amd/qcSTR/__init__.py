"""qcSTR's data reductions on the device.  The plotting half of the reference's qcSTR
(trtools/qcSTR/qcSTR.py:42-340) is out of scope; ``reductions.qc_reductions`` produces the arrays it plots."""
from .reductions import qc_reductions  # noqa: F401
