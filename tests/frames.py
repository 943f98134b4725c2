"""Test-side access to the synthetic frame generator (kept in the package because bench.py uses it)."""
from rcvpose_amd.synthetic import LINEMOD_K, YCB_K, synthetic_frame  # noqa: F401
