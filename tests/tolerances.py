"""Shim: the comparison lives in oracle/tolerances.py (test infrastructure that smoke() may import too)."""
from oracle.tolerances import assert_close_per_channel  # noqa: F401
