"""Shim: the name-keyed deterministic weights live in the package (seevcn_amd.seeding) so that bench.py does not depend on the test tree."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
from seevcn_amd.seeding import seeded_state_dict  # noqa: E402,F401
