"""Stand-in for the `qutip` package, used ONLY by tests/golden/make_goldens.py to import the
reference's sim_plain.py / demo_maxcut.py in place (qutip is not installed in this image).
It re-exports the project's own numpy-backed Qobj; no reference code lives here."""
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..", "..")))
from diffquantum_amd.qobj import Qobj, qeye  # noqa: E402,F401


def mesolve(*a, **k):
    raise NotImplementedError("qutip.mesolve is outside the propagator hot path (SURVEY.md section 2, rows 5-6)")
