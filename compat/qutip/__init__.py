"""Fallback `qutip` namespace for machines without qutip: the Qobj subset the propagator path uses."""
from diffquantum_amd.qobj import Qobj, qeye  # noqa: F401


def mesolve(*a, **k):
    raise NotImplementedError("qutip.mesolve is not part of the Trotter hot path; install qutip for it")
