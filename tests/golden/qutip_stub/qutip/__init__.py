"""Stand-in for the `qutip` package, used ONLY by tests/golden/make_goldens.py to import the
reference's sim_plain.py / demo_maxcut.py in place (qutip is not installed in this image).

Self-contained on purpose: it imports nothing from `diffquantum_amd`, so the golden vectors do not
depend on the product's own Qobj (diffquantum_amd/qobj.py).  Only the operations the reference's
Trotter path performs on its operands are provided, each written directly
as the numpy expression of its textbook definition.  No reference code lives here."""
import numpy as np


class Qobj:
    __array_priority__ = 100.0

    def __init__(self, arr):
        m = np.array(arr.m if isinstance(arr, Qobj) else arr, dtype=complex)
        self.m = m.reshape(len(m), -1)            # vectors are kets (d, 1)

    shape = property(lambda s: s.m.shape)
    data = property(lambda s: s.m)

    def full(s):
        return np.array(s.m)

    def dag(s):
        return Qobj(np.conjugate(s.m).transpose())

    def __getitem__(s, i):
        return s.m[i]

    def __mul__(s, o):
        return Qobj(np.dot(s.m, o.m)) if isinstance(o, Qobj) else Qobj(s.m * complex(o))

    def __rmul__(s, o):
        return Qobj(complex(o) * s.m)

    def __truediv__(s, o):
        return Qobj(s.m / complex(o))

    def __add__(s, o):
        return Qobj(s.m + o.m)

    def __sub__(s, o):
        return Qobj(s.m - o.m)

    def matrix_element(s, bra, ket):                # <bra| s |ket>, bra given as a ket
        return complex(np.vdot(bra.m[:, 0], np.dot(s.m, ket.m[:, 0])))

    def norm(s):                                    # vectors: 2-norm; operators: trace norm
        return float(np.sqrt(np.vdot(s.m, s.m).real)) if 1 in s.m.shape else float(np.linalg.svd(s.m)[1].sum())

    def eigenenergies(s):
        return np.linalg.eigvalsh(s.m)

    def eigenstates(s):
        w, v = np.linalg.eigh(s.m)
        kets = np.empty(len(w), dtype=object)
        kets[:] = [Qobj(v[:, j]) for j in range(len(w))]
        return w, kets


def qeye(d):
    return Qobj(np.identity(int(d)))


def mesolve(*a, **k):
    raise NotImplementedError("qutip.mesolve is outside the propagator hot path (SURVEY.md section 2, rows 5-6)")
