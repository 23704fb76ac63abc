"""Minimal numpy-backed quantum object: the subset of ``qutip.Qobj`` that the
reference's simulator API consumes.

The reference's ``SimulatorPlain`` (/root/reference/sim_plain.py:119-153, 156-231,
245-305, 494-505) and its MaxCut demo (/root/reference/demo_maxcut.py:64-88) take
and return ``qutip.Qobj`` operands and use only: ``.full()``, ``.dag()``, ``*``,
``+``, ``-``, ``/``, ``.shape``, ``.data``, ``[i]``, ``.matrix_element(bra, ket)``,
``.eigenenergies()``, ``.eigenstates()``, ``.norm()`` and ``qeye(d)``.  qutip is not
installed on the MI355X image, so the host-side mirror of that API carries its own
dense container with exactly those operations.  It is host plumbing, not compute:
the propagator hot path never runs through this file.
"""
from __future__ import annotations

import numpy as np

__all__ = ["Qobj", "qeye", "as_array"]


def as_array(x) -> np.ndarray:
    """Dense complex128 2-D array of a Qobj / array-like (kets become (d, 1))."""
    if isinstance(x, Qobj):
        return x._a
    a = np.asarray(x)
    if a.ndim == 0:
        a = a.reshape(1, 1)
    elif a.ndim == 1:
        a = a.reshape(-1, 1)
    return np.ascontiguousarray(a, dtype=np.complex128)


class Qobj:
    __array_priority__ = 1000  # make ndarray * Qobj dispatch to Qobj.__rmul__

    def __init__(self, data=None, dims=None, copy=True):
        a = as_array(data if data is not None else [[0.0]])
        self._a = a.copy() if copy else a
        self.dims = dims

    # --- containers -----------------------------------------------------
    @property
    def shape(self):
        return self._a.shape

    @property
    def data(self):
        return self._a

    @property
    def isket(self):
        return self._a.shape[1] == 1 and self._a.shape[0] >= 1

    def full(self):
        return self._a.copy()

    def __getitem__(self, idx):
        return self._a[idx]

    def __repr__(self):
        return "Qobj(shape={}, data=\n{})".format(self.shape, self._a)

    # --- algebra --------------------------------------------------------
    def dag(self):
        return Qobj(self._a.conj().T, copy=True)

    def conj(self):
        return Qobj(self._a.conj())

    def trans(self):
        return Qobj(self._a.T)

    def __mul__(self, other):
        if isinstance(other, Qobj):
            return Qobj(self._a @ other._a, copy=False)
        if np.isscalar(other):
            return Qobj(self._a * other, copy=False)
        return NotImplemented

    def __rmul__(self, other):
        if np.isscalar(other):
            return Qobj(other * self._a, copy=False)
        return NotImplemented

    def __matmul__(self, other):
        return self.__mul__(other)

    def __truediv__(self, other):
        if np.isscalar(other):
            return Qobj(self._a / other, copy=False)
        return NotImplemented

    def __add__(self, other):
        if isinstance(other, Qobj):
            return Qobj(self._a + other._a, copy=False)
        if np.isscalar(other):
            if other == 0:
                return Qobj(self._a)
            if self._a.shape[0] == self._a.shape[1]:
                return Qobj(self._a + other * np.eye(self._a.shape[0]), copy=False)
        return NotImplemented

    __radd__ = __add__

    def __sub__(self, other):
        if isinstance(other, Qobj):
            return Qobj(self._a - other._a, copy=False)
        return self.__add__(-other)

    def __rsub__(self, other):
        return (-self).__add__(other)

    def __neg__(self):
        return Qobj(-self._a, copy=False)

    def __eq__(self, other):
        return isinstance(other, Qobj) and self.shape == other.shape and bool(np.all(self._a == other._a))

    __hash__ = None

    # --- measurements ---------------------------------------------------
    def matrix_element(self, bra, ket):
        """``<bra| self |ket>``; a ket passed as ``bra`` is conjugated, as qutip does."""
        b = as_array(bra)
        k = as_array(ket)
        if b.shape[1] == 1 and b.shape[0] == self._a.shape[0]:
            b = b.conj().T
        if k.shape[0] == 1 and k.shape[1] == self._a.shape[1] and self._a.shape[1] != 1:
            k = k.conj().T
        return complex((b @ (self._a @ k)).reshape(-1)[0])

    def norm(self):
        a = self._a
        if a.shape[0] == 1 or a.shape[1] == 1:
            return float(np.linalg.norm(a.reshape(-1)))
        return float(np.sum(np.linalg.svd(a, compute_uv=False)))  # trace norm

    def unit(self):
        return self / self.norm()

    def tr(self):
        return complex(np.trace(self._a))

    def _is_herm(self):
        return self._a.shape[0] == self._a.shape[1] and np.allclose(self._a, self._a.conj().T, atol=1e-12)

    def eigenenergies(self):
        if self._is_herm():
            return np.linalg.eigvalsh(self._a)
        w = np.linalg.eigvals(self._a)
        return w[np.argsort(w.real, kind="stable")]

    def eigenstates(self):
        if self._is_herm():
            w, v = np.linalg.eigh(self._a)
        else:
            w, v = np.linalg.eig(self._a)
            order = np.argsort(w.real, kind="stable")
            w, v = w[order], v[:, order]
        kets = np.empty(len(w), dtype=object)
        for j in range(len(w)):
            kets[j] = Qobj(v[:, j].reshape(-1, 1))
        return w, kets


def qeye(d):
    return Qobj(np.eye(int(d), dtype=np.complex128), copy=False)
