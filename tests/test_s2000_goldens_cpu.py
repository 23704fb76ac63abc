"""tests/golden/oracle_s2000.npz (the oracle's answers over the full 2000-step chain, tests/s2000_cases.py): present, complete, and
made for the inputs the GPU test builds (signature of every case)."""
import numpy as np
import pytest

import s2000_cases


def test_fixture_is_complete():
    z = np.load(s2000_cases.GOLDEN)
    for name, make in s2000_cases.CASES.items():
        for key in ("signature", "loss", "grad", "fin", "energies"):
            assert name + "/" + key in z.files, (name, key)
        g = z[name + "/grad"]
        assert g.shape[0] == s2000_cases.S and np.all(np.isfinite(g)) and np.max(np.abs(g)) > 1e-4
        assert z[name + "/fin"].shape[1] == s2000_cases.B and z[name + "/fin"].dtype == np.complex128
        assert abs(float(z[name + "/loss"]) - float(np.sum(z[name + "/energies"]))) < 1e-10
        assert np.max(np.abs(np.sum(np.abs(z[name + "/fin"]) ** 2, axis=0) - 1.0)) < 1e-10      # the oracle's chain is unitary


@pytest.mark.parametrize("case", list(s2000_cases.CASES))
def test_fixture_was_made_for_the_inputs_the_tests_build(case):
    P, a = s2000_cases.answer(case)        # raises when the stored signature differs
    assert a["grad"].shape == P["u"].shape == (2000, len(P["Hs"])) and a["energies"].shape == (P["B"],)
