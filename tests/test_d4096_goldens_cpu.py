"""tests/golden/oracle_d4096.npz (the oracle's answers at the headline dimension, tests/d4096_cases.py): present, complete, and
made for the inputs the GPU tests build -- checked here for all three cases (a few seconds and up to 3.5 GB of host memory each)."""
import numpy as np
import pytest

import d4096_cases


def test_fixture_is_complete():
    z = np.load(d4096_cases.GOLDEN)
    for name in d4096_cases.CASES:
        for key in ("signature", "loss", "grad", "fin", "energies"):
            assert name + "/" + key in z.files, (name, key)
        assert z[name + "/fin"].shape == (4096, d4096_cases.FIN_COLS) and z[name + "/fin"].dtype == np.complex128
        assert np.all(np.isfinite(z[name + "/grad"])) and np.max(np.abs(z[name + "/grad"])) > 1e-4
        assert abs(float(z[name + "/loss"]) - float(np.sum(z[name + "/energies"]))) < 1e-9


@pytest.mark.parametrize("case", list(d4096_cases.CASES))
def test_fixture_was_made_for_the_inputs_the_tests_build(case):
    P, a = d4096_cases.answer(case, oracle=None)      # (oracle=None: a signature mismatch would fail loudly instead of recomputing)
    assert a["stored"] and a["grad"].shape == P["u"].shape and a["energies"].shape == (P["B"],)
