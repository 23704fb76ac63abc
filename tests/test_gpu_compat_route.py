"""The `compat/` route: a script that imports only the REFERENCE's module names (`sim_plain`, `logger`, `diffqc`,
`qutip`: demo_maxcut.py:2,4, sim_plain.py:1,10) runs on the MI355X engine when `compat/` is first on PYTHONPATH, and
reproduces the reference's own training trace (golden G5, captured from the reference's demo_maxcut.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n_epoch, tmp_path):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "compat"), ROOT])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "compat_user_script.py"), str(n_epoch), str(tmp_path)],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_reference_module_names_reproduce_two_g5_epochs(golden_dir, tmp_path):
    G = np.load(os.path.join(golden_dir, "reference_goldens.npz"))
    out = _run(2, tmp_path)
    for name, f in out["modules"].items():
        assert os.path.dirname(f).rstrip("/").endswith("compat") or "/compat/" in f, (name, f)
    assert out["diffqc_version"] == "dev"                      # diffqc.cc:227
    assert np.max(np.abs(np.array(out["losses"]) - G["g5_losses_energy"][:2])) < 1e-9
    assert len(out["prob"]) == 16 and abs(sum(out["prob"]) - 1.0) < 1e-12


@pytest.mark.gpu
def test_reference_module_names_reproduce_the_five_epoch_trace(golden_dir, tmp_path):
    G = np.load(os.path.join(golden_dir, "reference_goldens.npz"))
    out = _run(5, tmp_path)
    assert np.max(np.abs(np.array(out["losses"]) - G["g5_losses_energy"])) < 1e-9
    fin = np.array([complex(a, b) for a, b in out["final"]])
    assert np.max(np.abs(fin - G["g5_final_state"])) < 1e-9
    assert np.max(np.abs(np.array(out["prob"]) - G["g5_prob"])) < 1e-9


def test_compat_names_resolve_without_a_gpu(tmp_path):
    """CPU: the four module names import from compat/ and expose the reference's surface (diffqc.cc:210-228,
    sim_plain.py:14, logger.py:4); no compute call is made."""
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "compat"), ROOT])
    code = ("import sim_plain, logger, diffqc, qutip, sys\n"
            "assert all('/compat/' in m.__file__ for m in (sim_plain, logger, diffqc, qutip))\n"
            "assert callable(sim_plain.SimulatorPlain) and callable(logger.Logger)\n"
            "assert all(callable(getattr(diffqc, n)) for n in ('print_test', 'complex_test', 'test_eigen', 'set_H', 'trotter'))\n"
            "assert diffqc.__version__ == 'dev' and diffqc.complex_test([1 + 2j]) == [1 + 2j]\n"
            "assert diffqc.test_eigen([[1.0, 2.0]]) == [[1.0, 2.0]]\n"
            "q = qutip.Qobj([[0, 1], [1, 0]]); assert (q * qutip.qeye(2)).full()[0, 1] == 1\n")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
