"""The compiled `diffqc` module (diffquantum_amd/_diffqc_native*.so, csrc/diffqc_module.cpp) without a GPU: it loads, exposes
the reference module's surface (diffqc.cc:210-228), and its C++ pulse model f_u (diffqc.cc:95-135) equals the Python twin's
(diffquantum_amd/pulse.py) and the oracle's scalar restatement.  No compute call is made."""
import numpy as np
import pytest

from diffquantum_amd import pulse


@pytest.fixture(scope="module")
def native():
    from diffquantum_amd import _diffqc_native
    return _diffqc_native


def test_surface(native, capsys):
    for name in ("print_test", "complex_test", "test_eigen", "set_H", "trotter", "trotter_batch", "f_u_table"):
        assert callable(getattr(native, name)), name
    assert native.__version__ == "dev" and native.engine_version().startswith("diffqc_hip")
    native.print_test()
    assert capsys.readouterr().out == "hello\n"
    assert native.complex_test([1 + 2j, 3]) == [1 + 2j, 3 + 0j]
    assert native.test_eigen([[1, 2], [3, 4]]) == [[1.0, 2.0], [3.0, 4.0]]
    with pytest.raises(RuntimeError):
        native.trotter([1, 0], 0.0, 1.0, 10, np.zeros((2, 1, 6)).tolist())      # set_H not called


@pytest.mark.parametrize("func_type", [0, 1])
def test_pulse_model_equals_python_twin_and_oracle(native, oracle, func_type):
    rng = np.random.default_rng(func_type + 5)
    vv = rng.normal(0, 1.2, (2, 4, 6))
    vv[:, 3, :] = 0.0                                   # N = 0: inside the 1e-6 cutoff
    channels = [[[0.0, 2.0, 1.5, 0.0], [0.0, 0.8, -2.5, 1.0]], [[0.0, 3.0, 0.0, 2.0]], [[0.0, 1.0, 0.7, 3.0]], []]
    ts = np.concatenate([np.linspace(0.0, 1.7, 41), [0.0, 0.85, 1.7, -0.1, 2.0]])
    got = native.f_u_table(vv.tolist(), ts.tolist(), channels, 1.7, func_type)
    twin = pulse.f_u_table(vv, channels, 1.7, func_type, ts)
    assert got.shape == (len(ts), 4) and np.max(np.abs(got - twin)) < 1e-14
    ref = np.array([[oracle.f_u(h, float(t), vv, channels, 1.7, func_type) for h in range(4)] for t in ts])
    assert np.max(np.abs(got - ref)) < 1e-13
    assert np.all(got[:, 2] == 0.0) and np.all(got[:, 3] == 0.0)      # zero amplitudes / a control without channels
    with pytest.raises(ValueError):
        native.f_u_table(vv[0].tolist() and [vv[0].tolist()], ts.tolist(), channels, 1.7, func_type)   # vv not [2][..][..]
    with pytest.raises(ValueError):
        native.f_u_table(vv.tolist(), ts.tolist(), [[[0.0, 1.0, 0.0, 9.0]]], 1.7, func_type)           # idx out of range
