"""diffquantum_amd -- MI355X-native propagator engine behind the diffquantum Python API.

Only the Schroedinger-propagator hot path of YilingQiao/diffquantum lives here
(SURVEY.md section 8): the HIP library (csrc/, C-ABI in include/diffqc_hip.h), its ctypes
handle (engine.py) and the host-side mirrors of the reference interfaces for that path
(diffqc.py ~ the `diffqc` pybind module, sim_plain.py ~ `SimulatorPlain`, logger.py, qobj.py).
"""
__version__ = "0.1"
