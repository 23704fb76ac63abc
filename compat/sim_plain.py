from diffquantum_amd.sim_plain import SimulatorPlain  # noqa: F401
