from diffquantum_amd.diffqc import *  # noqa: F401,F403
from diffquantum_amd.diffqc import __version__, set_H, trotter, trotter_batch, print_test, complex_test, test_eigen  # noqa: F401
