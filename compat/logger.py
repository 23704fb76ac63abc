from diffquantum_amd.logger import Logger  # noqa: F401
