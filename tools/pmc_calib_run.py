"""Runs tools/libpmc_calib.so under rocprofv3 --pmc: 1 GiB read, then 1 GiB written, 8 bytes per lane."""
import ctypes, os
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpmc_calib.so"))
L.pmc_calib.argtypes = [ctypes.c_size_t]
print("calib rc", L.pmc_calib(1 << 27))
