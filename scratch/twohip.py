import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
x = torch.zeros(4, device="cuda")
print("torch ok", torch.version.hip)
from nutpie_amd import _lib
L = _lib.lib()
print("device_count", L.nphip_device_count())
import subprocess
maps = open("/proc/self/maps").read()
print(sorted(set(l.split()[-1] for l in maps.splitlines() if "amdhip" in l or "hsa-runtime" in l)))
hip = ctypes.CDLL("libamdhip64.so.7")
n = ctypes.c_int(-1)
rc = hip.hipGetDeviceCount(ctypes.byref(n))
hip.hipGetErrorString.restype = ctypes.c_char_p
print("rc", rc, hip.hipGetErrorString(rc), n.value)
