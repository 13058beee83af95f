import sys, os, time, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import _lib
from nutpie_amd.radon import synthetic_radon_data
_lib.lib()
fix = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "fixtures")
rl = C.CDLL(os.path.join(fix, "libradon_device.so"))
rl.radon_device_create.restype = C.c_void_p; rl.radon_device_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
data = synthetic_radon_data(); n = int(data["county_idx"].max()) + 1
cty = np.ascontiguousarray(data["county_idx"], dtype=np.int32); fl = np.ascontiguousarray(data["floor"]); yy = np.ascontiguousarray(data["log_radon"])
h = rl.radon_device_create(n, len(yy), cty.ctypes.data, fl.ctypes.data, yy.ctypes.data)
fn = C.cast(rl.radon_device_logp, C.c_void_p).value
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 512
s = _lib.PyNutsSettings.Diag(1); s.update(num_tune=200, num_draws=200, num_chains=chains)
t = time.time(); smp = _lib.PySampler(s, _lib.NativeDeviceCallbackModel(2 * n + 3, fn, h, keep_alive=rl)); smp.wait(); el = time.time() - t
tr = smp.take_results(); ticks = int(tr.stats["n_steps"].sum(1).max())
print(f"native radon {chains} chains: {el:.2f} s, {tr.stats['n_steps'].sum()/el/1e6:.2f} M leapfrogs/s, {el/ticks*1e6:.1f} us per tick")
