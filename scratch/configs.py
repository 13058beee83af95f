"""Timings of BASELINE.json configs 1, 3, 4 (the non-headline configs) on one MI355X."""
import sys, os, time, ctypes, subprocess, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nutpie_amd
from nutpie_amd.compile_pymc import from_raw_callback
def report(name, tr, el):
    n = int(tr.stats["n_steps"].sum()); T = tr.stats["n_steps"].shape[1]
    ticks = int(tr.stats["n_steps"].sum(1).max())
    print(f"{name}: {el:.2f}s wall, {n} leapfrogs -> {n/el/1e6:.3f} M leapfrogs/s; max leapfrogs/chain {ticks} -> {el/ticks*1e6:.1f} us per tick; div(post) {int(tr.stats['diverging'][:, T-1000:].sum())}")
# config 1: 10-dim std normal, 4 chains
t=time.time(); tr = nutpie_amd.sample(nutpie_amd.std_normal(10), chains=4, tune=400, draws=1000, seed=123, progress_bar=False, return_raw_trace=True); report("config1 stdnormal D=10 4 chains (fused)", tr, time.time()-t)
# config 4: eight schools, 256 chains, host callback
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "fixtures", "eight_schools.c")
so = "/tmp/libeight.so"; subprocess.run(["gcc","-O2","-fPIC","-shared","-ffp-contract=off","-o",so,src,"-lm"],check=True)
lib = ctypes.CDLL(so); addr = ctypes.cast(lib.eight_schools_logp, ctypes.c_void_p).value
for nt in (1, 4, 16):
    m = from_raw_callback(10, addr, name="theta", n_threads=nt, init="normal", keep_alive=lib)
    t=time.time(); tr = nutpie_amd.sample(m, chains=256, tune=400, draws=1000, seed=4, progress_bar=False, return_raw_trace=True); report(f"config4 eight-schools 256 chains host callback ({nt} threads)", tr, time.time()-t)
# config 3 with a NATIVE device log-density (tests/fixtures/radon_device.hip) instead of torch: what the callback path
# costs when the model side is one kernel per evaluation
import ctypes as C
from nutpie_amd import _lib
from nutpie_amd.radon import synthetic_radon_data
fix = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "fixtures")
_lib.lib()
rl = C.CDLL(os.path.join(fix, "libradon_device.so"))
rl.radon_device_create.restype = C.c_void_p; rl.radon_device_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
data = synthetic_radon_data(); n = int(data["county_idx"].max()) + 1
cty = np.ascontiguousarray(data["county_idx"], dtype=np.int32); fl = np.ascontiguousarray(data["floor"]); yy = np.ascontiguousarray(data["log_radon"])
h = rl.radon_device_create(n, len(yy), cty.ctypes.data, fl.ctypes.data, yy.ctypes.data)
fn = C.cast(rl.radon_device_logp, C.c_void_p).value
ref = None
for chains, gs in ((512, 0), (512, 16), (512, 64)):
    s = _lib.PyNutsSettings.Diag(1); s.update(num_tune=400, num_draws=1000, num_chains=chains, maxdepth=8)
    t = time.time(); smp = _lib.PySampler(s, _lib.NativeDeviceCallbackModel(2 * n + 3, fn, h, keep_alive=rl), graph_steps=gs); smp.wait(); el = time.time() - t
    tr = smp.take_results(); report(f"config3 radon {chains} chains, native HIP density (device callback, graph_steps={gs})", tr, el)
    if chains == 512:
        if ref is None: ref = tr.draws
        else: print("   identical to the ungraphed run:", np.array_equal(ref, tr.draws))
