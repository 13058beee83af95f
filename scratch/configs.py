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
