"""The launch-per-evaluation kernels behind a device callback that costs almost nothing (tests/fixtures/scaled_normal_device.hip,
the wave-parallel form): what one evaluation step costs on the engine's side.
usage: python scratch/cbtime.py dim chains [waves [graph_steps [tune draws]]]"""
import sys, os, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nutpie_amd import _lib as hip

dim, chains = int(sys.argv[1]), int(sys.argv[2])
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 0
graph = int(sys.argv[4]) if len(sys.argv) > 4 else 0
tune, draws = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (60, 20)
hip.lib()   # first: it loads torch's HIP runtime before any other library can pull in a second one
fix = ctypes.CDLL(os.path.join(ROOT, "tests", "fixtures", "libscaled_normal_device.so"))
fn = ctypes.cast(fix.scaled_normal_device_fast, ctypes.c_void_p).value
for rep in range(2):
    s = hip.PyNutsSettings.Diag(11)
    s.update(num_tune=tune, num_draws=draws, num_chains=chains)
    m = hip.NativeDeviceCallbackModel(dim, fn, 0, keep_alive=fix)
    t0 = time.perf_counter()
    smp = hip.PySampler(s, m, waves_per_chain=waves, graph_steps=graph, store_draws=False)
    smp.wait()
    secs = smp.seconds
    n = smp._copy("n_steps", np.int64)
    print(f"scaled normal D={dim} chains={chains} waves={smp.waves_per_chain} graph_steps={graph}: {n.sum() / secs / 1e6:.2f} M leapfrogs/s, job {secs:.3f} s, "
          f"{smp.launches} launches, {secs / n.sum(1).max() * 1e6:.1f} us per evaluation step of the longest chain (engine kernel + callback kernel + gaps; "
          f"{n.sum(1).max()} evaluations), mean depth {smp._copy('depth', np.int64).mean():.2f}", flush=True)
    out = (ctypes.c_int64 * 16)(); hip.lib().nphip_sampler_profile(smp._h, out); o = list(out)
    if o[3]:
        print(f"   cycles: leaf pass (leaf_cb / lf2) {o[0] / o[3]:.0f} per leaf; whole leaf, no rare path {o[1] / max(o[4], 1):.0f} (n={o[4]}); leaf that ends in a rare path "
              f"{o[2] / max(o[5], 1):.0f} (n={o[5]}): adapt / position pass {o[13] / max(o[5], 1):.0f}, begin_draw {o[14] / max(o[5], 1):.0f}; "
              f"whole launch per chain {o[15] / max(smp.launches, 1) / chains:.0f} (sum over {smp.launches} launches)")
        print(f"   draw end (rare_end_draw, whole) {o[7] / max(o[5], 1):.0f}: position pass alone {o[10] / max(o[5], 1):.0f}, momentum refresh {o[8] / max(o[5], 1):.0f}")
    smp.close()
