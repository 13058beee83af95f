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
fix = ctypes.CDLL(os.path.join(ROOT, "tests", "fixtures", "libscaled_normal_device.so"))
fn = ctypes.cast(fix.scaled_normal_device_fast, ctypes.c_void_p).value
hip.lib()
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
          f"{smp.launches} launches, {secs / smp.launches * 1e6:.1f} us per step (engine kernel + callback kernel + gaps), mean depth {smp._copy('depth', np.int64).mean():.2f}", flush=True)
    smp.close()
