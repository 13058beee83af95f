"""Round 6: what a rendezvous round of the dense Gaussian's resident kernel costs — variants (NPHIP_DG_VARIANT: bit 0 no fences, bit 1
agent-scope accesses to the exchanged rows, bit 2 no GEMM), one per process."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from nutpie_amd import _lib
from nutpie_amd.gaussian import dense_precision

D = int(sys.argv[1]); check = len(sys.argv) > 2
P = dense_precision(D)
s = _lib.PyNutsSettings.Diag(1)
s.update(num_tune=30, num_draws=10, num_chains=1024)
smp = _lib.PySampler(s, _lib.DenseGaussianModel(P), device=0, store_draws=check, evals_per_launch=256)
smp.wait()
n = smp._copy("n_steps", np.int64)
rounds = smp.launches * 256
line = f"variant {os.environ.get('NPHIP_DG_VARIANT', '0')} D={D} mode={smp.host_mode}: {n.sum() / smp.seconds / 1e6:.2f} M leapfrogs/s, {smp.seconds:.3f} s, {smp.launches} launches, {smp.seconds / rounds * 1e6:.1f} us per round"
if check:
    got = smp.take_results()
    want = oracle.sample_dense(oracle.default_settings(seed=1, num_chains=1024, num_tune=30, num_draws=10, n_threads=16), P)
    line += f"; bit-identical to the oracle: {bool(np.array_equal(got.draws, want.draws))}"
if int(os.environ.get("NPHIP_DG_VARIANT", "0")) & 32:
    import ctypes as C
    out = (C.c_int64 * 16)()
    _lib.lib().nphip_sampler_profile(smp._h, out) if smp._h else None
    pr = list(out)
    if pr[12]:
        r = pr[12]
        line += f"; per wave and round: wait-positions {pr[8] / r:.0f} cyc, GEMM {pr[9] / r:.0f} cyc, wait-gradients {pr[10] / r:.0f} cyc, round {pr[11] / r / 100:.1f} us (=> clock {(pr[8] + pr[9] + pr[10]) / (pr[11] / 100):.0f} MHz)"
print(line, flush=True)
