"""Round 6: the low-rank job's wall time on radon (512 chains): engine seconds, wall, and the hand-in log."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import _lib as hip, low_rank
from nutpie_amd.radon import radon_symbolic_model
m = radon_symbolic_model().compile()
for rep in range(3):
    s = hip.PyNutsSettings.LowRank(20260926)
    s.update(num_tune=400, num_draws=1000, num_chains=512)
    t0 = time.perf_counter()
    smp = low_rank.make_sampler(m, s, None, 1, None, None, None, None)
    t1 = time.perf_counter()
    smp.wait()
    t2 = time.perf_counter()
    log = list(smp.switch_log)
    print(f"rep {rep}: create {t1 - t0:.3f} s, wait {t2 - t1:.3f} s, engine {smp.seconds:.3f} s, launches {smp.launches}, hand-ins {len(log)}, estimating {sum(e[2] for e in log):.3f} s", flush=True)
    for e in log[:60]:
        print(f"   boundary {e[0]:4d}: {e[3]:4d} chains, {e[2] * 1e3:7.1f} ms, mean columns {e[1]:.3f}, at {e[4] * 1e3:7.1f} ms")
    smp.close()

if os.environ.get('LR_ONLY'): sys.exit(0)
# the floor: the same job with the low-rank kernels and no hand-in at all (every chain keeps the diagonal metric it adapts itself)
s = hip.PyNutsSettings.LowRank(20260926)
s.update(num_tune=400, num_draws=1000, num_chains=512)
inner_settings = s.clone(); inner_settings.update(low_rank_metric=True, store_gradient=True)
for rep in range(2):
    t0 = time.perf_counter()
    inner = m._make_sampler(inner_settings, None, 1, None, None, None, None, manual=True)
    while True:
        done, _, _ = inner.step(16)
        if done: break
    print(f"no hand-ins, low-rank kernels: wall {time.perf_counter() - t0:.3f} s, engine {inner.seconds:.3f} s, launches {inner.launches}", flush=True)
    inner.close()
s = hip.PyNutsSettings.Diag(20260926)
s.update(num_tune=400, num_draws=1000, num_chains=512)
for rep in range(2):
    t0 = time.perf_counter()
    inner = m._make_sampler(s, None, 1, None, None, None, None)
    inner.wait()
    print(f"diag: wall {time.perf_counter() - t0:.3f} s, engine {inner.seconds:.3f} s, launches {inner.launches}", flush=True)
    inner.close()
