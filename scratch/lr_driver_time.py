import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nutpie_amd
from nutpie_amd import _lib as hip, low_rank as lr
import symbolic_models as zoo
name = sys.argv[1] if len(sys.argv) > 1 else "radon"
chains, tune, draws = 512, 400, 1000
if name.startswith("demo"):
    sys.path.insert(0, os.path.join(ROOT, "scratch"))
    import lowrank_demo_compiled as demo
    m, _ = demo.target(*{"demo60": (60, 3, 400.0), "demo500": (500, 6, 400.0)}[name])
    chains, tune, draws = 256, 500, 500
else:
    m = zoo.ALL[name]().compile()
for rep in range(2):
    s = hip.PyNutsSettings.LowRank(3)
    s.update(num_tune=tune, num_draws=draws, num_chains=chains)
    t0 = time.perf_counter()
    smp = lr.make_sampler(m, s, None, 1, None, None, None, None)
    t1 = time.perf_counter()
    smp.wait()
    t2 = time.perf_counter()
    log = smp.switch_log
    print("hand-ins by boundary:", {b: (sum(1 for e in log if e[0] == b), round(sum(e[2] for e in log if e[0] == b), 3)) for b in sorted({e[0] for e in log})})
    print(f"{name}: create {t1 - t0:.3f} s, run {t2 - t1:.3f} s, engine seconds {smp.seconds:.3f}, launches {smp.launches}, hand-ins {len(smp.switch_log)}, estimating {sum(e[2] for e in smp.switch_log):.3f} s, first {[(d, round(k, 1), round(sec, 3), nc) for d, k, sec, nc in smp.switch_log[:6]]}")
    smp.close()
# the estimator alone
n, mwin, D = 512, 64, m.n_dim
x = torch.randn(n, mwin, D, dtype=torch.float64, device="cuda"); g = -x + 0.1 * torch.randn_like(x)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); T = lr.estimate(x, g, 1e-5, 2.0); torch.cuda.synchronize(); print(f"estimate(512 x {mwin} x {D}): {time.perf_counter() - t0:.3f} s")
