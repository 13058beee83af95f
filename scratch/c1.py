import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nutpie_amd
for rep in range(3):
    t = time.time(); tr = nutpie_amd.sample(nutpie_amd.std_normal(10), chains=4, tune=400, draws=1000, seed=123, progress_bar=False); el = time.time() - t
    n = int(tr.sample_stats.n_steps.values.sum() + tr.warmup_sample_stats.n_steps.values.sum())
    print(f"config1 stdnormal D=10, 4 chains, call {rep}: {el*1e3:.0f} ms wall, {n} leapfrogs")
