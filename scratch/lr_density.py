"""adaptation="low_rank" on compiled densities: the resident kernel under the metric (round 4) against the batched callback on the
memory-resident kernels (round 3), and against "diag".  usage: python scratch/lr_density.py [model [chains]]"""
import sys, os, time, dataclasses
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import nutpie_amd
import symbolic_models as zoo
name = sys.argv[1] if len(sys.argv) > 1 else "radon"
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 512
m = zoo.ALL[name]().compile()
for label, model, adaptation in (("diag, resident", m, "diag"), ("low_rank, resident kernel under the metric", m, "low_rank"),
                                 ("low_rank, batched callback (memory-resident kernels)", dataclasses.replace(m, _resident=False), "low_rank")):
    for rep in range(2):
        t0 = time.perf_counter()
        tr = nutpie_amd.sample(model, adaptation=adaptation, chains=chains, tune=400, draws=1000, seed=3, progress_bar=False)
        dt = time.perf_counter() - t0
    n = tr.sample_stats.n_steps.values; nw = tr.warmup_sample_stats.n_steps.values
    print(f"{name} D={m.n_dim} chains={chains} {label}: job {dt:.3f} s, {(n.sum() + nw.sum()) / dt / 1e6:.2f} M leapfrogs/s, {n.mean():.1f} leapfrogs per draw (sampling), "
          f"divergences {int(tr.sample_stats.diverging.values.sum())}", flush=True)
