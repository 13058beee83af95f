"""Catch the matrix on which nphip_batched_eigh gives up inside a low_rank job and save it (gpurun_out/eigh_fail.npy)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from nutpie_amd import _lib as hip, low_rank as lr
import symbolic_models as zoo
orig = hip.batched_eigh
def wrapped(A):
    try:
        return orig(A)
    except RuntimeError as e:
        msg = str(e); print(msg)
        i = int(msg.split("matrix ")[1].split()[0])
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.save(os.path.join(ROOT, "gpurun_out", "eigh_fail.npy"), A[i].cpu().numpy())
        w = torch.linalg.eigvalsh(A[i:i + 1])[0]
        print("order", A.shape[-1], "finite", bool(torch.isfinite(A[i]).all()), "eigs lo/hi", w[:4].tolist(), w[-3:].tolist())
        raise
hip.batched_eigh = wrapped
m = zoo.radon().compile()
s = hip.PyNutsSettings.LowRank(3)
s.update(num_tune=400, num_draws=1000, num_chains=512)
smp = lr.make_sampler(m, s, None, 1, None, None, None, None)
try:
    smp.wait()
except Exception as e:
    print("job failed:", str(e)[:100])
