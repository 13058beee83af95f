"""Round 6: two concurrent 512-chain radon jobs — both on default-priority streams against the second on a high-priority stream (hardware queues)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import _lib as hip
from nutpie_amd.radon import radon_symbolic_model
m = radon_symbolic_model().compile()
def settings(seed):
    s = hip.PyNutsSettings.Diag(seed); s.update(num_tune=400, num_draws=1000, num_chains=512); return s
junk = [torch.cuda.Stream() for _ in range(3)]   # (a process that has used torch streams, like bench.py)
for rep in range(8):
    for prio in (False, True):
        hi = torch.cuda.Stream(0, priority=-1)
        smps = [m._make_sampler(settings(20260926 + k), None, 1, None, None, None, None, **({"stream": hi.cuda_stream} if (k and prio) else {})) for k in range(2)]
        for smp in smps: smp.wait()
        n = sum(int(smp._copy("n_steps", np.int64).sum()) for smp in smps); secs = [smp.seconds for smp in smps]
        for smp in smps: smp.close()
        print(f"rep {rep} second job on a high-priority stream: {prio}:  {n / max(secs) / 1e6:6.1f} M leapfrogs/s  (engine seconds {secs[0]:.3f} / {secs[1]:.3f})", flush=True)
