"""Round 5: chains per workgroup of a runtime-compiled density's resident kernel (NPHIP_JIT_CPB=1|2|4; default: by the job's size).
One process per setting (the override is read once); prints the job rate and a digest of the draws (must not depend on the setting)."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from nutpie_amd import _lib as hip
from nutpie_amd.radon import radon_symbolic_model, radon_traced_model

chains = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tag = f"cpb={os.environ.get('NPHIP_JIT_CPB', 'auto')} chains={chains}"
spec = os.environ.get("SPECIALIZE", "1") != "0"
tag += f" specialize={int(spec)}"
for label, make in (("generated", lambda: radon_symbolic_model().compile(specialize=spec)), ("traced", radon_traced_model)):
    m = make()
    for rep in range(2):
        s = hip.PyNutsSettings.Diag(20260926)
        s.update(num_tune=400, num_draws=1000, num_chains=chains)
        smp = m._make_sampler(s, None, 1, None, None, None, None)
        smp.wait()
        n = smp._copy("n_steps", np.int64)
        e = smp._copy("energy", np.float64)
        dig = hashlib.sha256(n.tobytes() + e.tobytes()).hexdigest()[:12]
        print(f"[{tag}] {label} rep {rep}: {n.sum() / smp.seconds / 1e6:.2f} M leapfrogs/s, job {smp.seconds:.3f} s, leapfrogs {int(n.sum())}, digest {dig}")
        smp.close()
