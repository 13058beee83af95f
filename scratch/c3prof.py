"""cycle attribution of the resident density kernel: NUTPIE_AMD_JIT_FLAGS=-DNPHIP_PROFILE python scratch/c3prof.py"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nutpie_amd import _lib as hip
from nutpie_amd.radon import radon_density_model
m = radon_density_model()
s = hip.PyNutsSettings.Diag(20260926)
s.update(num_tune=400, num_draws=1000, num_chains=512)
smp = m._make_sampler(s, None, 1, None, None, None, None)
smp.wait()
out = (C.c_int64 * 16)(); hip.lib().nphip_sampler_profile(smp._h, out); o = list(out)
n = smp._copy("n_steps", np.int64)
print(f"job {smp.seconds:.3f} s, {n.sum()/smp.seconds/1e6:.2f} M leapfrogs/s")
if o[3]:
    print(f"cycles/leaf (hot) {o[1]/max(o[4],1):.0f} (n={o[4]}) = math+density {o[0]/o[3]:.0f} + reduce4 {o[6]/o[3]:.0f} + stores/issue {o[7]/max(o[4],1):.0f} + collector {o[8]/o[3]:.0f} + level0 {o[9]/o[3]:.0f} + level>=1 {o[10]/o[3]:.0f} + merge {o[11]/o[3]:.0f}")
    print(f"rare leaves: {o[2]/max(o[5],1):.0f} cycles each (n={o[5]}); per draw: regrad {o[12]/max(o[5],1):.0f} adapt/position pass {o[13]/max(o[5],1):.0f} begin_draw {o[14]/max(o[5],1):.0f}")
    tot = o[1] + o[2]
    print(f"share of cycles: hot leaves {o[1]/tot:.2f}, rare (draw ends etc.) {o[2]/tot:.2f}")
