import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd import _lib as hip
from nutpie_amd.gaussian import ar1_gaussian
def run(dim, chains, noreg, W=0, E=256, steps=40, warm=40, nocache=False):
    model = ar1_gaussian(dim)
    s = hip.PyNutsSettings.Diag(20260926)
    s.update(num_tune=400, num_draws=(steps+warm+2)*E, num_chains=chains)
    smp = hip.PySampler(s, hip.TridiagGaussianModel(model.diag, model.offdiag), waves_per_chain=W, store_draws=False, evals_per_launch=E, manual=True, no_register_kernel=noreg, no_stream_cache=nocache)
    smp.step(warm)
    n0 = sum(p.total_num_steps for p in smp.progress())
    t=time.perf_counter(); done, l, ms = smp.step(steps); el=time.perf_counter()-t
    n1 = sum(p.total_num_steps for p in smp.progress())
    print(f"dim={dim} chains={chains} W={smp.waves_per_chain} noreg={noreg} nocache={nocache} E={E}: {(n1-n0)/el/1e6:.1f} M leapfrogs/s, kernel {ms/steps:.3f} ms/launch, {ms/steps/E*1e3:.2f} us/leapfrog/chain, alg {(n1-n0)*40*dim/(ms/1e3)/1e9:.0f} GB/s")
    import ctypes as C
    out=(C.c_int64*16)(); hip.lib().nphip_sampler_profile(smp._h, out); o=list(out)
    if o[3]: print(f"   cycles/leaf: total(hot) {o[1]/max(o[4],1):.0f} (n={o[4]}) = math {o[0]/o[3]:.0f} + reduce4 {o[6]/o[3]:.0f} + stores/issue {o[7]/max(o[4],1):.0f} + cascade(rest);  draw-end {o[2]/max(o[5],1):.0f} (n={o[5]})")
    if o[3] and o[15]: print(f"   lean: sweep 1 (+ source reload, barrier) {o[15]/o[3]:.0f} of math {o[0]/o[3]:.0f}")
    if o[3]: print(f"   cascade per leaf: collector {o[8]/o[3]:.0f}  level0-check {o[9]/o[3]:.0f}  level>=1 checks {o[10]/o[3]:.0f}  merge scalar {o[11]/o[3]:.0f}")
    if o[5]: print(f"   draw end per draw: regrad {o[12]/o[5]:.0f}  adapt/position pass {o[13]/o[5]:.0f}  begin_draw (momentum, first leaf issue) {o[14]/o[5]:.0f}")
    smp.close()
for a in sys.argv[1:]:
    exec(a)
