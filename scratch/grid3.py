"""Lean kernels: 4 waves per chain (VGPR + AGPR state) against 8 waves per chain, 1024 chains."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
import ab
print("# lean kernels, W = 8 vs W = 4 (AR(1) Gaussian, 1024 chains, tuning phase, HIP-event kernel time)")
for d, e in [(4200, 64), (5000, 64), (6000, 32), (7000, 32), (8000, 32), (9000, 32), (10000, 32)]:
    for w in (8, 4):
        ab.run(d, 1024, False, W=w, E=e, steps=10, warm=20)
ab.run(10000, 256, False, W=4, E=32, steps=10, warm=20)
ab.run(10000, 2048, False, W=4, E=32, steps=6, warm=12)
