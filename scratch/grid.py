"""Throughput grid of the fused engine on one MI355X: leapfrog steps/s vs dimension and vs chain count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
import ab
print("# k_advance throughput grid (AR(1) Gaussian, tuning phase, HIP-event kernel time); columns as printed by scratch/ab.py")
for d, e in [(10, 256), (100, 256), (256, 256), (512, 256), (768, 256), (1000, 256), (1024, 256), (1500, 128), (2000, 128), (3000, 64), (4000, 64), (5000, 64), (6000, 32),
             (7000, 32), (8000, 32), (9000, 32), (10000, 32), (12000, 32)]:
    ab.run(d, 1024, False, E=e, steps=12, warm=40)
for n in (64, 256, 512, 2048, 4096):
    ab.run(1000, n, False, steps=12, warm=40)
ab.run(1000, 256, False, W=4, steps=12, warm=40)
