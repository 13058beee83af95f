"""Round-2 throughput grid: lean register kernels (4096 < D <= 10240) against the memory-resident kernels of the same geometry,
and the 1000-dim headline kernel with one and two waves per chain."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
import ab
print("# round 2 grid (AR(1) Gaussian, 1024 chains, tuning phase, HIP-event kernel time)")
for d, e in [(4200, 64), (5000, 64), (6000, 32), (8000, 32), (10000, 32)]:
    ab.run(d, 1024, False, E=e, steps=12, warm=24)
    ab.run(d, 1024, True, E=e, steps=12, warm=24)
ab.run(10000, 256, False, E=32, steps=12, warm=24)
ab.run(10000, 2048, False, E=32, steps=8, warm=16)
for d, w in [(1000, 0), (1000, 2), (1000, 4), (1500, 0), (2000, 0), (3000, 0), (4000, 0), (2000, 8), (4000, 8)]:
    ab.run(d, 1024, False, W=w, E=128, steps=12, warm=40)
