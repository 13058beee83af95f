"""Lean kernels (4 waves per chain) with the LDS summary slot, 1024 chains."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
import ab
for d, e in [(4600, 64), (5000, 64), (6000, 32), (6200, 32), (7000, 32), (8000, 32), (9000, 32), (10000, 32), (10240, 32)]:
    ab.run(d, 1024, False, W=4, E=e, steps=10, warm=20)
