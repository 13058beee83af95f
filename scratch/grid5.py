"""Throughput grid with the engine's default launch length (512 leapfrogs per chain per launch), 1024 chains, tuning phase."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.argv = sys.argv[:1]
import ab
for d in (10, 100, 256, 384, 512, 768, 1000, 1500, 2000, 3000, 4000, 5000, 6000, 7000, 8000, 9000, 10000, 12000):
    ab.run(d, 1024, False, E=512, steps=8 if d <= 4000 else 4, warm=8 if d <= 4000 else 4)
