"""A few points of the throughput grid: usage grid_pts.py D [D ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
ds = [int(a) for a in sys.argv[1:]]
sys.argv = sys.argv[:1]
import ab
for d in ds:
    ab.run(d, 1024, False, E=512, steps=8 if d <= 4000 else 4, warm=8 if d <= 4000 else 4)
