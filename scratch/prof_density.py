"""Cycle attribution of a generated density (nutpie_amd.symbolic.Model.profile).  usage: python scratch/prof_density.py [model] [waves]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import symbolic_models as zoo

name = sys.argv[1] if len(sys.argv) > 1 else "radon"
waves = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = zoo.ALL[name]().profile(n_chains=1024 // waves, waves_per_chain=waves)
total = sum(c for _, c in rows)
print(f"{name}, {waves} wave(s) per chain, {1024 // waves} chains: {total:.0f} cycles per evaluation")
for label, c in rows:
    print(f"  {c:8.0f}  {label}")
