"""Round 5 (final build): cycles of the sections of one evaluation of config 3's generated density (Model.profile)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nutpie_amd.radon import radon_symbolic_model

for spec in (True, False):
    m = radon_symbolic_model()
    m._specialize = spec
    prof = m.profile()
    print(f"[specialize={int(spec)}] one evaluation: {sum(c for _, c in prof):.0f} cycles")
    for name, c in prof:
        if c > 100:
            print(f"[specialize={int(spec)}]    {c:8.0f}  {name}")
