"""profiles/traffic.json from the committed PMC summaries: HBM bytes per leapfrog per (dim, waves per chain).

bytes per leapfrog = (2 x FETCH_SIZE + WRITE_SIZE) KB per launch x 1024 / leapfrogs per launch
(MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads; WRITE_SIZE as reported).
`leapfrogs_per_launch` is what the profiled command ran: chains x evals_per_launch.
usage: python profiles/make_traffic.py   (rewrites profiles/traffic.json)"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
# (dim, waves) -> (summary file, leapfrogs per launch of the profiled run, kernel)
SOURCES = {
    "1000:1": ("r2_d1000_kernel_pmc.txt", 1024 * 256, "k_advance<fused,W=1,NV=8>"),
    "2000:2": ("r1_d2000_multiwave_kernel_pmc.txt", 1024 * 128, "k_advance<fused,W=2,NV=8>"),
    "10000:4": ("r2_d10000_lean_lds_slot_e128_pmc.txt", 1024 * 128, "k_advance<fused,W=4,NV=20,lean>"),
}
out = {}
for key, (fn, lpl, kernel) in SOURCES.items():
    txt = open(os.path.join(HERE, fn)).read()
    f = float(re.search(r"FETCH_SIZE\s+n=\s*\d+\s+mean=([0-9.e+]+)", txt).group(1))
    w = float(re.search(r"WRITE_SIZE\s+n=\s*\d+\s+mean=([0-9.e+]+)", txt).group(1))
    out[key] = {"bytes_per_leapfrog": (2 * f + w) * 1024 / lpl, "leapfrogs_per_launch": lpl, "kernel": kernel, "source": "profiles/" + fn,
                "fetch_size_kb": f, "write_size_kb": w}
json.dump(out, open(os.path.join(HERE, "traffic.json"), "w"), indent=1)
for k, v in out.items():
    d = int(k.split(":")[0])
    print(f"{k}: {v['bytes_per_leapfrog']:.0f} B per leapfrog = {v['bytes_per_leapfrog'] / (40 * d):.2f} x algorithmic (40 D)")
