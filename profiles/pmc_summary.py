"""Summarise rocprofv3 --pmc passes (rocpd sqlite files) for the k_advance kernel: mean per dispatch.
usage: python profiles/pmc_summary.py gpurun_out/pmc_<tag>/ [skip_first_n_dispatches | -last_n_dispatches]
A negative second argument keeps only the LAST n dispatches of the kernel — the timed region of bench.py (its K timed launches
come last: warm-up of the chains, W warm launches, K timed launches)."""
import glob
import os
import sqlite3
import sys

KERNEL = os.environ.get("PMC_KERNEL", "k_advance")   # substring of the kernel's name (round 6: k_lr_estimate)

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
which = f"the last {-skip} dispatches" if skip < 0 else f"first {skip} dispatches skipped"
print(f"# PMC counters for {KERNEL} dispatches in {d} (mean per dispatch, {which})")
for f in sorted(glob.glob(d + "/*_results.db")):
    db = sqlite3.connect(f)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    rows = list(cur.execute("select * from counters_collection"))
    ik, ic, iv, idp = cols.index("kernel_name") if "kernel_name" in cols else None, cols.index("counter_name"), cols.index("value"), cols.index("dispatch_id")
    agg = {}
    for r in rows:
        if ik is not None and KERNEL not in str(r[ik]):
            continue
        agg.setdefault(r[ic], {}).setdefault(r[idp], 0.0)
        agg[r[ic]][r[idp]] += r[iv]
    for name, per in agg.items():
        ids = sorted(per)[skip:] if skip >= 0 else sorted(per)[skip:]
        vals = [per[i] for i in ids]
        if vals:
            print(f"{name:28s} n={len(vals):4d} mean={sum(vals)/len(vals):.6g}")
