"""Dump the kernel summary of a rocprofv3 (rocpd sqlite) results file as text.
usage: python profiles/summarize.py gpurun_out/<dir>/<name>_results.db > profiles/<name>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("# rocprofv3 --kernel-trace --stats summary (from", sys.argv[1].split("/")[-1] + ")")
print(f"{'kernel':70s} {'calls':>7s} {'total_us':>14s} {'avg_us':>12s} {'pct':>7s}")
for name, calls, total, avg, pct in cur.execute("select * from top_kernels"):
    print(f"{name[:70]:70s} {calls:7d} {total:14.1f} {avg:12.2f} {pct:7.2f}")
try:
    rows = list(cur.execute("select name, count(*), avg(value) from counters_collection group by name"))
    if rows:
        print("\n# PMC counters (mean per dispatch)")
        for r in rows:
            print(r)
except Exception:
    pass
