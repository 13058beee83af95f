"""Dump the kernel summary of a rocprofv3 (rocpd sqlite) results file as text.
usage: python profiles/summarize.py gpurun_out/<dir>/<name>_results.db [N] > profiles/<name>.txt
N: also print the mean duration of the LAST N dispatches of the kernel with the most time — bench.py's timed region (--steps N) when the
trace is of a bench.py run: the all-dispatch average of the table includes the warm-up and tuning launches in front of it."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
print("# rocprofv3 --kernel-trace --stats summary (from", sys.argv[1].split("/")[-1] + ")")
print(f"{'kernel':70s} {'calls':>7s} {'total_us':>14s} {'avg_us':>12s} {'pct':>7s}")
for name, calls, total, avg, pct in cur.execute("select * from top_kernels"):
    print(f"{name[:70]:70s} {calls:7d} {total:14.1f} {avg:12.2f} {pct:7.2f}")
if len(sys.argv) > 2:
    n_last = int(sys.argv[2])
    top = cur.execute("select name from top_kernels limit 1").fetchone()[0]
    d = [r[0] for r in cur.execute("select duration from kernels where name = ? order by start desc limit ?", (top, n_last))]
    print(f"\n# last {len(d)} dispatches of {top[:70]}: mean {sum(d) / len(d) / 1e3:.2f} us (min {min(d) / 1e3:.2f}, max {max(d) / 1e3:.2f})")
try:
    rows = list(cur.execute("select name, count(*), avg(value) from counters_collection group by name"))
    if rows:
        print("\n# PMC counters (mean per dispatch)")
        for r in rows:
            print(r)
except Exception:
    pass
