"""Per-launch log of a low-rank job on a compiled density: kernel time, chains running / waiting / finished and the draw range the
chains are at, one line per launch.  Shows where the wall-clock of the manual-mode driver goes."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import nutpie_amd
from nutpie_amd import _lib as hip, low_rank as lr
import symbolic_models as zoo
name = sys.argv[1] if len(sys.argv) > 1 else "radon"
m = zoo.ALL[name]().compile()
s = hip.PyNutsSettings.LowRank(3)
s.update(num_tune=400, num_draws=1000, num_chains=512)
import threading
_start = threading.Thread.start
threading.Thread.start = lambda self: None      # the driver thread is not started: this script drives the engine itself
smp = lr.make_sampler(m, s, None, 1, None, None, None, None)
threading.Thread.start = _start
inner = smp._inner
rows = []
held = 0
LOCKSTEP = 'lockstep' in sys.argv   # the round-3 driver: nobody gets a metric until every chain has stopped
with smp._step_lock:
    while True:
        done, cnt, ms = inner.step(1)
        code = inner.waiting_codes()
        pr = inner.progress()
        fin = np.array([p.finished_draws for p in pr]); st = np.array([p.total_num_steps for p in pr])
        rows.append((ms, int((code == 0).sum()), int((code == 1).sum()), int((code == 2).sum()), int(fin.min()), int(np.median(fin)), int(fin.max()), int(st.max()), int(st.mean())))
        if done:
            break
        wait = code == 1
        if wait.any():
            n_wait, n_run = int(wait.sum()), int((code == 0).sum())
            if LOCKSTEP:
                if n_run == 0:
                    smp._adapt(np.nonzero(wait)[0])
            elif n_run == 0 or 4 * n_wait >= n_wait + n_run or held >= lr.HOLD_LAUNCHES:
                smp._adapt(np.nonzero(wait)[0]); held = 0
            else:
                held += 1
for i, r in enumerate(rows):
    print(i, "ms %.2f run %d wait %d done %d draws min/med/max %d %d %d steps max/mean %d %d" % r)
print("total kernel ms", sum(r[0] for r in rows), "launches", len(rows), "switches", smp.switch_log)
