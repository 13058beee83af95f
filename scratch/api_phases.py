import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import nutpie_amd as nutpie
from nutpie_amd import _lib, sample as S
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
m = nutpie.ar1_gaussian(1000)
t0 = time.perf_counter()
h = nutpie.sample(m, chains=chains, tune=400, draws=1000, seed=1, progress_bar=False, blocking=False)
tc = time.perf_counter()
h._sampler.wait(None)
t1 = time.perf_counter()
print(f"create {tc-t0:.2f} s, wait {t1-tc:.2f} s, engine seconds {h._sampler.seconds if hasattr(h._sampler,'seconds') else -1}")
res = h._sampler.take_results()
t2 = time.perf_counter()
tr = h._extract(res)
t3 = time.perf_counter()
print(f"chains={chains}: sampling {t1-t0:.2f} s, take_results (D2H) {t2-t1:.2f} s ({res.draws.nbytes/1e9/(t2-t1):.1f} GB/s), extract/build_trace {t3-t2:.2f} s")
