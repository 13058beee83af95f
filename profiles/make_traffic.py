"""profiles/traffic.json from the committed PMC summaries: HBM bytes and instructions per leapfrog per (dim, waves per chain).

bytes per leapfrog = (2 x FETCH_SIZE + WRITE_SIZE) KB per launch x 1024 / leapfrogs per launch
(MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads; WRITE_SIZE as reported).
instructions per leapfrog = SQ_INSTS_{VALU, SALU, LDS, VMEM_RD, VMEM_WR} per launch / leapfrogs per launch;
issuing / waiting fraction = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES, SQ_WAIT_ANY / SQ_WAVE_CYCLES.
`leapfrogs_per_launch` is what the profiled command ran: chains x evals_per_launch.
usage: python profiles/make_traffic.py   (rewrites profiles/traffic.json)"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
# (dim, waves) -> (summary file, leapfrogs per launch of the profiled run, kernel, configuration of the profiled run)
SOURCES = {
    "1000:1": ("r6_d1000_timed_config_pmc.txt", 1024 * 2048, "k_advance<fused,W=1,NV=8>",
               "bench.py default: sampling phase, positions stored, 2048 leapfrogs per chain per launch; the timed launches"),
    "2000:2": ("r1_d2000_multiwave_kernel_pmc.txt", 1024 * 128, "k_advance<fused,W=2,NV=8>", "round 1: tuning phase, 128 leapfrogs per chain per launch"),
    "10000:4": ("r5_d10000_timed_config_pmc.txt", 1024 * 512, "k_advance<fused,W=4,NV=20,lean>",
                "bench.py --dim 10000: sampling phase, positions stored, 512 leapfrogs per chain per launch; the timed launches"),
}
# whole jobs (round 5, scratch/r5_pmc_jobs.py): the first line of the summary holds the job's leapfrogs and launches
JOBS = {
    "config3_compiled_density": ("r6_config3_compiled_density_final_pmc.txt", "k_advance<callback,W=1,NV=2,REMOTE> of the generated radon density (512 chains, tune 400 + draws 1000)"),
    "config3_traced_torch_density": ("r6_config3_traced_torch_density_final_pmc.txt", "the same model traced from its torch log-density (nutpie_amd.torch_trace)"),
    "low_rank_173_k4": ("r5_low_rank_d173_k4_pmc.txt", "k_advance<fused,W=1,NV=2,LR>: AR(1) Gaussian D = 173, 512 chains, 4 columns handed in; the launches after the hand-in"),
    "config2ii_dense_resident": ("r6_dense_resident_pmc.txt", "k_advance<callback,W=1,NV=8,REMOTE,DENSEG>: the dense 1000-dim Gaussian's resident kernel (launch-wide fp64 MFMA GEMM inside the leaf), 1024 chains; 12 launches of 256 evaluation rounds in the sampling phase (scratch/r6_dense_job.py)"),
    "low_rank_1000_k16": ("r5_low_rank_d1000_k16_pmc.txt", "k_advance<fused,W=1,NV=8,LR>: AR(1) Gaussian D = 1000, 1024 chains, 16 columns handed in; the launches after the hand-in"),
}
# summaries a key falls back to while its timed-configuration passes have not been taken yet
FALLBACK = {
    "1000:1": ("r2_d1000_kernel_pmc.txt", 1024 * 256, "round 2: tuning phase, 256 leapfrogs per chain per launch"),
    "10000:4": ("r2_d10000_lean_lds_slot_e128_pmc.txt", 1024 * 128, "round 2: tuning phase, 128 leapfrogs per chain per launch"),
}


def counter(txt, name):
    m = re.search(name + r"\s+n=\s*\d+\s+mean=([0-9.e+]+)", txt)
    return float(m.group(1)) if m else None


def main():
    out = {}
    for key, (fn, lpl, kernel, config) in SOURCES.items():
        if not os.path.exists(os.path.join(HERE, fn)) and key in FALLBACK:
            fn, lpl, config = FALLBACK[key]
        txt = open(os.path.join(HERE, fn)).read()
        f, w = counter(txt, "FETCH_SIZE"), counter(txt, "WRITE_SIZE")
        e = {"bytes_per_leapfrog": (2 * f + w) * 1024 / lpl, "leapfrogs_per_launch": lpl, "kernel": kernel, "source": "profiles/" + fn,
             "config": config, "fetch_size_kb": f, "write_size_kb": w}
        insts = {k: counter(txt, "SQ_INSTS_" + k.upper()) for k in ("valu", "salu", "lds", "vmem_rd", "vmem_wr")}
        if all(v is not None for v in insts.values()):
            e["insts_per_leapfrog"] = {k: v / lpl for k, v in insts.items()}
            e["insts_per_leapfrog"]["total"] = sum(insts.values()) / lpl
        wc, act, wait = counter(txt, "SQ_WAVE_CYCLES"), counter(txt, "SQ_ACTIVE_INST_ANY"), counter(txt, "SQ_WAIT_ANY")
        if wc and act:
            e["issuing_fraction"] = act / wc
            e["waiting_fraction"] = wait / wc if wait else None
            e["wave_quad_cycles_per_leapfrog"] = wc / lpl
        out[key] = e
    for key, (fn, config) in JOBS.items():
        path = os.path.join(HERE, fn)
        if not os.path.exists(path):
            continue
        txt = open(path).read()
        m = re.search(r"leapfrogs=(\d+) launches=(\d+)", txt)
        lpl = int(m.group(1)) / int(m.group(2))
        f, w = counter(txt, "FETCH_SIZE"), counter(txt, "WRITE_SIZE")
        e = {"bytes_per_leapfrog": (2 * f + w) * 1024 / lpl, "leapfrogs_per_launch": lpl, "kernel": config.split(":")[0].split(" of ")[0], "source": "profiles/" + fn, "config": config,
             "fetch_size_kb": f, "write_size_kb": w, "note": "FETCH_SIZE / WRITE_SIZE count L2 misses: traffic that the 256 MB Infinity Cache serves is in them"}
        insts = {k: counter(txt, "SQ_INSTS_" + k.upper()) for k in ("valu", "salu", "lds", "vmem_rd", "vmem_wr")}
        e["insts_per_leapfrog"] = {k: v / lpl for k, v in insts.items()}
        e["insts_per_leapfrog"]["total"] = sum(insts.values()) / lpl
        wc, act, wait = counter(txt, "SQ_WAVE_CYCLES"), counter(txt, "SQ_ACTIVE_INST_ANY"), counter(txt, "SQ_WAIT_ANY")
        e["issuing_fraction"], e["waiting_fraction"], e["wave_quad_cycles_per_leapfrog"] = act / wc, wait / wc, wc / lpl
        e["waves"] = counter(txt, "SQ_WAVES")
        out[key] = e
    json.dump(out, open(os.path.join(HERE, "traffic.json"), "w"), indent=1)
    for k, v in out.items():
        if ":" not in k:
            print(f"{k}: {v['bytes_per_leapfrog']:.0f} B per leapfrog past L2, {v['insts_per_leapfrog']['total']:.0f} instructions per leapfrog, "
                  f"{100 * v['issuing_fraction']:.0f} % of wave cycles issuing, {100 * v['waiting_fraction']:.0f} % waiting   [{v['source']}]")
            continue
        d = int(k.split(":")[0])
        extra = f", {v['insts_per_leapfrog']['total']:.0f} instructions per leapfrog, {100 * v['issuing_fraction']:.0f} % of wave cycles issuing" if "insts_per_leapfrog" in v and "issuing_fraction" in v else ""
        print(f"{k}: {v['bytes_per_leapfrog']:.0f} B per leapfrog = {v['bytes_per_leapfrog'] / (40 * d):.2f} x algorithmic (40 D){extra}   [{v['source']}]")


if __name__ == "__main__":
    main()
