"""Turn the rocprofv3 outputs of tools/profile_round2.sh into the small summaries that are
committed under profiles/ (and that bench.py reads): per-launch means of the instruction
counters of ais_half_kernel per ntransitions setting, HBM traffic, kernel-stats rows."""
import collections
import csv
import glob
import json
import os
import sys

O = sys.argv[1]


def means(pattern, sub="ais_half"):
    agg = collections.defaultdict(list)
    for fn in glob.glob(os.path.join(O, pattern, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if sub in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


insts = {}
for nt in (1, 16, 100):
    m, n = means(f"pmc_inst_nt{nt}")
    if m:
        insts[str(nt)] = dict(m, dispatches=max(n.values()))
json.dump({"command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD "
                      "SQ_INSTS_BRANCH -- python3 bench.py --ntransitions NT --no-alt --no-smc --no-cpu-baseline "
                      "--steps 50 --warmup 5 --min-seconds 0.2",
           "kernel": "ais_half_kernel<8, rosenbrock, BOX, kernelized>", "unit": "wave-instructions per launch (mean)",
           **insts}, open(os.path.join(O, "pmc_insts.json"), "w"), indent=1)
f, nf = means("pmc_fetch")
w, nw = means("pmc_write")
if f and w:
    fk, wk = f["FETCH_SIZE"], w["WRITE_SIZE"]
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 bench.py --no-alt "
                          "--no-smc --no-cpu-baseline --steps 30 --warmup 5 --min-seconds 0.1 (ntransitions = 100)",
               "kernel": "ais_half_kernel<8, rosenbrock, BOX, kernelized>", "dispatches": nf["FETCH_SIZE"],
               "FETCH_SIZE_KB_mean": fk, "WRITE_SIZE_KB_mean": wk,
               "gfx950_correction": "FETCH_SIZE counts 64 B per 128-B request for wide (16 B/lane) reads: x2 "
                                    "(MI355X_MICROARCH.md, HBM)",
               "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024),
               "algorithmic_bytes_per_launch": 32768 * 100 * 224, "ntransitions": 100},
              open(os.path.join(O, "pmc_traffic_nt100.json"), "w"), indent=1)
fs, nfs = means("pmc_fetch_smc", "smc_loop")
ws, nws = means("pmc_write_smc", "smc_loop")
if fs and ws:
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 tools/smc_c4_probe.py",
               "kernel": "smc_loop_kernel<16, hier_gauss_sim, SIMPLE> (one launch = the whole C4 run, 190 iterations; "
                         "mean over the warm-up and the timed launch)",
               "dispatches": nfs["FETCH_SIZE"], "FETCH_SIZE_KB_mean": fs["FETCH_SIZE"], "WRITE_SIZE_KB_mean": ws["WRITE_SIZE"],
               "gfx950_correction": "FETCH_SIZE x2 for wide reads (MI355X_MICROARCH.md, HBM)",
               "hbm_bytes_per_launch": int(2 * fs["FETCH_SIZE"] * 1024 + ws["WRITE_SIZE"] * 1024),
               "algorithmic_bytes_per_launch": 6225920 * 545},
              open(os.path.join(O, "pmc_traffic_smc_loop.json"), "w"), indent=1)
print("smc traffic", fs, ws)
c, _ = means("pmc_cyc")
print("insts", json.dumps(insts)[:600])
print("traffic", f, w)
print("cycles", c)
for d in ("stats_nt1", "stats_nt16", "stats_nt100", "smc_stats", "smc_stats_kernels", "readme_stats"):
    for fn in glob.glob(os.path.join(O, d, "**", "*kernel_stats.csv"), recursive=True):
        print(d, fn)
        print("".join(open(fn).readlines()[:6]))
