#!/usr/bin/env python3
"""bench.py -- walker proposal+cost evaluations per second of the AIS hot path.

Workload (BASELINE.json configs[2]): AIS, 65 536 walkers per GPU,
D = 8, prior Factored(Uniform(-5,5))^8, cost sqrt(sum 100(x[k+1]-x[k]^2)^2 +
(1-x[k])^2), ApproxKernelizedPosterior scale 1.0, seed 1, ntransitions = 100 -- the
value the reference itself samples with (README.md:57, four of the AIS testsets of
test/runtests.jl; BASELINE.json configs[0]); SURVEY §8d's ntransitions = 16 is timed
in the same run and reported beside it as `also_at_ntransitions_16`.
A "step" is one GENERATION: every walker receives `ntransitions` transition!()
calls (two half-generation kernel launches per GPU; for N>1 one RCCL all-gather
after each, i.e. the exchange is amortised over ntransitions sub-steps exactly as
the reference amortises its per-sample overhead).  Weak scaling: per-GPU walkers are
fixed, N_total = 65 536 * gpus.

Prints ONE JSON line (rank 0).  `roofline` is for the half-generation kernel:
algorithmic bytes per launch = rows * ntransitions * 8(3D+4)
over the kernel's average duration measured with hipEvents on its stream over
the timed region.  `cpu_baseline` times the CPU oracle's faithful serial
restatement of the reference on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WALKERS_PER_GPU = 65536
D = 8
NT = 100
NT_ALT = 16
SEED = 1
HBM_PEAK_GBS = 8000.0  # MI355X spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def build_model(k):
    prior = k.Factored(*[k.Uniform(-5, 5)] * D)
    return k.ApproxKernelizedPosterior(prior, k.costs.Rosenbrock(), 1.0)


def _cpu_chain(args):
    """One independent serial chain (the MCMCThreads analogue, src/KissABC.jl:108)."""
    nwalkers, seed, budget_s = args
    import kissabc_jl_amd as k
    from oracle import oracle as orc
    o = orc.OracleAIS(build_model(k), nwalkers, seed=seed).init()
    o.steps_serial(1024, NT, collect=False)  # warm-up
    nsteps, done, t0 = 4096, 0, time.perf_counter()
    while True:
        o.steps_serial(nsteps, NT, collect=False)
        done += nsteps
        el = time.perf_counter() - t0
        if el > budget_s:
            return done, el


def cpu_baseline(k, budget_s):
    """Oracle, serial reference schedule (src/KissABC.jl:66-80), same model, bounded
    sample: (i) 1 core, N = 65 536 walkers, as many step() calls as fit the budget;
    (ii) all host cores as independent chains of N/cores walkers each."""
    import multiprocessing as mp
    done, el = _cpu_chain((WALKERS_PER_GPU, SEED, budget_s))
    out = {
        "value": done * NT / el, "unit": "evals/s", "cores": 1, "kind": "port",
        "sample": f"oracle ref_serial (C restatement of src/transition.jl + src/KissABC.jl:66-80), "
                  f"N={WALKERS_PER_GPU} D={D} rosenbrock, {done} step() calls x ntransitions={NT} "
                  f"in {el:.1f}s on 1 host core",
    }
    try:
        cores = len(os.sched_getaffinity(0))
        if cores > 1:
            per = max(D + 5, WALKERS_PER_GPU // cores)
            with mp.get_context("spawn").Pool(cores) as pool:
                res = pool.map(_cpu_chain, [(per, SEED + 1 + c, budget_s / 2) for c in range(cores)])
            out["all_cores"] = {
                "value": sum(d * NT / e for d, e in res), "cores": cores, "unit": "evals/s",
                "sample": f"{cores} independent chains of {per} walkers (MCMCThreads analogue), "
                          f"{budget_s / 2:.0f}s each"}
    except Exception as e:  # the baseline is informational; never fail the bench on it
        out["all_cores"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ntransitions", type=int, default=NT)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true",
                    help="skip the secondary ntransitions=16 region (clean rocprof summaries)")
    args = ap.parse_args()
    nt = args.ntransitions

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # The CPU baseline runs FIRST, before this process touches the GPU: its all-core
    # leg spawns worker processes, and a process that has initialised HIP must not
    # fork+exec on this pool.
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        import kissabc_jl_amd as k0
        cpu = cpu_baseline(k0, args.cpu_seconds)

    import torch
    import torch.distributed as dist

    import kissabc_jl_amd as k
    from kissabc_jl_amd.sharded import ShardedAIS
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # dry-run knobs (tests/test_gpu_bench_two_ranks.py): all ranks on one device, exchange
    # over gloo through host memory -- exercises this script's N > 1 branch on a 1-GPU box
    backend = os.environ.get("KABC_BENCH_BACKEND", "nccl")
    if os.environ.get("KABC_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["KABC_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_pg = world > 1 or os.environ.get("KABC_FORCE_COLLECTIVE") == "1"
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    model = build_model(k)
    n_total = WALKERS_PER_GPU * world
    sh = ShardedAIS(model, n_total, seed=SEED, device=dev).init()
    ens = sh.engine.ens

    def sync():
        torch.cuda.synchronize(dev)
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed_region(nt_r, steps, warm):
        for _ in range(warm):
            sh.generation(nt_r)
        sync()
        s0 = sh.global_stats()
        # hipEvent pairs on the kernel's stream, one pair per 8 consecutive half-generation
        # launches (a pair per launch adds ~3 us of marker overhead to every figure); with
        # collectives between the launches each launch gets its own pair instead
        ens.set_timing(2 * steps, stride=1 if use_pg else 8)
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            sh.generation(nt_r)
        sync()
        el_r = time.perf_counter() - t0
        kms_r, nl_r = ens.kernel_ms()
        ens.set_timing(0)
        s1 = sh.global_stats()
        tmax = torch.tensor([el_r], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if use_pg:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item()), kms_r, nl_r, s0, s1

    el, kms, nl, st0, st1 = timed_region(nt, args.steps, args.warmup)
    # secondary figure at SURVEY §8d's ntransitions = 16 (launch prologue, tail and -- for
    # N > 1 -- the exchange weigh 6x more per evaluation); reported beside the headline
    alt = None
    if nt != NT_ALT and not args.no_alt:
        k2 = max(20, args.steps)
        el2, kms2, nl2, a0, a1 = timed_region(NT_ALT, k2, 5)
        alt = (el2, kms2, nl2, a1["proposals"] - a0["proposals"], k2)

    # RCCL prints a version banner to the C stdout of every rank when its communicator
    # comes up; push it out now so that the JSON below is the LAST line of the job
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if use_pg:
        dist.barrier()
    if rank == 0:
        proposals = st1["proposals"] - st0["proposals"]
        cost_evals = st1["cost_evals"] - st0["cost_evals"]
        assert proposals == n_total * nt * args.steps, (proposals, n_total * nt * args.steps)
        bytes_per_eval = 8 * (3 * D + 4)
        rows = WALKERS_PER_GPU // 2
        alg_bytes_launch = rows * nt * bytes_per_eval
        achieved = alg_bytes_launch / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
        # HBM bytes per launch from PMC counters are collected in separate rocprofv3
        # passes (FETCH_SIZE / WRITE_SIZE cannot share a pass); the committed summary
        # of that run is reported here when it was taken on this very workload.
        traffic = None
        tf = os.path.join(ROOT, "profiles", f"r01_pmc_traffic_nt{nt}.json")
        if os.path.exists(tf):
            traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
        out = {
            "metric": "walker proposal+cost evals/sec at N=65536 walkers, D=8",
            "value": proposals / el, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "AIS C3: 65536 walkers/GPU x 8-param Rosenbrock-like cost, "
                                   "Uniform(-5,5)^8 prior, kernelized scale 1.0",
                       "walkers_per_gpu": WALKERS_PER_GPU, "walkers_total": n_total, "D": D,
                       "ntransitions": nt, "evals_per_step": n_total * nt, "seed": SEED,
                       "parallelism": f"walker-sharded x{world}, 1 all-gather per half-generation"
                       if world > 1 else "single GPU"},
            "cost_evals_per_s": cost_evals / el,
            "accept_rate": (st1["accepted"] - st0["accepted"]) / max(1, proposals),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "ais_half_kernel<8, rosenbrock>",
                         "kernel_avg_ms": kms, "kernel_launches_timed": nl,
                         "algorithmic_bytes_per_launch": alg_bytes_launch},
        }
        if alt is not None:
            el2, kms2, nl2, prop2, k2 = alt
            ach2 = rows * NT_ALT * bytes_per_eval / (kms2 * 1e-3) / 1e9 if kms2 > 0 else 0.0
            out[f"also_at_ntransitions_{NT_ALT}"] = {
                "value": prop2 / el2, "unit": "evals/s", "steps": k2, "ms_per_step": el2 / k2 * 1e3,
                "kernel_avg_ms": kms2, "roofline_achieved_GBps": ach2,
                "roofline_frac": ach2 / HBM_PEAK_GBS,
                "why": "SURVEY 8d times C3 at ntransitions = 16 as well"}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
