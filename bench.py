#!/usr/bin/env python3
"""bench.py -- walker proposal+cost evaluations per second of the AIS hot path.

Workload (BASELINE.json configs[2]): AIS, 65 536 walkers per GPU,
D = 8, prior Factored(Uniform(-5,5))^8, cost sqrt(sum 100(x[k+1]-x[k]^2)^2 +
(1-x[k])^2), ApproxKernelizedPosterior scale 1.0, seed 1.
A "step" is one GENERATION: every walker receives `ntransitions` transition!()
calls (two half-generation kernel launches per GPU; for N>1 one RCCL all-gather
after each, issued by the library itself -- include/kabc.h "multi-GPU" -- so this
script needs neither torch nor torch.distributed).  Weak scaling: per-GPU walkers
are fixed, N_total = 65 536 * gpus.

`ntransitions`: SURVEY §8d prescribes {1, 16}; the reference's own examples sample
with 100 (README.md:57, BASELINE.json configs[0]).  All three are timed in every
run and reported under `by_ntransitions`; the headline (`value`, `roofline`) is the
one --ntransitions selects (default 100) and says so in `config`.

Timed region: W warm-up steps, then blocks of EXACTLY K steps, each block bracketed
by a barrier + stream synchronisation on both sides; blocks are repeated until at
least --min-seconds of device work has been timed (so that an external sampler sees
the GPU busy); `ms_per_step` is the mean over all timed steps.

Prints ONE JSON line (rank 0).  `roofline` is for the half-generation kernel:
algorithmic bytes per launch = rows * ntransitions * 8(3D+4) over the
kernel's average duration measured with hipEvents on its stream over the timed
region; `roofline.valu` is the second, binding resource: VALU wave-instructions per
launch (committed PMC summary, profiles/) x the kernel's mix-weighted issue cycles
(profiles/r*_valu_mix.json) / (1024 SIMDs x 2.4 GHz x kernel time).  `smc_c4` times BASELINE.json configs[3] (smc, 32 768 particles, D = 16) the
same way.  `next_rows` are wall-clock legs of SURVEY 8's f3 / f4 rows (ABCDE, pfilter) beside the
oracle on the same runs.  `cpu_baseline` times the CPU oracle's faithful serial restatement of the
reference on a bounded sample of the same workloads.
"""
import argparse
import glob
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# this host never imports torch: keep the process on the system ROCm runtime alone
os.environ.setdefault("KABC_NO_TORCH_PRELOAD", "1")

WALKERS_PER_GPU = 65536
D = 8
NT_SET = (1, 16, 100)
NT_HEADLINE = 100
SEED = 1
HBM_PEAK_GBS = 8000.0   # MI355X spec peak, /opt/skills/guides/MI355X_MICROARCH.md
SIMDS = 1024            # 256 CUs x 4
CLOCK_HZ = 2.4e9        # max engine clock, same guide
SMC_N, SMC_D = 32768, 16


def build_model(k):
    prior = k.Factored(*[k.Uniform(-5, 5)] * D)
    return k.ApproxKernelizedPosterior(prior, k.costs.Rosenbrock(), 1.0)


def c4_problem(k):
    """theta = (m, s, z_1..z_14), sim ybar_g = m + s z_g + randn/sqrt(8)."""
    import numpy as np
    rng = np.random.default_rng(1)
    zstar = rng.normal(size=14)
    ybar = 1.0 + 0.5 * zstar + rng.normal(size=14) / np.sqrt(8)
    prior = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 14)
    return prior, k.costs.HierGaussSim(ybar), dict(nparticles=SMC_N, alpha=0.95, epstol=0.05, seed=1)


def readme_problem(k):
    """BASELINE.json configs[0] = README.md:31-57: tdata = 1000 draws N(2, 0.04); prior
    Factored(Uniform(1,3), Truncated(Normal(0,0.1),0,100)); cost = hypot of the mean / 50 x std
    differences of 1000 simulated draws; ApproxKernelizedPosterior(..., 0.005); AIS(10),
    1000 samples, ntransitions = 100."""
    import numpy as np
    tdata = np.random.default_rng(0).normal(2.0, 0.04, 1000)
    prior = k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100))
    cost = k.costs.NormalMeanStdSim(1000, tdata.mean(), tdata.std(ddof=1))
    return k.ApproxKernelizedPosterior(prior, cost, 0.005)


def c2_problem(k):
    """BASELINE.json configs[1] = SURVEY 8d C2: N 4096, D 2, Normal(0,5)^2, ||x - (1,-0.5)||, scale 0.1"""
    prior = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    return k.ApproxKernelizedPosterior(prior, k.costs.GaussDist([1.0, -0.5]), 0.1)


def prior_class_problems(k):
    """(name, model, N, D): the prior classes the reference's own tests use beside boxes
    (test/runtests.jl:50-53,78,107,241) at the bench size."""
    import numpy as np
    socks = k.Factored(k.NegativeBinomial(900 / 195, (900 / 195) / (30 + 900 / 195)), k.Beta(15, 2))
    hier = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 14)
    ybar = np.random.default_rng(1).normal(size=14)
    four = k.Factored(k.Gamma(2.5, 0.7), k.LogNormal(0.3, 0.6), k.Beta(2, 3), k.Normal(0, 1))
    return [
        ("normal8_gauss_dist", k.ApproxKernelizedPosterior(
            k.Factored(*[k.Normal(0, 5)] * 8), k.costs.GaussDist(np.zeros(8)), 1.0), 65536, 8),
        ("socks_negbin_beta", k.ApproxKernelizedPosterior(socks, k.costs.GaussDist([40.0, 0.8]), 3.0),
         65536, 2),
        ("four_family_d4", k.ApproxKernelizedPosterior(four, k.costs.NormShell(2.0), 0.5), 65536, 4),
        ("c4_hier_prior_sim", k.ApproxKernelizedPosterior(hier, k.costs.HierGaussSim(ybar), 0.3),
         32768, 16),
    ]


class _env:
    """set an environment variable for a block and put back what was there (or remove it)"""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.prev = os.environ.get(self.name)
        if self.value is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.value

    def __exit__(self, *exc):
        if self.prev is None:
            os.environ.pop(self.name, None)
        else:
            os.environ[self.name] = self.prev


# sample() calls shaped like the reference's own tests and examples (test/runtests.jl:82-131,177-198,
# examples/example_n1.jl:40; src/KissABC.jl:71: ntransitions = 1 is the default): (walkers, samples,
# discard_initial, ntransitions) on the C2 model
REFERENCE_SHAPES = [(12, 500, 1000, 1), (100, 1000, 10000, 1), (50, 100, 50000, 1), (20, 100, 2000, 40),
                    (10, 10000, 0, 50)]


def _shape_key(w, ns, disc, nt):
    return f"AIS({w}) samples={ns} discard_initial={disc} ntransitions={nt}"


def _cpu_chain(args):
    """One independent serial chain (the MCMCThreads analogue, src/KissABC.jl:108)."""
    nwalkers, seed, budget_s = args
    import kissabc_jl_amd as k
    from oracle import oracle as orc
    o = orc.OracleAIS(build_model(k), nwalkers, seed=seed).init()
    o.steps_serial(1024, NT_HEADLINE, collect=False)  # warm-up
    nsteps, done, t0 = 4096, 0, time.perf_counter()
    while True:
        o.steps_serial(nsteps, NT_HEADLINE, collect=False)
        done += nsteps
        el = time.perf_counter() - t0
        if el > budget_s:
            return done, el


def cpu_baseline(k, budget_s, with_smc):
    """Oracle, serial reference schedule (src/KissABC.jl:66-80), same model, bounded
    sample: (i) 1 core, N = 65 536 walkers, as many step() calls as fit the budget;
    (ii) all host cores as independent chains of N/cores walkers each; (iii) the smc
    restatement (src/smc.jl:92-206) on C4, one full run on 1 core."""
    import multiprocessing as mp
    done, el = _cpu_chain((WALKERS_PER_GPU, SEED, budget_s))
    out = {
        "value": done * NT_HEADLINE / el, "unit": "evals/s", "cores": 1, "kind": "port",
        "sample": f"oracle ref_serial (C restatement of src/transition.jl + src/KissABC.jl:66-80), "
                  f"N={WALKERS_PER_GPU} D={D} rosenbrock, {done} step() calls x "
                  f"ntransitions={NT_HEADLINE} in {el:.1f}s on 1 host core",
    }
    try:
        cores = len(os.sched_getaffinity(0))
        if cores > 1:
            per = max(D + 5, WALKERS_PER_GPU // cores)
            with mp.get_context("spawn").Pool(cores) as pool:
                res = pool.map(_cpu_chain, [(per, SEED + 1 + c, budget_s / 2) for c in range(cores)])
            out["all_cores"] = {
                "value": sum(d * NT_HEADLINE / e for d, e in res), "cores": cores, "unit": "evals/s",
                "sample": f"{cores} independent chains of {per} walkers (MCMCThreads analogue), "
                          f"{budget_s / 2:.0f}s each"}
    except Exception as e:  # the baseline is informational; never fail the bench on it
        out["all_cores"] = {"error": repr(e)}
    # the reference itself, if a julia with KissABC exists on this box
    try:
        import shutil
        import subprocess
        jl = shutil.which("julia")
        if jl:
            r = subprocess.run([jl, "-t", "1", os.path.join(ROOT, "bench", "cpu_reference.jl"),
                                str(WALKERS_PER_GPU), str(D), str(NT_HEADLINE), str(budget_s)],
                               capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out["julia_reference"] = json.loads(line[-1]) if line else {"error": r.stderr[-500:]}
        else:
            out["julia_reference"] = None   # no julia on this box (none in the build image either)
    except Exception as e:
        out["julia_reference"] = {"error": repr(e)}
    # the reference's own workloads (BASELINE.json configs[0], [1]) on one host core
    try:
        from oracle import oracle as orc
        t0 = time.perf_counter()
        o = orc.OracleAIS(readme_problem(k), 10, seed=1).init()
        o.steps_serial(1000, NT_HEADLINE)
        w = time.perf_counter() - t0
        out["readme_c1"] = {"wall_s": w, "transitions_per_s": 1000 * NT_HEADLINE / w, "cores": 1,
                            "kind": "port", "sample": "oracle ref_serial: AIS(10), 1000 step() calls x "
                                                      "ntransitions=100, 1000 normals per cost (README.md:31-57)"}
        t0 = time.perf_counter()
        rs = orc.smc(readme_problem(k).prior, readme_problem(k).cost, nparticles=100, seed=1)
        w = time.perf_counter() - t0
        out["readme_smc"] = {"wall_s": w, "iterations": rs["iterations"], "cores": 1, "kind": "port",
                             "sample": "oracle smc (src/smc.jl:92-206) with its defaults (100 particles) on the "
                                       "README simulator (README.md:80-84)"}
        o = orc.OracleAIS(c2_problem(k), 4096, seed=1).init()
        o.steps_serial(4096, NT_HEADLINE, collect=False)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 2.0:
            o.steps_serial(8192, NT_HEADLINE, collect=False)
            n += 8192
        w = time.perf_counter() - t0
        out["c2"] = {"value": n * NT_HEADLINE / w, "unit": "evals/s", "cores": 1, "kind": "port",
                     "sample": f"oracle ref_serial: C2 (4096 walkers, D=2), {n} step() calls x "
                               f"ntransitions=100 in {w:.1f}s"}
    except Exception as e:
        out["readme_c1"] = {"error": repr(e)}
    try:  # the reference's own sample() shapes: the same number of step() calls, serial schedule, one core
        from oracle import oracle as orc
        rs = {}
        for w_, ns_, disc_, nt_ in REFERENCE_SHAPES:
            gens = -(-disc_ // w_) + max(1, -(-ns_ // w_))
            ws = []
            for _ in range(3):
                t0 = time.perf_counter()
                o = orc.OracleAIS(c2_problem(k), w_, seed=1).init()
                o.steps_serial(gens * w_, nt_, collect=True)   # one device generation = N step() calls
                ws.append(time.perf_counter() - t0)
            rs[_shape_key(w_, ns_, disc_, nt_)] = {"wall_ms": sorted(ws)[1] * 1e3, "step_calls": gens * w_}
        out["reference_shapes"] = {"cores": 1, "kind": "port", "shapes": rs,
                                   "sample": "oracle ref_serial: init + ceil(discard/N)*N + ceil(Ns/N)*N step() "
                                             "calls on the C2 model (median of 3)"}
    except Exception as e:
        out["reference_shapes"] = {"error": repr(e)}
    try:  # the "next" rows (ABCDE, pfilter): the oracle on the same runs as the device legs
        from oracle import oracle as orc
        n2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
        gd = k.costs.GaussDist([1.0, -0.5])
        t0 = time.perf_counter()
        orc.abcde(n2, gd, 0.01, nparticles=2000, generations=50, seed=3)
        out["abcde"] = {"wall_s": time.perf_counter() - t0, "cores": 1, "kind": "port",
                        "sample": "oracle ABCDE (src/smc.jl:347-430), 2000 particles, 50 generations"}
        t0 = time.perf_counter()
        orc.pfilter(n2, gd, 16384, q=0.7, eff_tol=0.1, epstol=0.02, seed=3)
        out["pfilter"] = {"wall_s": time.perf_counter() - t0, "cores": 1, "kind": "port",
                          "sample": "oracle pfilter (src/smc.jl:275-340), 16384 particles"}
    except Exception as e:
        out["abcde"] = {"error": repr(e)}
    if with_smc:
        try:
            from oracle import oracle as orc
            prior, cost, kw = c4_problem(k)
            t0 = time.perf_counter()
            ro = orc.smc(prior, cost, **kw)
            w = time.perf_counter() - t0
            out["smc_c4"] = {"wall_s": w, "particle_updates_per_s": ro["proposals"] / w,
                             "iterations": ro["iterations"], "eps": ro["eps"], "cores": 1,
                             "kind": "port",
                             "sample": "oracle smc (C restatement of src/smc.jl:92-206), one full "
                                       "C4 run on 1 host core"}
        except Exception as e:
            out["smc_c4"] = {"error": repr(e)}
    return out


def _valu_cycles():
    """average issue cycles per VALU wave-instruction of the headline kernel: its static class mix
    (profiles/r*_valu_mix.json) x the measured issue time per class"""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_mix.json")))
    if not files:
        return 4.0, None
    try:
        m = json.load(open(files[-1]))
        return float(m["avg_issue_cycles"]), os.path.basename(files[-1])
    except Exception:
        return 4.0, None


VALU_CYCLES, VALU_MIX_FILE = _valu_cycles()


def _pmc_table():
    """VALU wave-instructions per half-generation launch from the newest committed PMC
    summary (profiles/r*_pmc_insts.json: {"<nt>": {"SQ_INSTS_VALU": per-launch mean, ...}})."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_insts.json")))
    if not files:
        return None, {}
    try:
        return os.path.basename(files[-1]), json.load(open(files[-1]))
    except Exception:
        return None, {}


def latency_floor_inputs():
    """The two inputs of smc_c4.latency_floor, measured in THIS run by child processes (before this process touches
    the GPU): (i) the XCD-aware device-wide barrier at 128 workgroups -- tools/xcd_barrier_probe.hip (rows written,
    barrier, random rows of other workgroups read and checked), built on first use; (ii) the propose / accept pass's
    dependent chain from the loop kernel's phase stamps -- the PROBES build of the library (make PROBES=1), when it
    is there.  Anything that cannot be measured here falls back to the newest committed profiles/ file and says so."""
    import re
    import subprocess
    out = {}
    try:
        exe = os.path.join(ROOT, "tools", "_bin", "xcdbar")
        if not os.path.exists(exe):
            os.makedirs(os.path.dirname(exe), exist_ok=True)
            subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "--offload-arch=gfx950",
                            os.path.join(ROOT, "tools", "xcd_barrier_probe.hip"), "-o", exe],
                           check=True, capture_output=True, timeout=300)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")][-1]
        j = json.loads(line)
        # (the kernel's barrier: release once per XCD, buffer_inv sc1 per workgroup = the probe's "xcd_release_only")
        if j.get("xcd_release_only_errors_128", 1) == 0:
            out["barrier_128_us"] = float(j["xcd_release_only_128"])
            out["barrier_source"] = "measured in this run (tools/xcd_barrier_probe.hip, xcd_release_only_128)"
    except Exception as e:
        out["barrier_error"] = repr(e)[:200]
    if "barrier_128_us" not in out:
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_xcd_barrier.json")))[-1:]:
            out["barrier_128_us"] = float(json.load(open(f))["xcd_release_only_128"])
            out["barrier_source"] = os.path.basename(f) + " (xcd_release_only_128)"
    try:
        probes = os.path.join(ROOT, "kissabc.jl_amd", "lib", "libkabc_hip_probes.so")
        if os.path.exists(probes):
            env = dict(os.environ, KABC_PROBES="1", KABC_SMC_STAMPS="1", KABC_SPECIALIZE="0")
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "smc_c4_probe.py")], env=env,
                               capture_output=True, text=True, timeout=300)
            m = None
            for ln in r.stderr.splitlines():
                mm = re.search(r"mcmc split: philox\+select (\d+) issue\+pre (\d+) wait (\d+) logpdf (\d+) cost\+accept (\d+) tail (\d+)", ln)
                if mm:
                    m = mm
            if m:
                out["pass_chain_us"] = sum(int(v) for v in m.groups()) / 100.0   # 10 ns ticks
                out["pass_chain_source"] = "measured in this run (phase stamps of libkabc_hip_probes.so, tools/smc_c4_probe.py)"
    except Exception as e:
        out["pass_chain_error"] = repr(e)[:200]
    if "pass_chain_us" not in out:
        out["pass_chain_us"] = 7.2
        out["pass_chain_source"] = "profiles/r04_smc_c4.txt (phase stamps; no PROBES build of the library in this tree)"
    return out


def smc_sharded_world1():
    """kabc_smc_run_dist_mode on an RCCL communicator of world size 1 (all a one-GPU box offers), in a child
    process with its own rendezvous (tools/smc_dist_probe.py): wall per iteration of C4's model at 131 072
    particles for kabc_smc_run, the sharded cost loop and the sharded particles, with the collectives and host
    looks each made (kabc_smc_dist_stats).  The floor a multi-GPU run starts from, not a scaling figure."""
    import socket
    import subprocess
    try:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                   LOCAL_RANK="0", KABC_NO_TORCH_PRELOAD="1")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "smc_dist_probe.py"), "131072"], env=env,
                           capture_output=True, text=True, timeout=240)
        line = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")][-1]
        return json.loads(line)
    except Exception as e:
        return {"error": repr(e)[:200]}


def spawn_ranks(n, argv, stub=None, timeout=3600.0):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per
    GPU, BEFORE this process has made any GPU call (the children are fresh interpreters; nothing
    is exec'ed after a GPU initialisation).  The ranks get the variables a launcher would set
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT); their stderr passes through,
    rank 0's stdout is relayed with its JSON line LAST, and the exit code is non-zero when any
    rank failed (the others are ended: a rank that lost its peers would wait in a collective).
    Returns (exit code, the JSON line or None)."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:   # a free rendezvous port (it names the unique-id file, comm.py)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import tempfile
    out_f = tempfile.TemporaryFile(mode="w+")   # (a pipe would fill up: the line is tens of KB)
    procs = []

    def end_children():
        """every rank that still runs: SIGTERM to its process group, SIGKILL after 10 s"""
        alive = [p for p in procs if p.poll() is None]
        for p in alive:
            try:
                os.killpg(p.pid, signal.SIGTERM)   # (each rank leads a session of its own: start_new_session)
            except OSError:
                pass
        for p in alive:
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except OSError:
                    pass

    def on_signal(signum, frame):   # the launcher is told to stop: so are its ranks (a peer in a collective never exits)
        raise KeyboardInterrupt(f"signal {signum}")

    old_handlers = {}
    for sg in (signal.SIGTERM, signal.SIGINT):
        try:
            old_handlers[sg] = signal.signal(sg, on_signal)
        except ValueError:   # (not the main thread: tests)
            pass
    t0, rc = time.time(), 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            if stub is not None:
                env["KABC_BENCH_STUB_RANK"] = stub
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=out_f if r == 0 else subprocess.DEVNULL, start_new_session=True))
        live = set(range(n))
        while live:
            for r in sorted(live):
                c = procs[r].poll()
                if c is not None:
                    live.discard(r)
                    if c != 0:
                        rc = rc or c
                        print(f"[bench] rank {r} exited with code {c}", file=sys.stderr, flush=True)
            if (rc or (timeout and time.time() - t0 > timeout)) and live:
                rc = rc or 124
                end_children()
                live.clear()
            if live:
                time.sleep(0.05)
    except KeyboardInterrupt as e:
        print(f"[bench] launcher interrupted ({e}): ending the ranks", file=sys.stderr, flush=True)
        rc = rc or 130
    finally:
        end_children()   # (no rank outlives the launcher, whatever ended the loop)
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    out_f.seek(0)
    out0 = out_f.read()
    out_f.close()
    line = None
    lines = [ln for ln in (out0 or "").splitlines() if ln.strip()]
    for ln in lines:
        if ln.lstrip().startswith("{"):
            try:
                json.loads(ln)
                line = ln
            except ValueError:
                pass
    for ln in lines:
        if ln is not line:
            print(ln, flush=True)
    if line is None and rc == 0:
        print("[bench] rank 0 printed no JSON line", file=sys.stderr, flush=True)
        rc = 1
    if line is not None and rc == 0:
        print(line, flush=True)
    return rc, line


def stub_rank(mode, rank, world, args):
    """tests/test_bench_spawn.py: a rank that stands in for the GPU work (no library call)"""
    toks = mode.split(",")   # "ok" | "fail<r>" | "hang<r>", comma-separated
    if f"hang{rank}" in toks:
        time.sleep(600)
    if f"fail{rank}" in toks:
        time.sleep(0.5)
        print(f"stub rank {rank} fails", file=sys.stderr, flush=True)
        raise SystemExit(7)
    time.sleep(0.2 * rank)
    if rank == 0:
        print("stub: a line that is not the result")
        print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ranks_env": [os.environ.get(v) for v in
                                                               ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR")]}),
              flush=True)
    raise SystemExit(0)


def cold_spec_child():
    """`bench.py --cold-spec`: the default (non-blocking) specialisation with an EMPTY code-object
    cache, in a process of its own: first-call latency of k.AisEnsemble(...) against the same call
    with KABC_SPECIALIZE=0, the time until the handle switched to the model's own kernels, and the
    launches that ran before.  One JSON line."""
    import tempfile
    d = tempfile.mkdtemp(prefix="kabc_cold_rtc_")
    os.environ["KABC_RTC_CACHE_DIR"] = d
    os.environ["KABC_RTC_WORKERS"] = "4"
    os.environ.pop("KABC_SPECIALIZE", None)
    import kissabc_jl_amd as k
    ctx = k.Context(int(os.environ.get("KABC_BENCH_DEVICE", "0")))
    probs = prior_class_problems(k)
    k.AisEnsemble(build_model(k), 4096, seed=SEED, ctx=ctx).init().advance(1, 4)   # context, module load
    ctx.synchronize()
    res, live = {}, {}

    def first_call(m, Nm):
        t0 = time.perf_counter()
        e = k.AisEnsemble(m, Nm, seed=SEED, ctx=ctx).init()
        e.advance(1, NT_HEADLINE)
        ctx.synchronize()
        return e, (time.perf_counter() - t0) * 1e3
    for name, m, Nm, Dm in probs:
        with _env("KABC_SPECIALIZE", "0"):
            first_call(m, Nm)[0].close()     # (the prebuilt kernel's code object is loaded by its first launch)
            e0, ms0 = first_call(m, Nm)
            e0.close()
        t_create = time.perf_counter()
        e, ms1 = first_call(m, Nm)
        res[name] = {"first_call_ms_prebuilt_only": ms0, "first_call_ms_default": ms1,
                     "state_after_first_call": e.spec_state()[0]}
        live[name] = (e, t_create)
    t0 = time.perf_counter()
    while live and time.perf_counter() - t0 < 300:
        for name in list(live):
            e, t_create = live[name]
            e.advance(1, NT_HEADLINE)
            st, before = e.spec_state()
            if st != "pending":
                ctx.synchronize()
                res[name].update(cold_compile_s=time.perf_counter() - t_create, state=st,
                                 switched_after_launches=before)
                e.close()
                del live[name]
        time.sleep(0.01)
    for name in live:
        res[name].update(state="pending after 300 s")
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    print(json.dumps({"cold_spec": res}), flush=True)


def main():
    if "--cold-spec" in sys.argv:
        return cold_spec_child()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ntransitions", type=int, default=NT_HEADLINE,
                    help="the setting the headline value is quoted on")
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="repeat the K-step block until this much has been timed (per setting)")
    ap.add_argument("--headline-seconds", type=float, default=7.0,
                    help="timed device work of the headline setting (one uninterrupted stretch)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true",
                    help="time the headline ntransitions only (clean rocprof summaries)")
    ap.add_argument("--no-smc", action="store_true", help="skip the C4 smc leg")
    ap.add_argument("--no-cold-spec", action="store_true",
                    help="skip the cold-cache measurement of the default specialisation path (a child process)")
    args = ap.parse_args()
    nt_head = args.ntransitions

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    emulate = int(os.environ.get("KABC_BENCH_EMULATE_RANKS", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not emulate:
        # no launcher: this process becomes the launcher (it has made no GPU call)
        rc, _ = spawn_ranks(args.gpus, sys.argv[1:], stub=os.environ.get("KABC_BENCH_STUB_RANK"))
        raise SystemExit(rc)
    if os.environ.get("KABC_BENCH_STUB_RANK"):
        stub_rank(os.environ["KABC_BENCH_STUB_RANK"], rank, world, args)
    if world != args.gpus and not emulate:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` "
                         "(the ranks are started here) or `python -m torch.distributed.run "
                         "--nproc-per-node N bench.py --gpus N`")

    # The CPU baseline runs FIRST, before this process touches the GPU: its all-core
    # leg spawns worker processes, and a process that has initialised HIP must not
    # fork+exec on this pool.
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        import kissabc_jl_amd as k0
        cpu = cpu_baseline(k0, args.cpu_seconds, with_smc=not args.no_smc)

    floor_in = None
    sharded1 = None
    if world == 1 and not emulate and not args.no_smc:
        floor_in = latency_floor_inputs()
        if not args.no_alt:
            sharded1 = smc_sharded_world1()

    # ... and so does the cold-cache leg of the default specialisation path (its own process: the
    # code-object cache must be empty and no unit loaded)
    cold_spec = None
    if world == 1 and not emulate and not args.no_alt and not args.no_cold_spec:
        import subprocess
        try:
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cold-spec"],
                                capture_output=True, text=True, timeout=420)
            for ln in cp.stdout.splitlines():
                if ln.startswith("{"):
                    cold_spec = json.loads(ln).get("cold_spec")
            if cold_spec is None:
                cold_spec = {"error": (cp.stderr or "no output")[-500:]}
        except Exception as e:   # informational: never fail the bench on it
            cold_spec = {"error": repr(e)}

    import kissabc_jl_amd as k
    from kissabc_jl_amd.comm import Comm, EnsembleGroup

    dev = int(os.environ.get("KABC_BENCH_DEVICE", local_rank))
    use_comm = (world > 1 or os.environ.get("KABC_FORCE_COLLECTIVE") == "1") and not emulate
    comm = Comm.from_env(device=dev) if use_comm else None

    model = build_model(k)
    if emulate:
        # Rehearsal of the world-N line on ONE GPU: N emulated ranks on device `dev` in this process,
        # a real exchange between their shards through the P2P communicator (kabc_comm_init_all with
        # a repeated device id).  Says so in the line; no scaling figure can be read from it.
        world = emulate

        class _Group:
            def __init__(self):
                self.g = EnsembleGroup(model, WALKERS_PER_GPU * emulate, seed=SEED, devices=[dev] * emulate,
                                       backend="p2p").init()

            def advance(self, gens, nt):
                self.g.advance(gens, nt)

            def stats(self):
                return self.g.stats()

            def set_timing(self, n, stride=1):
                for sh in self.g.shards:
                    sh.set_timing(n, stride=stride)

            def kernel_ms(self):
                v = [sh.kernel_ms() for sh in self.g.shards]
                return sum(a for a, _ in v) / len(v), sum(b for _, b in v)

            def exchange_us(self):
                xs = [sh.exchange_us() for sh in self.g.shards]
                return {kk: max(x[kk] for x in xs) for kk in xs[0]}

            def synchronize(self):
                self.g.synchronize()

            def close(self):
                self.g.close()
        ens = _Group()
        ctx = ens
    else:
        ctx = comm.ctx if comm else k.Context(dev)
    n_total = WALKERS_PER_GPU * world
    if not emulate:
        ens = k.AisEnsemble(model, n_total, seed=SEED, ctx=ctx, comm=comm).init()

    def sync():
        ctx.synchronize()
        if comm:
            comm.barrier()   # an all-reduce over the ranks + a stream synchronisation
            ctx.synchronize()

    def global_stats():
        st = ens.stats()
        v = [st["proposals"], st["cost_evals"], st["accepted"]]
        if comm:
            v = comm.allreduce_sum(v)
        return dict(zip(("proposals", "cost_evals", "accepted"), v))

    def timed_region(nt_r, steps, warm):
        ens.advance(warm, nt_r)
        sync()
        t0 = time.perf_counter()   # calibration block (not reported): how long K steps take
        ens.advance(steps, nt_r)
        sync()
        blocks = max(1, int(math.ceil(args.min_seconds / max(time.perf_counter() - t0, 1e-9))))
        if comm:   # every rank must run the same number of blocks
            blocks = int(comm.allreduce_max([float(blocks)])[0])
        s0 = global_stats()
        # hipEvent pairs on the kernel's stream, one pair per 8 consecutive half-generation
        # launches (a pair per launch adds ~3 us of marker overhead to every figure); with
        # collectives between the launches each launch gets its own pair instead
        ens.set_timing(min(2 * steps * blocks, 8192), stride=1 if (comm or emulate) else 8)
        el_r = 0.0
        for _ in range(blocks):
            sync()
            t0 = time.perf_counter()
            ens.advance(steps, nt_r)
            sync()
            el_r += time.perf_counter() - t0
        kms_r, nl_r = ens.kernel_ms()
        # exchange diagnostics (kabc_ais_exchange_us: hipEvent pairs on the stream the all-gather
        # runs on, over the first 128 timed half-generations): max over the ranks
        xch = None
        if comm or emulate:
            x = ens.exchange_us()
            v = [x["compute_us_per_half"], x["exchange_us_per_half"], x["exposed_us_per_half"]]
            if comm:
                v = comm.allreduce_max(v)
            xch = {"compute_us_per_half": v[0], "exchange_us_per_half": v[1], "exposed_us_per_half": v[2],
                   "chunks": x["chunks"],
                   "note": "per half-generation, max over ranks: kernels / all-gather(s) on their stream / "
                           "what the context stream then waits for the gathered half"}
        ens.set_timing(0)
        s1 = global_stats()
        if comm:
            el_r = comm.allreduce_max([el_r])[0]
        return {"el": el_r, "kms": kms_r, "nl": nl_r, "blocks": blocks, "steps": steps, "exchange": xch,
                "proposals": s1["proposals"] - s0["proposals"],
                "cost_evals": s1["cost_evals"] - s0["cost_evals"],
                "accepted": s1["accepted"] - s0["accepted"]}

    settings = [nt_head] if args.no_alt else sorted(set(NT_SET) | {nt_head})
    regions = {}
    base_min = args.min_seconds
    for nt_r in settings:
        # the same K everywhere would make the nt = 1 block 100x shorter; the block count
        # (min-seconds) evens the timed duration out instead.  The headline setting runs
        # --headline-seconds without a gap, so that an external sampler with a period of several
        # seconds sees the GPU busy at least once.
        args.min_seconds = max(base_min, args.headline_seconds) if nt_r == nt_head else base_min
        regions[nt_r] = timed_region(nt_r, args.steps, args.warmup if nt_r == nt_head else 5)
    args.min_seconds = base_min

    def kernel_leg(model, N, Dm, nts=(NT_HEADLINE,)):
        """half-generation kernel time and evals/s of another single-GPU workload"""
        e = k.AisEnsemble(model, N, seed=SEED, ctx=ctx).init()
        res = {}
        for nt_r in nts:
            gens = max(8, min(400, int(3e8 / (N * nt_r))))
            e.advance(3, nt_r)
            ctx.synchronize()
            e.set_timing(2 * gens, stride=8)
            t0 = time.perf_counter()
            e.advance(gens, nt_r)
            ctx.synchronize()
            el = time.perf_counter() - t0
            kms, nl = e.kernel_ms()
            e.set_timing(0)
            Bm = 8 * (3 * Dm + 4)
            res[str(nt_r)] = {"kernel_avg_us": kms * 1e3, "evals_per_s_kernel": (N // 2) * nt_r / (kms * 1e-3),
                              "evals_per_s_wall": N * nt_r * gens / el, "bytes_per_eval": Bm,
                              "roofline_frac": (N // 2) * nt_r * Bm / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        st = e.stats()
        res["spec_state"], res["switched_after_launches"] = e.spec_state()
        e.close()
        res["accept_rate"] = st["accepted"] / max(1, st["proposals"])
        return res

    extra = {}
    if world == 1 and not args.no_alt and isinstance(ctx, k.Context):
        # BASELINE.json configs[0]: the README example end to end through sample()
        rm = readme_problem(k)
        k.sample(rm, k.AIS(10), 1000, ntransitions=NT_HEADLINE, seed=1, ctx=ctx, return_array=True)
        walls = []
        for _ in range(5):
            t0 = time.perf_counter()
            res = k.sample(rm, k.AIS(10), 1000, ntransitions=NT_HEADLINE, seed=1, ctx=ctx, return_array=True)
            walls.append(time.perf_counter() - t0)
        w = sorted(walls)[len(walls) // 2]
        extra["readme_c1"] = {
            "workload": "README.md:31-57 (BASELINE.json configs[0]): AIS(10), 1000 samples, "
                        "ntransitions=100, 1000 normals per cost evaluation, through sample()",
            "wall_ms": w * 1e3, "transitions_per_s": 1000 * NT_HEADLINE / w,
            "posterior_mean": res.mean(0).tolist(),
            "reference_documented": "2.0 +- 0.018, 0.0395 +- 0.00093, ~2 s incl. JIT (README.md:57-66)"}
        if cpu and isinstance(cpu.get("readme_c1"), dict) and "wall_s" in cpu["readme_c1"]:
            extra["readme_c1"]["cpu_baseline"] = cpu["readme_c1"]
            extra["readme_c1"]["vs_cpu_port_1core"] = cpu["readme_c1"]["wall_s"] / w
        # the same model as 50 chains at once (sample(model, AIS(10), MCMCThreads(), 1000, 50):
        # test/runtests.jl:88-104 runs 50 chains of AIS(12)): chains are a grid dimension of every
        # launch, so a single chain's latency is shared by all of them
        try:
            k.sample(rm, k.AIS(10), k.MCMCThreads(), 1000, 50, ntransitions=NT_HEADLINE, seed=1, ctx=ctx,
                     return_array=True)
            walls = []
            for _ in range(3):
                t0 = time.perf_counter()
                rc50 = k.sample(rm, k.AIS(10), k.MCMCThreads(), 1000, 50, ntransitions=NT_HEADLINE, seed=1,
                                ctx=ctx, return_array=True)
                walls.append(time.perf_counter() - t0)
            w50 = sorted(walls)[1]
            extra["readme_c1"]["chains_50"] = {
                "workload": "sample(model, AIS(10), MCMCThreads(), 1000, 50; ntransitions=100): 50 independent "
                            "chains in one handle (kabc_ais_create_batch)",
                "wall_ms": w50 * 1e3, "transitions_per_s": 50 * 1000 * NT_HEADLINE / w50,
                "vs_single_chain_throughput": (50 * 1000 * NT_HEADLINE / w50) / (1000 * NT_HEADLINE / w),
                "posterior_mean": rc50.reshape(-1, 2).mean(0).tolist()}
        except Exception as e:   # informational
            extra["readme_c1"]["chains_50"] = {"error": repr(e)}
        # README.md:80-84: `smc(prior, cost)` with its defaults on the same simulator
        for _ in range(2):
            k.smc(rm.prior, rm.cost, nparticles=100, seed=1, ctx=ctx, return_array=True)
        walls = []
        for _ in range(5):
            t0 = time.perf_counter()
            rs = k.smc(rm.prior, rm.cost, nparticles=100, seed=1, ctx=ctx, return_array=True)
            walls.append(time.perf_counter() - t0)
        w = sorted(walls)[len(walls) // 2]
        extra["readme_smc"] = {"workload": "README.md:80-84: smc(prior, cost), defaults (100 particles), "
                                           "1000 normals per cost evaluation",
                               "wall_ms": w * 1e3, "iterations": rs.info["iterations"], "eps": rs.eps,
                               "posterior_mean": rs.P.mean(0).tolist(),
                               "reference_documented": "2.0 +- 0.0062, 0.0401 +- 0.00081, eps 0.0111 (README.md:84)"}
        if cpu and isinstance(cpu.get("readme_smc"), dict) and "wall_s" in cpu["readme_smc"]:
            extra["readme_smc"]["cpu_baseline"] = cpu["readme_smc"]
            extra["readme_smc"]["vs_cpu_port_1core"] = cpu["readme_smc"]["wall_s"] / w
        # the "next" rows of SURVEY 8 (f3, f4): ABCDE and pfilter, wall clock against the oracle
        try:
            n2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
            gd = k.costs.GaussDist([1.0, -0.5])
            nr = {}
            for name, fn, kw_n in (
                    ("abcde", lambda **kw: k.ABCDE(n2, gd, 0.01, ctx=ctx, return_array=True, **kw),
                     dict(nparticles=2000, generations=50, seed=3)),
                    ("pfilter", lambda **kw: k.pfilter(n2, gd, 16384, ctx=ctx, return_array=True, **kw),
                     dict(q=0.7, eff_tol=0.1, epstol=0.02, seed=3))):
                fn(**kw_n)
                ws = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    rr = fn(**kw_n)
                    ws.append(time.perf_counter() - t0)
                w = sorted(ws)[2]
                nr[name] = {"wall_ms": w * 1e3,
                            "workload": ("ABCDE(prior, cost, 0.01; nparticles=2000, generations=50), Normal(0,5)^2 + gauss_dist "
                                         "(src/smc.jl:347-430)" if name == "abcde" else
                                         "pfilter(prior, cost, 16384; q=0.7, eff_tol=0.1, epstol=0.02), Normal(0,5)^2 + gauss_dist "
                                         "(src/smc.jl:275-340)"),
                            "iterations": int(rr.info["generations_run"] if name == "abcde" else rr.info["iterations"])}
                if cpu and isinstance(cpu.get(name), dict) and "wall_s" in cpu[name]:
                    nr[name]["cpu_baseline"] = cpu[name]
                    nr[name]["vs_cpu_port_1core"] = cpu[name]["wall_s"] / w
            extra["next_rows"] = nr
        except Exception as e:  # informational legs: never fail the bench on them
            extra["next_rows"] = {"error": repr(e)}
        # sample() calls shaped like the reference's own tests (REFERENCE_SHAPES): small ensembles, the
        # default ntransitions = 1, long burn-ins -- on the one-workgroup driver (every generation of an
        # advance call in ONE launch: csrc/ais_small_kernel.hpp; the default) and with a launch per
        # half-generation (KABC_AIS_SMALL=0), the oracle's one-core wall for the same step() calls beside
        try:
            m2 = c2_problem(k)
            shapes = {}
            cpu_rs = (cpu or {}).get("reference_shapes", {}).get("shapes", {}) if cpu else {}
            for w_, ns_, disc_, nt_ in REFERENCE_SHAPES:
                kw_s = dict(ntransitions=nt_, discard_initial=disc_, seed=1, ctx=ctx, return_array=True)
                gens = -(-disc_ // w_) + max(1, -(-ns_ // w_))
                ent = {"generations": gens}
                for drv in ("small", "halves"):
                    with _env("KABC_AIS_SMALL", "1" if drv == "small" else "0"):
                        k.sample(m2, k.AIS(w_), ns_, **kw_s)
                        ws = []
                        for _ in range(5):
                            t0 = time.perf_counter()
                            k.sample(m2, k.AIS(w_), ns_, **kw_s)
                            ws.append(time.perf_counter() - t0)
                    wm = sorted(ws)[2]
                    ent[drv] = {"wall_ms": wm * 1e3, "us_per_half_generation": wm * 1e6 / (2 * gens),
                                "transitions_per_s": gens * w_ * nt_ / wm}
                key = _shape_key(w_, ns_, disc_, nt_)
                if key in cpu_rs:
                    ent["cpu_port_1core_wall_ms"] = cpu_rs[key]["wall_ms"]
                    ent["vs_cpu_port_1core"] = cpu_rs[key]["wall_ms"] / ent["small"]["wall_ms"]
                if w_ == 12:   # 50 chains of AIS(12) in one handle (test/runtests.jl:88-104): chain = workgroup
                    k.sample(m2, k.AIS(w_), k.MCMCThreads(), ns_, 50, **kw_s)
                    ws = []
                    for _ in range(5):
                        t0 = time.perf_counter()
                        k.sample(m2, k.AIS(w_), k.MCMCThreads(), ns_, 50, **kw_s)
                        ws.append(time.perf_counter() - t0)
                    ent["small_50_chains_wall_ms"] = sorted(ws)[2] * 1e3
                shapes[key] = ent
            extra["reference_shapes"] = {
                "workload": "sample(model, AIS(N), Ns; discard_initial, ntransitions) on the C2 model (Normal(0,5)^2, "
                            "gauss_dist, scale 0.1), wall clock of the whole call incl. handle creation, init, the "
                            "trace copy and the Python wrapper; median of 5",
                "drivers": {"small": "one workgroup per chain, one launch per advance call (default, N <= 512)",
                            "halves": "one launch per half-generation (KABC_AIS_SMALL=0)"},
                "shapes": shapes}
        except Exception as e:   # informational
            extra["reference_shapes"] = {"error": repr(e)}
        # BASELINE.json configs[1]
        with _env("KABC_SPECIALIZE", "0"):
            c2_pre = kernel_leg(c2_problem(k), 4096, 2, nts=(NT_HEADLINE,))
        extra["c2"] = dict(kernel_leg(c2_problem(k), 4096, 2, nts=(NT_HEADLINE, 16)), prebuilt=c2_pre,
                           workload="C2: AIS 4096 walkers, D=2, Normal(0,5)^2, gauss_dist, scale 0.1 "
                                    "(BASELINE.json configs[1]); 32 workgroups: bound by the latency of "
                                    "one wavefront's chain of dependent transitions, not by throughput")
        # C2 is not a throughput workload: 32 workgroups, each ONE consumer wavefront working through
        # its walkers' dependent transitions.  Its floor is that chain issued back to back --
        # 0.36 us per sub-step (DESIGN.md 6 / profiles/r03_c2_ablation.txt: ~95 dependent issue
        # slots of 3.8 ns) -- plus the launch boundary; `latency_floor_frac` = floor / measured is
        # the number that can move (the contract roofline fraction cannot: 0.04 at best here).
        for nt_s, v in list(extra["c2"].items()):
            if isinstance(v, dict) and "kernel_avg_us" in v:
                floor_us = 0.36 * int(nt_s) + 3.6
                v["latency_floor_us"] = floor_us
                v["latency_floor_frac"] = floor_us / v["kernel_avg_us"]
        if cpu and isinstance(cpu.get("c2"), dict) and "value" in cpu["c2"]:
            extra["c2"]["cpu_baseline"] = cpu["c2"]
        # The prior classes the reference's own tests use beside boxes, on the DEFAULT path: nothing
        # but k.AisEnsemble(model, N) -- the library specialises the model on its own without ever
        # waiting for the compiler (include/kabc.h "THE DEFAULT"; here the unit comes from the
        # on-disk cache build() warmed: switched_after_launches = 0).  `prebuilt` = the same call
        # under KABC_SPECIALIZE=0; `cold` = the same call in a fresh process with an EMPTY cache:
        # first-call latency against the prebuilt-only call, seconds until the handle switched to
        # its own kernels, launches that ran before.  Same bits on every path.
        bpc = {}
        for name, m, Nm, Dm in prior_class_problems(k):
            with _env("KABC_SPECIALIZE", "0"):
                pre = kernel_leg(m, Nm, Dm)
            cur = kernel_leg(m, Nm, Dm)
            entry = dict(cur, N=Nm, D=Dm, prebuilt=pre,
                         kernel={"active": "the model's own (default path, k.AisEnsemble only)",
                                 "pending": "prebuilt while the worker compiles (cold cache)"}.get(
                                     cur["spec_state"], "prebuilt (" + cur["spec_state"] + ")"))
            if isinstance(cold_spec, dict) and name in cold_spec:
                entry["cold"] = cold_spec[name]
                entry["cold_compile_s"] = cold_spec[name].get("cold_compile_s")
            elif isinstance(cold_spec, dict) and "error" in cold_spec:
                entry["cold"] = cold_spec
            bpc[name] = entry
        extra["by_prior_class"] = bpc

    smc = None
    if world == 1 and not args.no_smc and isinstance(ctx, k.Context):
        prior, cost, kw = c4_problem(k)
        # warm-up: module load, allocations, and a one-off ~25 ms hiccup on the fourth call that
        # returns the 4 MiB particle array (host-side; steady state afterwards)
        for _ in range(6):
            k.smc(prior, cost, ctx=ctx, return_array=True, **kw)
        walls, r = [], None
        for _ in range(7):
            t0 = time.perf_counter()
            r = k.smc(prior, cost, ctx=ctx, return_array=True, **kw)
            walls.append(time.perf_counter() - t0)
        w = sorted(walls)[len(walls) // 2]
        Bs = 32 * SMC_D + 33
        ups = r.info["proposals"] / w
        smc = {"workload": "smc C4: 32768 particles x 16-param hierarchical Gaussian sim, "
                           "alpha=0.95 epstol=0.05 seed=1 (BASELINE.json configs[3])",
               "wall_ms": w * 1e3, "wall_ms_all": [x * 1e3 for x in walls],
               "iterations": r.info["iterations"], "eps": r.eps,
               "particle_updates": r.info["proposals"], "particle_updates_per_s": ups,
               "cost_evals": r.info["cost_evals"],
               "roofline": {"bound": "hbm", "bytes_per_update": Bs,
                            "achieved": ups * Bs / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": ups * Bs / 1e9 / HBM_PEAK_GBS,
                            "note": "end to end (select + propose/accept + control), wall clock"},
               "mcmc_kernel_avg_ms": r.info["kernel_ms_mcmc"]}
        # What an ε-iteration of the persistent loop kernel costs and what bounds it: two device-wide
        # barriers at the XCD-aware price measured on this chip (3.8 us each at 128 workgroups:
        # profiles/r03_xcd_barrier.json, rows written / barrier / another workgroup's rows read) and the
        # pass's own dependent chain (draws -> resample lookups -> three rows -> prior -> simulator ->
        # accept: 7.2 us by the kernel's phase stamps, profiles/r04_smc_c4.txt).  The contract roofline
        # fraction of this workload cannot move (a pass is 17.9 MB = 2.2 us at 8 TB/s);
        # latency_floor_frac = floor / measured is the number that can.
        it_us = w * 1e6 / max(1, r.info["iterations"])
        fi = floor_in or {"barrier_128_us": 3.745, "barrier_source": "profiles/r03_xcd_barrier.json",
                          "pass_chain_us": 7.2, "pass_chain_source": "profiles/r04_smc_c4.txt"}
        floor_us = 2 * fi["barrier_128_us"] + fi["pass_chain_us"]
        smc["iteration_us"] = it_us
        smc["latency_floor_us"] = floor_us
        smc["latency_floor_frac"] = floor_us / it_us
        smc["latency_floor_inputs"] = fi
        smc["latency_floor_formula"] = (f"2 x XCD-aware barrier at 128 workgroups ({fi['barrier_128_us']:.3f} us: "
                                        f"{fi['barrier_source']}) + the pass's dependent chain "
                                        f"({fi['pass_chain_us']:.2f} us: {fi['pass_chain_source']})")
        # HBM bytes of the loop kernel (one launch = the whole run) from the committed PMC passes
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_smc_loop.json"))):
            tj = json.load(open(tf))
            smc["roofline"]["traffic"] = tj.get("hbm_bytes_per_launch")
            smc["roofline"]["traffic_vs_algorithmic"] = (tj.get("hbm_bytes_per_launch", 0) /
                                                         max(1, r.info["proposals"] * Bs))
            smc["roofline"]["traffic_source"] = os.path.basename(tf)
        if cpu and isinstance(cpu.get("smc_c4"), dict) and "wall_s" in cpu["smc_c4"]:
            smc["cpu_baseline"] = cpu["smc_c4"]
            smc["vs_cpu_port_1core"] = cpu["smc_c4"]["wall_s"] / w

    # RCCL prints a version banner to the C stdout of every rank when its communicator
    # comes up; push it out now so that the JSON below is the LAST line of the job
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if comm:
        comm.barrier()
    if rank == 0:
        bytes_per_eval = 8 * (3 * D + 4)
        rows = WALKERS_PER_GPU // 2
        pmc_file, pmc = _pmc_table()

        def summarise(nt_r, reg):
            assert reg["proposals"] == n_total * nt_r * reg["steps"] * reg["blocks"], reg
            alg = rows * nt_r * bytes_per_eval
            kms = reg["kms"]
            ach = alg / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
            valu = None
            vi = (pmc.get(str(nt_r)) or {}).get("SQ_INSTS_VALU")
            if vi and kms > 0:
                # issue time per wave-instruction: the measured rates of this chip by class
                # (profiles/r01m_valu_rate_8waves.json: f64 add / mul / fma and v_mad_u64_u32 issue
                # in ~1.9 ns = 4.6 cycles per SIMD when 8 waves share it, other VALU in 4 cycles)
                # weighted with the kernel's own instruction mix (profiles/r04_valu_mix.json, from
                # the shipped code object: tools/kernel_disasm.py)
                valu = vi * VALU_CYCLES / (SIMDS * CLOCK_HZ * kms * 1e-3)
            return {"value": reg["proposals"] / reg["el"], "unit": "evals/s",
                    "steps": reg["steps"], "blocks": reg["blocks"], "timed_s": reg["el"],
                    "ms_per_step": reg["el"] / (reg["steps"] * reg["blocks"]) * 1e3,
                    "kernel_avg_us": kms * 1e3, "kernel_launches_timed": reg["nl"],
                    "algorithmic_bytes_per_launch": alg, "roofline_achieved_GBps": ach,
                    "roofline_frac": ach / HBM_PEAK_GBS, "valu_frac": valu,
                    "valu_wave_insts_per_launch": vi,
                    "cost_evals_per_s": reg["cost_evals"] / reg["el"],
                    "accept_rate": reg["accepted"] / max(1, reg["proposals"]),
                    **({"exchange": reg["exchange"]} if reg.get("exchange") else {})}

        by_nt = {str(nt_r): summarise(nt_r, reg) for nt_r, reg in regions.items()}
        h = by_nt[str(nt_head)]
        # What a smaller ntransitions could reach at best with this kernel: every half-generation
        # must see the rows the previous one wrote, and the cheapest way to pay for that on this
        # chip is the kernel boundary itself (profiles/r03_halfgen_floor.json: 3.6 us per half at
        # 512 workgroups, against 18-31 us for a device-wide barrier inside one launch).
        if str(100) in by_nt and by_nt["100"]["kernel_avg_us"] > 0:
            try:
                fl = json.load(open(os.path.join(ROOT, "profiles", "r03_halfgen_floor.json")))
                launch_us = float(fl.get("launch_512", 3.6))
            except Exception:
                launch_us = 3.6
            # steady-state time of one sub-step: the slope between the 16 and the 100 setting
            if "16" in by_nt:
                sub_us = (by_nt["100"]["kernel_avg_us"] - by_nt["16"]["kernel_avg_us"]) / 84.0
            else:
                sub_us = by_nt["100"]["kernel_avg_us"] / 100.0
            for nt_s, v in by_nt.items():
                nt_i = int(nt_s)
                floor_us = nt_i * sub_us + launch_us
                v["ceiling"] = {"substep_us": sub_us, "launch_boundary_us": launch_us,
                                "overhead_us": v["kernel_avg_us"] - nt_i * sub_us,
                                "kernel_floor_us": floor_us,
                                "roofline_frac_ceiling": rows * nt_i * bytes_per_eval / (floor_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                "formula": "ntransitions x steady-state sub-step + the launch boundary (measured "
                                           "without arithmetic, profiles/r03_halfgen_floor.json); overhead_us = "
                                           "what a launch adds to its sub-steps today (boundary + 3.3 us pipeline "
                                           "fill + table staging / drain)"}
        # HBM bytes per launch from PMC counters are collected in separate rocprofv3
        # passes (FETCH_SIZE / WRITE_SIZE cannot share a pass); the committed summary
        # of that run is reported here when it was taken on this very workload.
        traffic, traffic_file = None, None
        for tf in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_traffic_nt{nt_head}.json"))):
            traffic = json.load(open(tf)).get("hbm_bytes_per_launch")
            traffic_file = os.path.basename(tf)
        out = {
            "metric": "walker proposal+cost evals/sec at N=65536 walkers, D=8",
            "value": h["value"], "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": h["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "AIS C3: 65536 walkers/GPU x 8-param Rosenbrock-like cost, "
                                   "Uniform(-5,5)^8 prior, kernelized scale 1.0",
                       "walkers_per_gpu": WALKERS_PER_GPU, "walkers_total": n_total, "D": D,
                       "ntransitions": nt_head,
                       "ntransitions_note": "headline = the reference's own setting (README.md:57, "
                                            "BASELINE.json configs[0]); SURVEY 8d's {1, 16} are "
                                            "under by_ntransitions",
                       "evals_per_step": n_total * nt_head, "seed": SEED,
                       "timed_blocks": h["blocks"], "timed_seconds": h["timed_s"],
                       "parallelism": (f"REHEARSAL: {emulate} emulated ranks on ONE GPU, P2P exchange "
                                       f"(KABC_BENCH_EMULATE_RANKS); not a scaling figure" if emulate else
                                       f"walker-sharded x{world}, 1 RCCL all-gather per "
                                       f"half-generation issued by libkabc_hip (no torch.distributed)"
                                       if world > 1 else "single GPU"),
                       **({"emulated_ranks": emulate} if emulate else {})},
            # SURVEY 8d quotes C3 at ntransitions in {1, 16}: the same run's figures at 16, on the first screen
            **({"value_survey_8d": {"ntransitions": 16, "value": by_nt["16"]["value"], "unit": "evals/s",
                                    "roofline_frac": by_nt["16"]["roofline_frac"],
                                    "kernel_avg_us": by_nt["16"]["kernel_avg_us"],
                                    "ntransitions_1": {"value": by_nt["1"]["value"],
                                                       "roofline_frac": by_nt["1"]["roofline_frac"],
                                                       "kernel_avg_us": by_nt["1"]["kernel_avg_us"]}
                                    if "1" in by_nt else None}} if "16" in by_nt else {}),
            "kabc_specialize_env": os.environ.get("KABC_SPECIALIZE"),
            "cost_evals_per_s": h["cost_evals_per_s"], "accept_rate": h["accept_rate"],
            "roofline": {"bound": "hbm", "achieved": h["roofline_achieved_GBps"],
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": h["roofline_frac"],
                         "traffic": traffic,
                         "replayed_from": {"traffic": traffic_file, "valu.wave_insts_per_launch": pmc_file,
                                           "valu.issue_cycles_per_inst": VALU_MIX_FILE,
                                           "note": "PMC counters cannot be read inside this run (separate "
                                                   "rocprofv3 --pmc passes): these fields replay the committed "
                                                   "summaries of the builder's passes over this same command; "
                                                   "achieved / frac / kernel_avg_ms are measured live"},
                         "kernel": "ais_half_kernel<8, rosenbrock, BOX, kernelized>",
                         "kernel_avg_ms": h["kernel_avg_us"] / 1e3,
                         "kernel_launches_timed": h["kernel_launches_timed"],
                         "algorithmic_bytes_per_launch": h["algorithmic_bytes_per_launch"],
                         "valu": {"frac": h["valu_frac"],
                                  "wave_insts_per_launch": h["valu_wave_insts_per_launch"],
                                  "issue_cycles_per_inst": VALU_CYCLES,
                                  "formula": "SQ_INSTS_VALU x the mix-weighted measured issue cycles per "
                                             "instruction / (1024 SIMDs x 2.4 GHz x kernel time): the "
                                             "binding resource (state is register-resident, HBM traffic "
                                             "is ~3 % of the algorithmic bytes)",
                                  "source": pmc_file, "mix_source": VALU_MIX_FILE}},
            "by_ntransitions": by_nt,
        }
        out.update(extra)
        if smc is not None:
            out["smc_c4"] = smc
            if sharded1 is not None:
                out["smc_sharded_world1"] = sharded1
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    ens.close()
    if comm:
        comm.close()


if __name__ == "__main__":
    main()
