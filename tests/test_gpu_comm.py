"""The library-owned exchange of the walker-sharded path (include/kabc.h "multi-GPU"):
kabc_comm_* + kabc_ais_create_dist + the *_multi drivers.

A 1-GPU box cannot host two RCCL ranks, so the multi-rank logic (row ownership with
uneven / empty shards, global walker ids, partner draws over the GLOBAL complementary
half, the in-place all-gather layout, event ordering) runs on the P2P backend with
every rank on device 0 -- each rank owns its own copy of the global half buffers, so a
missing or misplaced exchange shows up as a mismatch against the single-process oracle.
The RCCL backend itself is exercised at world size 1 (communicator set-up from a
unique id, ncclAllGather in place on the context stream, the host-value reductions)
in a torch-free child process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(k, D=8):
    return k.ApproxKernelizedPosterior(k.Factored(*[k.Uniform(-5, 5)] * D), k.costs.Rosenbrock(), 1.0)


@pytest.mark.parametrize("world,N,D,nt,gens", [(2, 2048, 8, 5, 3), (3, 1001, 4, 3, 4),
                                                (8, 523, 8, 2, 3), (5, 13, 8, 4, 2)])
def test_p2p_emulated_ranks_match_oracle(k, orc, gpu_ctx, world, N, D, nt, gens):
    model = _model(k, D)
    grp = k.EnsembleGroup(model, N, seed=17, devices=[0] * world, backend="p2p").init()
    o = orc.OracleAIS(model, N, seed=17).init()
    assert np.array_equal(grp.ensemble(0), o.state()[0])           # init + first gather
    grp.advance(gens, nt)
    o.generations_sync(gens, nt, collect=False)
    xo, lpo, llo, _ = o.state()
    for r in range(world):                                         # every rank holds everything
        assert np.array_equal(grp.ensemble(r), xo), f"rank {r}"
    x, lp, ll = grp.state()
    assert np.array_equal(x, xo) and np.array_equal(lp, lpo) and np.array_equal(ll, llo)
    assert grp.stats() == o.stats()
    assert grp.last_stats["proposals"] == N * nt * gens
    grp.close()


@pytest.mark.parametrize("world,N,D,nt,gens,K", [(2, 2048, 8, 5, 3, 4), (3, 1001, 4, 3, 4, 3),
                                                  (8, 523, 8, 2, 3, 2), (5, 13, 8, 4, 2, 4),
                                                  (4, 4100, 2, 1, 6, 16)])
def test_pipelined_exchange_chunks_match_oracle(k, orc, gpu_ctx, monkeypatch, world, N, D, nt, gens, K):
    """KABC_EXCHANGE_CHUNKS = K: block-cyclic ownership, chunk k gathered on the exchange stream
    while the kernels of chunk k + 1 run.  Same trajectory as the oracle (and as K = 1): the
    draws are keyed by walker id, not by owner.  Includes shards with empty segments."""
    monkeypatch.setenv("KABC_EXCHANGE_CHUNKS", str(K))
    model = _model(k, D)
    grp = k.EnsembleGroup(model, N, seed=17, devices=[0] * world, backend="p2p").init()
    segs = [grp.shards[r].segments(0) for r in range(world)]
    assert all(len(sg) == K for sg in segs)
    covered = sorted((f, c) for sg in segs for f, c in sg if c > 0)
    assert covered[0][0] == 0 and sum(c for _, c in covered) == (N + 1) // 2
    assert all(a[0] + a[1] == b[0] for a, b in zip(covered, covered[1:]))   # a partition of the half
    o = orc.OracleAIS(model, N, seed=17).init()
    assert np.array_equal(grp.ensemble(world - 1), o.state()[0])
    for _ in range(2):                                             # two calls: the fences between them
        grp.advance(gens, nt)
        o.generations_sync(gens, nt, collect=False)
        xo, lpo, llo, _ = o.state()
        for r in range(world):
            assert np.array_equal(grp.ensemble(r), xo), f"rank {r}"
    x, lp, ll = grp.state()
    assert np.array_equal(x, xo) and np.array_equal(lp, lpo) and np.array_equal(ll, llo)
    assert grp.stats() == o.stats()
    grp.close()


def test_default_chunking_follows_the_residency_wave(k, orc, gpu_ctx):
    """No knob: one exchange chunk per 512 workgroups (one residency wave of the half-generation
    kernel) of a rank's share of a half -- 2^19 walkers on two ranks = 2048 workgroups per rank
    and half = 4 chunks; 65 536 walkers per rank (C5) stay at one."""
    model = _model(k, 8)
    N, nt = 1 << 19, 2
    grp = k.EnsembleGroup(model, N, seed=5, devices=[0, 0], backend="p2p").init()
    assert [len(s.segments(0)) for s in grp.shards] == [4, 4]
    assert grp.shards[1].segments(1)[0] == (32768, 32768)
    grp.advance(1, nt)
    o = orc.OracleAIS(model, N, seed=5).init()
    o.generations_sync(1, nt, collect=False)
    assert np.array_equal(grp.ensemble(0), o.state()[0]) and np.array_equal(grp.ensemble(1), o.state()[0])
    assert grp.stats() == o.stats()
    grp.close()


_C5 = {}


@pytest.mark.parametrize("K", [1, 4])
def test_c5_eight_emulated_ranks_full_size(k, orc, gpu_ctx, monkeypatch, K):
    """BASELINE.json configs[4] (C5): 524 288 walkers, D = 8, sharded 8 ways, at its full
    size -- 8 ranks on one GPU, exchange by the pull kernel, ntransitions = 16, two
    generations, bit-exact against the oracle; with the default single exchange chunk and
    with four pipelined ones."""
    model = _model(k, 8)
    N, nt, gens = 524288, 16, 2
    if K > 1:
        monkeypatch.setenv("KABC_EXCHANGE_CHUNKS", str(K))
    grp = k.EnsembleGroup(model, N, seed=1, devices=[0] * 8, backend="p2p").init()
    grp.advance(gens, nt)
    if not _C5:
        o = orc.OracleAIS(model, N, seed=1).init()
        o.generations_sync(gens, nt, collect=False)
        _C5["state"], _C5["stats"] = o.state(), o.stats()
    xo, lpo, llo, _ = _C5["state"]
    assert np.array_equal(grp.ensemble(7), xo) and np.array_equal(grp.ensemble(0), xo)
    x, lp, ll = grp.state()
    assert np.array_equal(lp, lpo) and np.array_equal(ll, llo)
    assert grp.stats() == _C5["stats"]
    assert [s.owned for s in grp.shards] == [(32768, 32768)] * 8
    assert [len(s.segments(0)) for s in grp.shards] == [K] * 8
    grp.close()


def test_group_argument_checks(k, gpu_ctx):
    model = _model(k, 4)
    grp = k.EnsembleGroup(model, 64, seed=1, devices=[0, 0], backend="p2p")
    with pytest.raises(k.KabcError, match="kabc_ais_init_multi"):
        grp.shards[0].init()
    grp.init()
    with pytest.raises(k.KabcError, match="kabc_ais_advance_multi"):
        grp.shards[1].advance(1, 1)
    grp.close()
    with pytest.raises(k.KabcError):
        k.comm.init_all([0] * 17, "p2p")


CHILD = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, {root!r})
import kissabc_jl_amd as k
assert "torch" not in sys.modules
model = k.ApproxKernelizedPosterior(k.Factored(*[k.Uniform(-5, 5)] * 4), k.costs.Rosenbrock(), 1.0)
mode = {mode!r}
if mode == "rank":          # one process per GPU: unique id -> ncclCommInitRank
    comm = k.Comm.from_env()
    ens = k.AisEnsemble(model, 1024, seed=11, comm=comm).init()
    ens.set_timing(32, stride=1)
    ens.advance(5, 7)
    xch = ens.exchange_us()      # kabc_ais_exchange_us: hipEvents around kernels and all-gathers
    ens.set_timing(0)
    x = ens.ensemble()
    st = ens.stats()
    red = comm.allreduce_sum([st["proposals"], 3])
    mx = comm.allreduce_max([1.5, -2.0])
    comm.barrier()
    ens.close(); comm.close()
    extra = {{"red": red, "mx": mx, "xch": xch}}
elif mode == "all":         # one process: ncclCommInitAll + grouped all-gather
    grp = k.EnsembleGroup(model, 1024, seed=11, devices=[0], backend="rccl").init()
    grp.advance(5, 7)
    x = grp.ensemble(0)
    st = grp.stats()
    grp.close()
    extra = {{}}
else:                       # no communicator at all
    ens = k.AisEnsemble(model, 1024, seed=11).init()
    ens.advance(5, 7)
    x = ens.ensemble()
    st = ens.stats()
    extra = {{}}
assert "torch" not in sys.modules
np.save({out!r}, x)
import ctypes
ctypes.CDLL(None).fflush(None)   # RCCL's start-up banner sits in the C stdout buffer
print(json.dumps(dict(st, **extra)), flush=True)
"""


def _run_child(tmp_path, mode, **extra_env):
    out = str(tmp_path / f"x_{mode}.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", KABC_NO_TORCH_PRELOAD="1", **extra_env)
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, out=out, mode=mode)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out), json.loads(r.stdout.strip().splitlines()[-1])


def test_rccl_world1_through_the_c_abi(tmp_path):
    x0, s0 = _run_child(tmp_path, "plain")
    x1, s1 = _run_child(tmp_path, "rank")
    assert np.array_equal(x1, x0)
    assert s1["proposals"] == s0["proposals"] == 1024 * 5 * 7 and s1["accepted"] == s0["accepted"]
    assert s1["red"] == [1024 * 5 * 7, 3] and s1["mx"] == [1.5, -2.0]
    # the exchange diagnostics of the first multi-GPU run exist at world 1 already
    xc = s1["xch"]
    assert xc["chunks"] == 1 and xc["compute_us_per_half"] > 1.0 and xc["exchange_us_per_half"] > 0.0
    assert 0.0 <= xc["exposed_us_per_half"] < 1e4
    x2, s2 = _run_child(tmp_path, "all")
    assert np.array_equal(x2, x0) and s2["accepted"] == s0["accepted"]


def test_rccl_world1_pipelined_chunks(tmp_path):
    """The pipelined path on the RCCL backend: ncclAllGather per exchange chunk on the exchange
    stream behind one event per chunk (one-process-per-GPU and grouped single-process forms)."""
    x0, s0 = _run_child(tmp_path, "plain")
    x1, s1 = _run_child(tmp_path, "rank", KABC_EXCHANGE_CHUNKS="3")
    assert np.array_equal(x1, x0) and s1["accepted"] == s0["accepted"]
    assert s1["red"] == [1024 * 5 * 7, 3]
    assert s1["xch"]["chunks"] == 3 and s1["xch"]["exchange_us_per_half"] > 0.0
    x2, s2 = _run_child(tmp_path, "all", KABC_EXCHANGE_CHUNKS="4")
    assert np.array_equal(x2, x0) and s2["accepted"] == s0["accepted"]


SMC_CHILD = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, {root!r})
import kissabc_jl_amd as k
prior = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
cost = k.costs.NoisyBanana(0.5)
kw = dict(nparticles=9000, alpha=0.9, epstol=0.05, seed=4, return_array=True)
plain = k.smc(prior, cost, **kw)
comm = k.Comm.from_env()
res = {{}}
for shard in ("cost_loop", "particles"):
    got = k.smc(prior, cost, comm=comm, shard=shard, **kw)
    res[shard] = bool(got.eps == plain.eps and np.array_equal(got.info["theta_all"], plain.info["theta_all"])
                      and np.array_equal(got.C, plain.C) and np.array_equal(got.info["alive"], plain.info["alive"])
                      and got.info["log"] == plain.info["log"])
comm.close()
assert "torch" not in sys.modules
import ctypes
ctypes.CDLL(None).fflush(None)
print(json.dumps(dict(res, iterations=plain.info["iterations"])), flush=True)
"""


def test_smc_dist_modes_on_an_rccl_communicator_world1():
    """both modes of kabc_smc_run_dist_mode on a one-process-per-GPU RCCL communicator (world 1: every
    all-gather of the sharded selection is a real ncclAllGather on the context's stream)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", KABC_NO_TORCH_PRELOAD="1")
    r = subprocess.run([sys.executable, "-c", SMC_CHILD.format(root=ROOT)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["cost_loop"] is True and d["particles"] is True and d["iterations"] > 20


def _bench(extra_env, launcher):
    env = dict(os.environ, KABC_FORCE_COLLECTIVE="1", **extra_env)
    cmd = launcher + [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "1",
                      "--no-cpu-baseline", "--min-seconds", "0.05", "--headline-seconds", "0.05"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0], r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_with_communicator_env_launch():
    d = _bench(dict(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", LOCAL_RANK="0",
                    WORLD_SIZE="1"), [sys.executable])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["roofline"]["frac"] > 0
    assert set(d["by_ntransitions"]) == {"1", "16", "100"} and d["config"]["ntransitions"] == 100
    assert d["smc_c4"]["iterations"] > 100 and d["smc_c4"]["wall_ms"] > 0
    # the fields the first multi-GPU run will be read by: exchange / exposed / compute per half, chunks
    for nt in ("16", "100"):
        x = d["by_ntransitions"][nt]["exchange"]
        assert x["chunks"] >= 1 and x["compute_us_per_half"] > 0 and x["exchange_us_per_half"] > 0
        assert x["exposed_us_per_half"] >= 0


def test_bench_under_the_drivers_launcher():
    """Launched exactly as the driver launches N > 1 (python -m torch.distributed.run ...),
    here with one rank: the ranks' rendezvous variables come from the launcher."""
    d = _bench({}, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                    "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29571"])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["value"] > 0


@pytest.mark.parametrize("world,N,name", [(2, 3000, "gauss"), (4, 1000, "mixture_retrys"), (3, 5001, "hier"),
                                          (3, 700, "readme_sim")])
def test_smc_sharded_cost_loop_matches_single_gpu_and_oracle(k, orc, gpu_ctx, world, N, name):
    """kabc_smc_run_dist: the reference's parallel leg (src/smc.jl:120-123,168) across ranks --
    every rank holds the ensemble, evaluates prior-MH + cost for its blocks of 64 particles,
    one grouped all-gather per pass.  Ranks are host threads on the P2P backend (one GPU);
    every rank returns the single-GPU result, which equals the oracle's, bit for bit (uneven and
    empty shards, retry passes, stochastic costs)."""
    import threading
    rng = np.random.default_rng(3)
    if name == "gauss":
        prior, cost = k.Factored(k.Normal(0, 2), k.Normal(0, 2)), k.costs.GaussDist([0.5, -0.25])
        kw = dict(nparticles=N, epstol=0.05, seed=8)
    elif name == "mixture_retrys":
        prior, cost = k.Uniform(-10, 10), k.costs.Mixture(0.0)
        kw = dict(nparticles=N, mcmc_retrys=3, epstol=0.2, seed=5)
    elif name == "readme_sim":   # the expensive simulator sharding is for; its pre-pass per shard
        prior = k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100))
        cost = k.costs.NormalMeanStdSim(400, 2.0012, 0.0401)
        kw = dict(nparticles=N, epstol=0.03, seed=3)
    else:
        prior = k.Factored(k.Normal(0, 5), k.Uniform(0, 5), *[k.Normal(0, 1)] * 6)
        cost = k.costs.HierGaussSim(rng.normal(size=6))
        kw = dict(nparticles=N, epstol=0.4, seed=2)
    ref = orc.smc(prior, cost, **kw)
    single = k.smc(prior, cost, return_array=True, **kw)
    assert single.eps == ref["eps"] and np.array_equal(single.info["theta_all"], ref["theta_all"])
    comms = k.comm.init_all([0] * world, "p2p")
    out, err = [None] * world, []

    def run(r):
        try:
            out[r] = k.smc(prior, cost, return_array=True, comm=comms[r], **kw)
        except Exception as e:          # a failing rank must not leave the others in the rendezvous
            err.append(repr(e))

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not err and all(o is not None for o in out), err
    for r in range(world):
        assert out[r].eps == ref["eps"], r
        assert np.array_equal(out[r].info["theta_all"], ref["theta_all"]), r
        assert np.array_equal(out[r].C, single.C) and np.array_equal(out[r].info["alive"], single.info["alive"])
        assert out[r].info["iterations"] == ref["iterations"]
        assert out[r].info["cost_evals"] == single.info["cost_evals"]
        assert out[r].info["proposals"] == single.info["proposals"]
    for c in comms:
        c.close()


def _run_ranks(k, comms, fn):
    """one host thread per rank of a P2P communicator group; returns the per-rank results"""
    import threading
    world = len(comms)
    out, err = [None] * world, []

    def run(r):
        try:
            out[r] = fn(comms[r])
        except Exception as e:          # a failing rank must not leave the others in the rendezvous
            err.append(repr(e))

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=900)
    assert not err and all(o is not None for o in out), err
    return out


@pytest.mark.parametrize("world,name", [(w, n) for n in ["banana_inf", "C4_hier16_small", "gauss_d2_minress",
                                                         "mixture_retrys", "dirac", "ties"] for w in (2, 4, 8)]
                         + [(3, "hier16_40k"), (3, "readme_sim_retrys"), (2, "readme_defaults")])
def test_smc_sharded_particles_equals_oracle(k, orc, gpu_ctx, world, name):
    """kabc_smc_run_dist_mode(KABC_SMC_DIST_PARTICLES) -- SURVEY §8e "SMC": the ranks OWN their particles;
    ε from all-gathered histograms of the order-preserving keys + a candidate gather, ESS from gathered
    counts, the resample index from the gathered compacted segments (src/smc.jl:131-153), partners from
    the gathered ensemble.  Ranks are host threads on the P2P backend (one GPU; uneven and empty shards:
    `dirac` has 100 particles = 2 blocks for up to 8 ranks).  Every rank returns the oracle's result bit
    for bit: positions, costs, alive mask, ε and the per-iteration log."""
    from test_gpu_smc_parity import _cases
    if name == "ties":   # heavy ties: whole key ranges collapse to one value (state 2 of the narrowing)
        prior, cost = k.Factored(k.Normal(1, 0.5), k.DiscreteUniform(1, 10)), k.costs.NoisyQuadDU(5.5)
        kw = dict(nparticles=6000, epstol=0.05)
    elif name == "hier16_40k":   # several workgroups per pass and rank, histogram rounds, uneven shards
        prior, cost, _ = _cases(k)["C4_hier16_small"]
        kw = dict(nparticles=40000, alpha=0.95, epstol=0.3)
    else:
        prior, cost, kw = _cases(k)[name]
    ref = orc.smc(prior, cost, seed=5, **kw)
    comms = k.comm.init_all([0] * world, "p2p")
    out = _run_ranks(k, comms, lambda c: k.smc(prior, cost, seed=5, return_array=True, comm=c,
                                               shard="particles", **kw))
    for r in range(world):
        got = out[r]
        assert got.info["iterations"] == ref["iterations"], r
        assert got.info["log"] == ref["log"], r
        assert got.eps == ref["eps"], r
        assert np.array_equal(got.info["alive"], ref["alive"]), r
        assert np.array_equal(got.info["theta_all"], ref["theta_all"]), r
        assert np.array_equal(got.C, ref["C"]), r
        assert got.info["cost_evals"] == ref["cost_evals"] and got.info["proposals"] == ref["proposals"]
    for c in comms:
        c.close()


@pytest.mark.parametrize("world", [1, 3])
def test_smc_sharded_particles_one_exchange_course(k, orc, gpu_ctx, world, monkeypatch):
    """the ONE-exchange selection (csrc/smc_dsel_kernels.hpp dsel2_*): batches of iterations enqueued between
    two looks, two collectives per iteration in the usual course (the selection's payload + the pass's grouped
    all-gather), the first two selections and every stalled one phase by phase; KABC_SMC_DIST_LOOKS=1 is the
    phase-by-phase course throughout.  Same bits either way (and the oracle's)."""
    from test_gpu_smc_parity import _cases
    prior, cost, _ = _cases(k)["C4_hier16_small"]
    kw = dict(nparticles=40000, alpha=0.95, epstol=0.3)
    ref = orc.smc(prior, cost, seed=5, **kw)
    res = {}
    for looks in ("0", "1"):
        monkeypatch.setenv("KABC_SMC_DIST_LOOKS", looks)
        comms = k.comm.init_all([0] * world, "p2p")
        res[looks] = _run_ranks(k, comms, lambda c: k.smc(prior, cost, seed=5, return_array=True, comm=c,
                                                          shard="particles", **kw))
        for c in comms:
            c.close()
    for looks, out in res.items():
        for got in out:
            assert got.info["log"] == ref["log"] and got.eps == ref["eps"], looks
            assert np.array_equal(got.info["theta_all"], ref["theta_all"]) and np.array_equal(got.C, ref["C"]), looks
            assert np.array_equal(got.info["alive"], ref["alive"]), looks
    it = ref["iterations"]
    d0, d1 = res["0"][0].info["dist"], res["1"][0].info["dist"]
    assert d0["batched"] and not d1["batched"] and d0["iterations"] == d1["iterations"] == it
    assert d0["collectives_per_usual_iteration"] == 2 and d1["collectives_per_usual_iteration"] == -1
    assert d1["one_exchange_selections"] == 0 and d1["host_looks"] > 2 * it and d1["collectives"] >= 4 * it
    # usual course: most selections decided by the one exchange; what the batches enqueue past the end of
    # the loop and behind a stalled selection is counted too
    assert d0["one_exchange_selections"] + d0["phase_by_phase_selections"] == it
    assert d0["one_exchange_selections"] >= 0.75 * it, d0
    assert d0["collectives"] <= 2 * it + 6 * d0["phase_by_phase_selections"] + 2 * 8 * (d0["phase_by_phase_selections"] + 1), d0
    assert d0["host_looks"] <= it // 2 + 8 * d0["phase_by_phase_selections"], d0


@pytest.mark.parametrize("shard,world", [("cost_loop", 3), ("particles", 3), ("particles", 2)])
def test_smc_sharded_beyond_16_parameters(k, orc, gpu_ctx, shard, world):
    """sharded smc on the run-time-dimension kernels (csrc/smc_dyn_kernels.hpp): every rank draws and costs the
    initial ensemble itself (counter-based draws), the team pass runs over the rank's particle range, the
    selection is the D-independent one -- the oracle's result on every rank, both modes, uneven shards"""
    rng = np.random.default_rng(8)
    D = 20
    pri = k.Factored(*[k.Normal(0, 2)] * (D - 1), k.DiscreteUniform(-3, 3))
    cost = k.costs.GaussDist(rng.normal(size=D))
    kw = dict(nparticles=2900, alpha=0.9, epstol=3.2, seed=4)
    ref = orc.smc(pri, cost, **kw)
    comms = k.comm.init_all([0] * world, "p2p")
    out = _run_ranks(k, comms, lambda c: k.smc(pri, cost, return_array=True, comm=c, shard=shard, **kw))
    for c in comms:
        c.close()
    assert ref["iterations"] > 5
    for got in out:
        assert got.info["log"] == ref["log"] and got.eps == ref["eps"]
        assert np.array_equal(got.info["theta_all"], ref["theta_all"]) and np.array_equal(got.C, ref["C"])
        assert np.array_equal(got.info["alive"], ref["alive"])
        assert got.info["cost_evals"] == ref["cost_evals"] and got.info["proposals"] == ref["proposals"]
        assert got.info["dist"]["batched"]


def test_smc_sharded_cost_loop_batches(k, orc, gpu_ctx, monkeypatch):
    """the cost-loop mode with the reference's default mcmc_retrys = 0: select, pass, all-gather, pass end
    enqueued eight iterations at a time (one collective per iteration)"""
    prior, cost = k.Factored(k.Normal(0, 2), k.Normal(0, 2)), k.costs.GaussDist([0.5, -0.25])
    kw = dict(nparticles=3000, epstol=0.05, seed=8)
    ref = orc.smc(prior, cost, **kw)
    comms = k.comm.init_all([0] * 2, "p2p")
    out = _run_ranks(k, comms, lambda c: k.smc(prior, cost, return_array=True, comm=c, shard="cost_loop", **kw))
    for c in comms:
        c.close()
    for got in out:
        assert got.info["log"] == ref["log"] and np.array_equal(got.info["theta_all"], ref["theta_all"])
        d = got.info["dist"]
        assert d["batched"] and d["iterations"] == ref["iterations"]
        assert d["collectives"] <= ref["iterations"] + 1 + 8 and d["host_looks"] <= ref["iterations"] // 8 + 2


def test_smc_dist_mode_argument_errors(k, gpu_ctx):
    import ctypes as C
    from kissabc_jl_amd import _cdefs as cd
    lib = k._lib.load()
    comms = k.comm.init_all([0], "p2p")
    prior = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    cost = k.costs.GaussDist([1.0, -0.5])
    o, r, cc = cd.SmcOpts(), cd.SmcResult(), cost.to_c()
    lib.kabc_smc_default_opts(C.byref(o))
    assert lib.kabc_smc_run_dist_mode(comms[0].handle, prior.to_c(), 2, C.byref(cc), C.byref(o), 7, C.byref(r)) != 0
    assert b"KABC_SMC_DIST_COST_LOOP or KABC_SMC_DIST_PARTICLES" in lib.kabc_last_error()
    assert lib.kabc_smc_run_dist_mode(None, prior.to_c(), 2, C.byref(cc), C.byref(o), 1, C.byref(r)) != 0
    assert b"communicator is NULL" in lib.kabc_last_error()
    with pytest.raises(ValueError):
        k.smc(prior, cost, comm=comms[0], shard="rows")
    # a world of one rank: the sharded selection without a peer equals kabc_smc_run
    a = k.smc(prior, cost, nparticles=5000, epstol=0.05, seed=2, return_array=True)
    b = k.smc(prior, cost, nparticles=5000, epstol=0.05, seed=2, return_array=True, comm=comms[0], shard="particles")
    assert a.eps == b.eps and np.array_equal(a.info["theta_all"], b.info["theta_all"]) and a.info["log"] == b.info["log"]
    comms[0].close()


def test_smc_sharded_particles_two_million(k, gpu_ctx):
    """the same at 2 097 152 particles x 16 parameters (C4's model) on 4 ranks, against kabc_smc_run on
    one GPU (which the parity suite pins to the oracle at the sizes the oracle finishes): several
    narrowing candidates per rank, 8192 blocks per rank, resamples of half a million rows"""
    from test_gpu_smc_parity import _cases
    prior, cost, _ = _cases(k)["C4_hier16_small"]
    kw = dict(nparticles=2097152, alpha=0.5, epstol=1.0, seed=11)
    single = k.smc(prior, cost, return_array=True, **kw)
    assert single.info["iterations"] >= 3 and any(l["resampled"] for l in single.info["log"])
    comms = k.comm.init_all([0] * 4, "p2p")
    out = _run_ranks(k, comms, lambda c: k.smc(prior, cost, return_array=True, comm=c, shard="particles", **kw))
    for got in out:
        assert got.eps == single.eps and got.info["log"] == single.info["log"]
        assert np.array_equal(got.info["alive"], single.info["alive"])
        assert np.array_equal(got.info["theta_all"], single.info["theta_all"])
        assert np.array_equal(got.C, single.C)
    for c in comms:
        c.close()


def test_bench_world_n_line_rehearsed_on_one_gpu(gpu_ctx):
    """bench.py's world-N line (the one `python bench.py --gpus N` relays from rank 0 on a multi-GPU
    node) rehearsed with N emulated ranks on ONE device through the P2P communicator
    (KABC_BENCH_EMULATE_RANKS): it carries n_gpus, the exchange diagnostics, and a value whose
    proposals equal walkers x sub-steps."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["KABC_BENCH_EMULATE_RANKS"] = "4"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--no-alt", "--no-smc", "--min-seconds", "0.05",
                        "--headline-seconds", "0.2"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 4 and d["config"]["emulated_ranks"] == 4 and d["scaling"] == "weak"
    assert d["config"]["walkers_total"] == 4 * 65536
    h = d["by_ntransitions"][str(d["config"]["ntransitions"])]
    x = h["exchange"]
    assert x["chunks"] >= 1 and x["compute_us_per_half"] > 0 and x["exchange_us_per_half"] > 0
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
