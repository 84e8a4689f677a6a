"""Wall time of sample() calls shaped like the reference's own tests (test/runtests.jl:82-290: small
ensembles, the default ntransitions = 1, long burn-ins), per call and per half-generation, on both
AIS drivers (KABC_AIS_SMALL: the one-workgroup kernel of csrc/ais_small_kernel.hpp / one launch per
half-generation), with the phases of a call and -- `--oracle` -- the CPU oracle's one-core wall for
the same number of transitions on the reference's serial schedule (src/KissABC.jl:66-80)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402

SHAPES = [(12, 500, 1000, 1), (100, 1000, 10000, 1), (50, 100, 50000, 1), (20, 100, 2000, 40), (10, 10000, 0, 50)]


def model():
    N2 = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    return k.ApproxKernelizedPosterior(N2, k.costs.GaussDist([1.0, -0.5]), 0.1)


def med(f, n=5):
    ws = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ws.append(time.perf_counter() - t0)
    return sorted(ws)[n // 2]


def phases(m, walkers, nsamples, discard, nt):
    """the calls sample() makes, timed one by one (median of 5 runs each)"""
    gd, gk = -(-discard // walkers), max(1, -(-nsamples // walkers))
    acc = {"create": [], "init": [], "discard": [], "keep": [], "close": []}
    for _ in range(5):
        t = [time.perf_counter()]
        ens = k.AisEnsemble(m, walkers, seed=1)
        t.append(time.perf_counter())
        ens.init()
        t.append(time.perf_counter())
        if gd:
            ens.advance(gd, nt)
        t.append(time.perf_counter())
        ens.advance(gk, nt, collect=True)
        t.append(time.perf_counter())
        ens.close()
        t.append(time.perf_counter())
        for i, key in enumerate(acc):
            acc[key].append(t[i + 1] - t[i])
    return {key: round(sorted(v)[2] * 1e3, 3) for key, v in acc.items()}


def main():
    m = model()
    with_oracle = "--oracle" in sys.argv
    if with_oracle:
        from oracle import oracle as orc
    for walkers, nsamples, discard, nt in SHAPES:
        kw = dict(ntransitions=nt, discard_initial=discard, seed=1, return_array=True)
        gens = -(-discard // walkers) + max(1, -(-nsamples // walkers))
        row = {"AIS": walkers, "samples": nsamples, "discard_initial": discard, "ntransitions": nt, "generations": gens}
        for drv in ("small", "halves"):
            os.environ["KABC_AIS_SMALL"] = "1" if drv == "small" else "0"
            k.sample(m, k.AIS(walkers), nsamples, **kw)
            w = med(lambda: k.sample(m, k.AIS(walkers), nsamples, **kw))
            row[drv] = {"wall_ms": round(w * 1e3, 3), "us_per_half_generation": round(w * 1e6 / (2 * gens), 2),
                        "phases_ms": phases(m, walkers, nsamples, discard, nt)}
        if walkers == 12:   # 50 chains in one handle (MCMCThreads): no slower than one?
            os.environ["KABC_AIS_SMALL"] = "1"
            k.sample(m, k.AIS(walkers), k.MCMCThreads(), nsamples, 50, **kw)
            w = med(lambda: k.sample(m, k.AIS(walkers), k.MCMCThreads(), nsamples, 50, **kw))
            row["small_50_chains_wall_ms"] = round(w * 1e3, 3)
        if with_oracle:
            nsteps = gens * walkers   # one generation = N reference step() calls
            o = orc.OracleAIS(m, walkers, seed=1).init()
            t0 = time.perf_counter()
            o.steps_serial(nsteps, nt, collect=True)
            row["oracle_one_core_wall_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
