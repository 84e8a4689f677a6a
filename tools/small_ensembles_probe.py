"""The reference's ACTUAL usage is tiny ensembles (AIS(10)...AIS(500), test/runtests.jl).
Wall times of the device path next to the CPU oracle's serial restatement for
  * the README example (README.md:31-57; BASELINE.json configs[0]): AIS(10), 1000 samples,
    ntransitions = 100, a simulator of 1000 normals per cost evaluation;
  * the MCMCThreads testset (test/runtests.jl:88-104): 50 chains x 100 samples x AIS(12),
    as ONE batch handle (chain = a grid dimension) and chain after chain;
  * 50 README chains as one batch."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402
from kissabc_jl_amd.api import chain_seeds  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def timed(f, reps=3):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], r


rng = np.random.default_rng(0)
tdata = rng.normal(2.0, 0.04, 1000)
prior = k.Factored(k.Uniform(1, 3), k.Truncated(k.Normal(0, 0.1), 0, 100))
readme = k.ApproxKernelizedPosterior(prior, k.costs.NormalMeanStdSim(1000, tdata.mean(), tdata.std(ddof=1)), 0.005)
dirac = k.ApproxKernelizedPosterior(k.Normal(1, 0.2), k.costs.DiracSq(1.5), 0.001)
out = {}

t_dev, res = timed(lambda: k.sample(readme, k.AIS(10), 1000, ntransitions=100, seed=1, return_array=True))


def readme_oracle():
    o = orc.OracleAIS(readme, 10, seed=1).init()
    return o.steps_serial(1000, 100)


t_orc, ro = timed(readme_oracle, reps=1)
out["readme_ais10_1000_samples_nt100"] = {
    "device_ms": t_dev * 1e3, "oracle_serial_1core_ms": t_orc * 1e3,
    "transitions": 1000 * 100, "device_mean": res.mean(0).tolist(), "oracle_mean": ro.mean(0).tolist(),
    "reference_documented": "2.0 ± 0.018, 0.0395 ± 0.00093 in ~2 s incl. JIT (README.md:57-66)"}

kw = dict(ntransitions=1, discard_initial=600)
t_b, _ = timed(lambda: k.sample(dirac, k.AIS(12), k.MCMCThreads(), 100, 50, seed=1, return_array=True, **kw))
t_s, _ = timed(lambda: [k.sample(dirac, k.AIS(12), 100, seed=s, return_array=True, **kw)
                        for s in chain_seeds(1, 50)], reps=1)


def chains_oracle():
    for c in range(50):
        o = orc.OracleAIS(dirac, 12, seed=100 + c).init()
        o.steps_serial(600, 1, collect=False)
        o.steps_serial(100, 1)


t_o, _ = timed(chains_oracle, reps=1)
out["mcmcthreads_50x100xAIS12"] = {"device_batch_ms": t_b * 1e3, "device_chain_after_chain_ms": t_s * 1e3,
                                   "oracle_serial_1core_ms": t_o * 1e3}

t_rb, _ = timed(lambda: k.sample(readme, k.AIS(10), k.MCMCThreads(), 1000, 50, ntransitions=100, seed=1,
                                 return_array=True), reps=2)
out["readme_50_chains_one_batch"] = {"device_batch_ms": t_rb * 1e3,
                                     "per_chain_ms": t_rb * 1e3 / 50, "transitions": 50 * 1000 * 100}
print(json.dumps(out))
