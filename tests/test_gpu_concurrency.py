"""Device-wide barriers under sharing: smc() on C4 -- the persistent cooperative loop kernel
and the kernel-per-phase path with its cooperative select kernel -- while a 2^20-walker AIS
ensemble keeps every CU busy from another stream of the same process, and while a second
smc() runs from a third.  Results must equal the quiet runs bit for bit, and nothing may hang:
the barriers spin with a bound and co-residency comes from hipLaunchCooperativeKernel
(an ordinary launch gave no such guarantee)."""
import os
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("path", ["loop", "kernels"])
def test_smc_c4_while_ais_saturates_the_device(k, gpu_ctx, monkeypatch, path):
    from smc_c4_probe import c4_problem
    monkeypatch.setenv("KABC_SMC_LOOP", "1" if path == "loop" else "0")
    prior, cost = c4_problem()
    kw = dict(nparticles=32768, alpha=0.95, epstol=0.05, seed=1, return_array=True)
    quiet = k.smc(prior, cost, **kw)

    big = k.ApproxKernelizedPosterior(k.Factored(*[k.Uniform(-5, 5)] * 8), k.costs.Rosenbrock(), 1.0)
    ctx_ais, ctx_smc2 = k.Context(0), k.Context(0)
    ens = k.AisEnsemble(big, 1 << 20, seed=3, ctx=ctx_ais).init()
    stop = threading.Event()
    done = {"gens": 0, "err": None, "smc2": None}

    def hammer():
        try:
            while not stop.is_set():
                ens.advance(4, 16)
                done["gens"] += 4
        except Exception as e:      # pragma: no cover
            done["err"] = e

    def second_smc():
        try:
            done["smc2"] = [k.smc(prior, cost, ctx=ctx_smc2, **kw) for _ in range(3)]
        except Exception as e:      # pragma: no cover
            done["err"] = e

    th, t2 = threading.Thread(target=hammer), threading.Thread(target=second_smc)
    th.start()
    t2.start()
    try:
        runs = [k.smc(prior, cost, **kw) for _ in range(4)]
    finally:
        stop.set()
        th.join(120)
        t2.join(300)
    assert done["err"] is None and done["gens"] > 0 and not th.is_alive() and not t2.is_alive()
    for r in runs + done["smc2"]:
        assert r.eps == quiet.eps and r.info["iterations"] == quiet.info["iterations"]
        assert np.array_equal(r.info["theta_all"], quiet.info["theta_all"])
    ens.close()
