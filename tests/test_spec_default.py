"""The DEFAULT path of a model that can be specialised (include/kabc.h, "THE DEFAULT"): an entry
point never waits for the compiler -- it starts on the prebuilt kernels, a detached worker process
compiles the model's own unit into the on-disk cache, an AIS handle switches kernels at a launch
boundary, the run-to-completion entry points take them from their next call on.  The reference
gets the same from Julia's per-type compilation of logpdf(::Factored) (src/priors.jl:11,30-36).
Every kernel involved computes the same bits, so every comparison below is exact."""
import ctypes as C
import os
import time

import numpy as np
import pytest


def _counters(k):
    out = (C.c_uint64 * 4)()
    k._lib.load().kabc_spec_counters(out)
    return dict(zip(("spawned", "loaded", "failed", "cache_hits"), list(out)))


def _fresh_cache(monkeypatch, tmp_path):
    monkeypatch.setenv("KABC_RTC_CACHE_DIR", str(tmp_path))
    monkeypatch.delenv("KABC_SPECIALIZE", raising=False)


def _wait_for(pred, timeout=240.0, step=0.2):
    t0 = time.time()
    while time.time() - t0 < timeout:
        if pred():
            return True
        time.sleep(step)
    return False


def _model(k, a=2.0, b=3.0):
    prior = k.Factored(k.Normal(0, 5), k.Beta(a, b))
    return k.ApproxKernelizedPosterior(prior, k.costs.GaussDist([1.0, 0.5]), 0.1)


# ---- no GPU needed: the worker process and its files ----------------------------------------
def test_worker_delivers_into_the_cache(k, monkeypatch, tmp_path):
    """kabc_prefetch_model returns at once; the detached worker writes the code object."""
    _fresh_cache(monkeypatch, tmp_path)
    lib = k._lib.load()
    cm = _model(k, 2.25, 3.5).to_c()
    c0 = _counters(k)
    t0 = time.time()
    k._lib.check(lib.kabc_prefetch_model(C.byref(cm), 1))
    assert time.time() - t0 < 0.5
    assert _counters(k)["spawned"] == c0["spawned"] + 1
    names = lambda: sorted(os.listdir(tmp_path))
    assert any(n.endswith(".lock") for n in names())
    assert _wait_for(lambda: any(n.endswith(".co") for n in names()))
    assert _wait_for(lambda: not any(n.endswith((".lock", ".job")) for n in names()), timeout=10)
    # a box prior is not eligible: nothing is started
    box = k.ApproxKernelizedPosterior(k.Factored(k.Uniform(0, 1), k.Uniform(0, 2)), k.costs.GaussDist([0.5, 0.5]), 1.0)
    c1 = _counters(k)
    k._lib.check(lib.kabc_prefetch_model(C.byref(box.to_c()), 1))
    assert _counters(k) == c1


def test_worker_reports_a_failed_compilation(k, monkeypatch, tmp_path):
    _fresh_cache(monkeypatch, tmp_path)
    monkeypatch.setenv("KABC_SPEC_INJECT_ERROR", "1")
    lib = k._lib.load()
    cm = _model(k, 2.5, 3.25).to_c()
    k._lib.check(lib.kabc_prefetch_model(C.byref(cm), 1))
    errs = lambda: [n for n in os.listdir(tmp_path) if n.endswith(".err")]
    assert _wait_for(lambda: len(errs()) == 1)
    assert "KABC_SPEC_INJECT_ERROR" in open(os.path.join(tmp_path, errs()[0])).read()
    c0 = _counters(k)
    time.sleep(0.01)
    k._lib.check(lib.kabc_prefetch_model(C.byref(cm), 1))   # the poll sees the .err: failed for good
    assert _counters(k)["failed"] == c0["failed"] + 1
    k._lib.check(lib.kabc_prefetch_model(C.byref(cm), 1))
    assert _counters(k)["failed"] == c0["failed"] + 1 and _counters(k)["spawned"] == c0["spawned"]


def test_missing_worker_or_switch_off_starts_nothing(k, monkeypatch, tmp_path):
    _fresh_cache(monkeypatch, tmp_path)
    lib = k._lib.load()
    cm = _model(k, 2.75, 3.75).to_c()
    c0 = _counters(k)
    monkeypatch.setenv("KABC_RTC_WORKER", "/nonexistent/kabc_rtc_worker")
    k._lib.check(lib.kabc_prefetch_model(C.byref(cm), 1))
    monkeypatch.delenv("KABC_RTC_WORKER")
    monkeypatch.setenv("KABC_SPECIALIZE", "0")
    k._lib.check(lib.kabc_prefetch_model(C.byref(cm), 1))
    assert _counters(k) == c0 and os.listdir(tmp_path) == []


def test_workers_are_bounded(k, monkeypatch, tmp_path):
    """KABC_RTC_WORKERS compilations at a time per cache directory, the rest deferred."""
    _fresh_cache(monkeypatch, tmp_path)
    monkeypatch.setenv("KABC_RTC_WORKERS", "1")
    lib = k._lib.load()
    c0 = _counters(k)
    models = [_model(k, 4.0 + 0.125 * i, 5.0).to_c() for i in range(3)]
    for cm in models:
        k._lib.check(lib.kabc_prefetch_model(C.byref(cm), 1))
    assert _counters(k)["spawned"] == c0["spawned"] + 1
    assert len([n for n in os.listdir(tmp_path) if n.endswith(".lock")]) <= 1

    def all_there():
        for cm in models:   # (a deferred unit is started by a later request, when there is room)
            k._lib.check(lib.kabc_prefetch_model(C.byref(cm), 1))
        return len([n for n in os.listdir(tmp_path) if n.endswith(".co")]) == 3
    assert _wait_for(all_there, timeout=400, step=0.1)
    assert _counters(k)["spawned"] == c0["spawned"] + 3


# ---- on the GPU -----------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("N", [256, 1024])   # the one-workgroup driver of small ensembles / a launch per half-generation
def test_ais_switches_kernels_across_a_run_bit_exact(k, orc, gpu_ctx, monkeypatch, tmp_path, N):
    """one ensemble from the prebuilt kernels over the switch to its own kernels = the oracle"""
    _fresh_cache(monkeypatch, tmp_path)
    model = _model(k, 2.0 + N / 4096.0)   # (a model of its own per case: nothing of it is compiled yet)
    nt, seed = 3, 21
    t0 = time.time()
    ens = k.AisEnsemble(model, N, seed=seed).init()
    first = [ens.advance(1, nt, collect=True)]
    first_call_s = time.time() - t0
    assert ens.spec_state()[0] == "pending"
    assert first_call_s < 5.0    # (the context exists already; no compilation on this path)
    got = first
    t0 = time.time()
    while ens.spec_state()[0] == "pending" and time.time() - t0 < 240:
        time.sleep(0.05)
        got.append(ens.advance(1, nt, collect=True))
    state, before = ens.spec_state()
    # two launches per generation -- one per advance call on the one-workgroup driver (N <= 512)
    per = 1 if ens.driver == "small" else 2
    assert state == "active" and before == per * (len(got) - 1) > 0
    for _ in range(4):
        got.append(ens.advance(1, nt, collect=True))
    got = np.concatenate(got)
    ref = orc.OracleAIS(model, N, seed=seed).init().generations_sync(got.shape[0], nt)
    assert np.array_equal(got, ref)
    assert ens.stats() is not None
    # the same model again: its unit is in the cache, loaded at create
    ens2 = k.AisEnsemble(model, N, seed=seed).init()
    assert ens2.spec_state() == ("active", 0)
    assert np.array_equal(ens2.advance(3, nt, collect=True), ref[:3])


@pytest.mark.gpu
def test_ais_stays_on_prebuilt_when_the_compilation_fails(k, orc, gpu_ctx, monkeypatch, tmp_path):
    _fresh_cache(monkeypatch, tmp_path)
    monkeypatch.setenv("KABC_SPEC_INJECT_ERROR", "1")
    model = _model(k, 2.125, 3.125)
    N, nt, seed = 200, 2, 5
    ens = k.AisEnsemble(model, N, seed=seed).init()
    got = [ens.advance(1, nt, collect=True)]
    t0 = time.time()
    while ens.spec_state()[0] == "pending" and time.time() - t0 < 240:
        time.sleep(0.05)
        got.append(ens.advance(1, nt, collect=True))
    assert ens.spec_state() == ("failed", -1)
    got.append(ens.advance(2, nt, collect=True))
    got = np.concatenate(got)
    assert np.array_equal(got, orc.OracleAIS(model, N, seed=seed).init().generations_sync(got.shape[0], nt))
    # no worker at all: the prebuilt kernels, state "none"
    monkeypatch.delenv("KABC_SPEC_INJECT_ERROR")
    monkeypatch.setenv("KABC_RTC_WORKER", "/nonexistent/kabc_rtc_worker")
    m2 = _model(k, 2.0625, 3.0625)
    e2 = k.AisEnsemble(m2, N, seed=seed).init()
    assert e2.spec_state() == ("none", -1)
    assert np.array_equal(e2.advance(2, nt, collect=True),
                          orc.OracleAIS(m2, N, seed=seed).init().generations_sync(2, nt))


@pytest.mark.gpu
def test_user_family_prior_never_falls_to_prebuilt(k, orc, gpu_ctx, monkeypatch, tmp_path):
    """a prior with a user family has no prebuilt kernels: when its specialisation fails -- blocking
    (KABC_SPECIALIZE=1) or in the worker -- the generic unit of the family serves, not NaNs"""
    prior = k.Factored(k.Poisson(3.5), k.Normal(0, 2), k.Laplace(0.0, 1.5))
    model = k.ApproxKernelizedPosterior(prior, k.costs.GaussDist([3.0, 0.5, 0.2]), 1.0)
    N, nt, seed = 300, 3, 4
    ref = orc.OracleAIS(model, N, seed=seed).init().generations_sync(3, nt)
    monkeypatch.setenv("KABC_RTC_CACHE_DIR", str(tmp_path))
    monkeypatch.setenv("KABC_SPEC_INJECT_ERROR", "1")
    for mode in ("1", None):
        if mode:
            monkeypatch.setenv("KABC_SPECIALIZE", mode)
        else:
            monkeypatch.delenv("KABC_SPECIALIZE")
        ens = k.AisEnsemble(model, N, seed=seed).init()
        assert np.array_equal(ens.advance(3, nt, collect=True), ref)
        for run, oref in ((k.smc, orc.smc),):
            kw = dict(nparticles=600, alpha=0.9, epstol=0.4)
            g = run(prior, k.costs.GaussDist([3.0, 0.5, 0.2]), seed=2, return_array=True, **kw)
            r = oref(prior, k.costs.GaussDist([3.0, 0.5, 0.2]), seed=2, **kw)
            assert g.eps == r["eps"] and np.array_equal(g.info["theta_all"], r["theta_all"])


@pytest.mark.gpu
@pytest.mark.parametrize("nparticles", [100, 3000, 70000])
def test_smc_default_path_takes_its_own_kernels_from_the_next_call(k, orc, gpu_ctx, monkeypatch, tmp_path,
                                                                    nparticles):
    """the three smc drivers (one workgroup / persistent loop / kernel per phase)"""
    _fresh_cache(monkeypatch, tmp_path)
    prior = k.Factored(k.Gamma(2.5, 0.7 + nparticles * 1e-6), k.LogNormal(0.3, 0.6), k.Beta(2.0, 3.0))
    cost = k.costs.NormShell(2.0)
    kw = dict(nparticles=nparticles, alpha=0.9, epstol=0.2)
    ref = orc.smc(prior, cost, seed=5, **kw)
    c0 = _counters(k)
    first = k.smc(prior, cost, seed=5, return_array=True, **kw)
    c1 = _counters(k)
    assert c1["spawned"] > c0["spawned"] and c1["loaded"] == c0["loaded"]
    assert _wait_for(lambda: not any(n.endswith((".lock", ".job")) for n in os.listdir(tmp_path))
                     and any(n.endswith(".co") for n in os.listdir(tmp_path)), timeout=400)
    assert not any(n.endswith(".err") for n in os.listdir(tmp_path))
    second = k.smc(prior, cost, seed=5, return_array=True, **kw)
    assert _counters(k)["loaded"] > c1["loaded"]
    for r in (first, second):
        assert r.info["iterations"] == ref["iterations"] and r.info["log"] == ref["log"]
        assert r.eps == ref["eps"] and np.array_equal(r.info["theta_all"], ref["theta_all"])
        assert np.array_equal(r.C, ref["C"])


@pytest.mark.gpu
def test_abcde_and_pfilter_default_path(k, orc, gpu_ctx, monkeypatch, tmp_path):
    _fresh_cache(monkeypatch, tmp_path)
    prior = k.Factored(k.Gamma(2.0, 0.9), k.Normal(0.0, 2.0), k.Beta(2.0, 2.5))
    cost = k.costs.GaussDist([1.5, 0.3, 0.4])
    ra = orc.abcde(prior, cost, 0.3, seed=9, nparticles=300, generations=25)
    rp = orc.pfilter(prior, cost, 300, seed=4, max_iters=12)
    c0 = _counters(k)
    for attempt in range(2):
        ga = k.ABCDE(prior, cost, 0.3, seed=9, return_array=True, nparticles=300, generations=25)
        gp = k.pfilter(prior, cost, 300, seed=4, return_array=True, max_iters=12)
        assert np.array_equal(ga.P, ra["P"]) and np.array_equal(ga.C, ra["C"])
        assert np.array_equal(gp.P, rp["P"]) and np.array_equal(gp.C, rp["C"])
        if attempt == 0:
            assert _wait_for(lambda: len([n for n in os.listdir(tmp_path) if n.endswith(".co")]) >= 2
                             and not any(n.endswith((".lock", ".job")) for n in os.listdir(tmp_path)), timeout=400)
    assert _counters(k)["loaded"] >= c0["loaded"] + 2


def test_cache_directory_is_bounded(k, monkeypatch, tmp_path):
    """KABC_RTC_CACHE_MB: the oldest code objects go before a new one is stored; leftovers of
    workers that died (old lock / job files) go on the same occasion."""
    _fresh_cache(monkeypatch, tmp_path)
    old = tmp_path / "kabc_0000000000000000ffffffffffffffff.co"
    old.write_bytes(b"x" * (600 * 1024))
    stale = tmp_path / "kabc_1111111111111111ffffffffffffffff.co.lock"
    stale.write_bytes(b"")
    past = time.time() - 3600
    os.utime(old, (past, past))
    os.utime(stale, (past, past))
    monkeypatch.setenv("KABC_RTC_CACHE_MB", "0.5")
    lib = k._lib.load()
    cm = _model(k, 6.5, 7.25).to_c()
    k._lib.check(lib.kabc_prefetch_model(C.byref(cm), 1))
    assert _wait_for(lambda: len([n for n in os.listdir(tmp_path) if n.endswith(".co")]) >= 1
                     and not any(n.endswith(".job") for n in os.listdir(tmp_path)), timeout=240)
    names = os.listdir(tmp_path)
    assert old.name not in names and stale.name not in names
    assert len([n for n in names if n.endswith(".co")]) == 1


SHARED_CHILD = r"""
import os, sys, json, time
import ctypes as C
import numpy as np
sys.path.insert(0, {root!r})
import kissabc_jl_amd as k
prior = k.Factored(k.Normal(0, 5), k.Beta(2.75, 3.25))
model = k.ApproxKernelizedPosterior(prior, k.costs.GaussDist([1.0, 0.5]), 0.1)
# all children start their first call together
while time.time() < {go!r}:
    time.sleep(0.005)
ens = k.AisEnsemble(model, 256, seed=9).init()
out = [ens.advance(1, 3, collect=True)]
t0 = time.time()
while ens.spec_state()[0] == "pending" and time.time() - t0 < 240:
    time.sleep(0.05)
    out.append(ens.advance(1, 3, collect=True))
state = ens.spec_state()[0]
for _ in range(3):
    out.append(ens.advance(1, 3, collect=True))
cnt = (C.c_uint64 * 4)()
k._lib.load().kabc_spec_counters(cnt)
np.save({out!r}, np.concatenate(out))
print(json.dumps(dict(state=state, spawned=int(cnt[0]), generations=len(out))), flush=True)
"""


@pytest.mark.gpu
def test_processes_sharing_a_cold_cache_compile_a_unit_once(k, orc, gpu_ctx, tmp_path):
    """four processes meet the same model at the same moment with an empty cache directory: the lock
    file lets ONE of them start the worker, all of them switch to the unit it delivers, every trajectory
    is the oracle's, and nothing is left behind in the cache but the code object"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cache = tmp_path / "cache"
    cache.mkdir()
    env = {k_: v for k_, v in os.environ.items() if k_ != "KABC_SPECIALIZE"}
    env["KABC_RTC_CACHE_DIR"] = str(cache)
    go = time.time() + 8.0
    procs = [subprocess.Popen([sys.executable, "-c",
                               SHARED_CHILD.format(root=root, go=go, out=str(tmp_path / f"x{i}.npy"))],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(4)]
    res = []
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-2000:]
        res.append(json.loads(so.strip().splitlines()[-1]))
    assert all(r["state"] == "active" for r in res), res
    assert sum(r["spawned"] for r in res) == 1, res          # one worker for the four of them
    prior = k.Factored(k.Normal(0, 5), k.Beta(2.75, 3.25))
    model = k.ApproxKernelizedPosterior(prior, k.costs.GaussDist([1.0, 0.5]), 0.1)
    gmax = max(r["generations"] for r in res)
    ref = orc.OracleAIS(model, 256, seed=9).init().generations_sync(gmax, 3)
    for i, r in enumerate(res):
        assert np.array_equal(np.load(tmp_path / f"x{i}.npy"), ref[:r["generations"]]), i
    left = sorted(os.listdir(cache))
    assert len(left) == 1 and left[0].endswith(".co"), left
