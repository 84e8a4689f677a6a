"""examples/abi_demo.c: the C ABI used from plain C99 (what a Julia ccall / cgo / JNI
binding does).  CPU: the headers compile as C and the program links against the library.
GPU: its results equal the Python mirror's for the same seeds, digit for digit."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "kissabc.jl_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "abi_demo")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "abi_demo.c"), "-o", exe, "-L", LIBDIR, "-lkabc_hip",
           "-lpthread", f"-Wl,-rpath,{LIBDIR}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_abi_demo_compiles_as_c99_and_links(tmp_path):
    _build(tmp_path)


@pytest.mark.gpu
def test_abi_demo_matches_python_mirror(tmp_path, k, gpu_ctx):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = dict(kv.split("=") for kv in r.stdout.strip().splitlines()[-1].split())
    prior = k.Factored(k.Normal(0, 5), k.Normal(0, 5))
    cost = k.costs.GaussDist([1.0, -0.5])
    ens = k.AisEnsemble(k.ApproxKernelizedPosterior(prior, cost, 0.1), 4096, seed=7).init()
    ens.advance(300, 1)
    tr = ens.advance(200, 1, collect=True).reshape(-1, 2)
    n = tr.shape[0]
    assert float(got["mean0"]) == float(np.cumsum(tr[:, 0])[-1]) / n     # same sequential sum
    assert float(got["mean1"]) == float(np.cumsum(tr[:, 1])[-1]) / n
    assert (float(got["last0"]), float(got["last1"])) == (tr[-1, 0], tr[-1, 1])
    assert int(got["proposals"]) == 4096 * 500
    # C2's analytic posterior mean c * 2500 / 2501, loose here: 200 generations
    assert abs(float(got["mean0"]) - 1.0 * 2500 / 2501) < 5e-3
    assert abs(float(got["mean1"]) + 0.5 * 2500 / 2501) < 5e-3
    s = k.smc(prior, cost, nparticles=2000, alpha=0.9, epstol=0.05, seed=11, return_array=True)
    assert float(got["smc_eps"]) == s.eps and int(got["smc_iterations"]) == s.info["iterations"]
    assert int(got["smc_alive"]) == s.info["n_alive"]
    assert float(got["smc_sum0"]) == float(np.cumsum(s.info["theta_all"][:, 0])[-1])
    assert int(got["version"]) == 321 and int(got["sharded_equal"]) == 1
    assert int(got["smc_sharded_equal"]) == 1     # kabc_smc_run_dist_mode, particles sharded over two ranks
