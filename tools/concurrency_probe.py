"""Several processes share one GPU: three run smc C4 (multi-workgroup select with its
device-wide spin barrier) in a loop while a fourth runs the AIS bench kernel.  Checks
that every smc result stays bit-identical to the first one and nothing stalls."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMC = r"""
import sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tools')
import numpy as np
import kissabc_jl_amd as k
from smc_c4_probe import c4_problem
prior, cost = c4_problem()
kw = dict(nparticles=32768, alpha=0.95, epstol=0.05, seed=1)
ref = k.smc(prior, cost, return_array=True, **kw)
t0 = time.time(); n = 0
while time.time() - t0 < 20:
    r = k.smc(prior, cost, return_array=True, **kw)
    assert r.eps == ref.eps and np.array_equal(r.info['theta_all'], ref.info['theta_all'])
    n += 1
print('smc runs', n, 'ms each', 1e3 * (time.time() - t0) / n)
""" % (ROOT, ROOT)
AIS = r"""
import sys, time
sys.path.insert(0, %r)
import bench, kissabc_jl_amd as k
e = k.AisEnsemble(bench.build_model(k), 65536, seed=1).init()
t0 = time.time(); n = 0
while time.time() - t0 < 20:
    e.advance(20, 100); n += 20
print('ais generations', n, 'us each', 1e6 * (time.time() - t0) / n)
""" % ROOT
procs = [subprocess.Popen([sys.executable, "-c", SMC], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for _ in range(3)]
procs.append(subprocess.Popen([sys.executable, "-c", AIS], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
ok = True
for p in procs:
    try:
        out, _ = p.communicate(timeout=180)
    except subprocess.TimeoutExpired:
        p.kill()
        out, ok = "TIMEOUT", False
    print(out.strip().splitlines()[-1] if out.strip() else "(no output)", "rc", p.returncode)
    ok = ok and p.returncode == 0
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
