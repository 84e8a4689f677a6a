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
first = k.smc(prior, cost, return_array=True, **kw)   # all processes start at once
t0 = time.time(); n = 1; odd = []
while time.time() - t0 < 20:
    r = k.smc(prior, cost, return_array=True, **kw)
    if not (r.eps == first.eps and np.array_equal(r.info['theta_all'], first.info['theta_all'])):
        odd.append((n, round(time.time() - t0, 3), r))
    n += 1
# the CPU oracle's answer says WHICH runs were wrong (computed last: the processes hit the GPU together from t = 0)
from oracle import oracle as orc
ref = orc.smc(prior, cost, **kw)
def same(r):
    return r.eps == ref['eps'] and np.array_equal(r.info['theta_all'], ref['theta_all'])
bad = []
if not same(first):
    bad.append(('first call', first.info['iterations'], first.eps))
for m, t, r in odd:
    if not same(r):
        d = np.flatnonzero((r.info['theta_all'] != ref['theta_all']).any(axis=1))
        bad.append((m, t, r.info['iterations'], r.eps, len(d), d[:4].tolist()))
print('smc runs', n, 'ms each', 1e3 * 20 / n, 'MISMATCHES' if bad else 'all equal to the oracle', bad[:6], 'differing from the first:', len(odd))
sys.exit(1 if bad else 0)
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
