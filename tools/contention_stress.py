"""Determinism under GPU sharing: `nproc` checking processes of one kind run the same workload over
and over for `seconds` while background processes of another kind keep the GPU busy; every checking
process compares each result with its first one and, at the end, the odd ones with the CPU oracle.
  python tools/contention_stress.py <kind> [nproc=3] [background=ais] [seconds=20]
kinds: smc_loop (persistent loop kernel), smc_kernels (kernel-per-phase path), ais (C3-shaped
ensemble, 40 generations x 16 transitions from the same start), sharded (four emulated ranks,
pipelined exchange), trace (streamed sample trace)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kind = sys.argv[1]
nproc = int(sys.argv[2]) if len(sys.argv) > 2 else 3
background = sys.argv[3] if len(sys.argv) > 3 else "ais"
seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0

CHILD = r"""
import os, sys, time
kind, seconds, check = sys.argv[1], float(sys.argv[2]), sys.argv[3] == '1'
if kind == 'smc_kernels':
    os.environ['KABC_SMC_LOOP'] = '0'
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tools')
import numpy as np
import kissabc_jl_amd as k
import bench
if kind.startswith('smc'):
    from smc_c4_probe import c4_problem
    prior, cost = c4_problem()
    kw = dict(nparticles=32768, alpha=0.95, epstol=0.05, seed=1)
    def run():
        r = k.smc(prior, cost, return_array=True, **kw)
        return (r.eps, r.info['theta_all'])
elif kind == 'sharded':   # four emulated ranks on this GPU, pipelined exchange chunks (second stream + events)
    os.environ['KABC_EXCHANGE_CHUNKS'] = '3'
    model = bench.build_model(k)
    def run():
        g = k.EnsembleGroup(model, 1 << 16, seed=1, devices=[0] * 4, backend='p2p').init()
        g.advance(12, 8)
        x = g.ensemble(3).copy()
        st = g.stats()
        g.close()
        return (float(st['accepted']), x)
elif kind == 'trace':     # streamed sample trace: device chunks in rotation, drain thread, copy stream
    model = bench.build_model(k)
    def run():
        e = k.AisEnsemble(model, 1 << 14, seed=1).init()
        tr = e.advance(48, 4, collect=True)
        e.close()
        return (float(tr[-1].sum()), np.ascontiguousarray(tr).reshape(-1, tr.shape[-1]))
else:
    model = bench.build_model(k)
    def run():
        e = k.AisEnsemble(model, 65536, seed=1).init()
        e.advance(40, 16)
        st = e.state()
        e.close()
        return (float(st[1].sum()), st[0])
first = run()
t0 = time.time(); n = 1; odd = 0
while time.time() - t0 < seconds:
    r = run()
    if check and not (r[0] == first[0] and np.array_equal(r[1], first[1])):
        odd += 1
    n += 1
print(kind, 'runs', n, 'ms each', round(1e3 * seconds / n, 2), 'DIFFERING FROM THE FIRST: %%d' %% odd if odd else 'all identical')
sys.exit(1 if odd else 0)
""" % (ROOT, ROOT)
procs = [subprocess.Popen([sys.executable, "-c", CHILD, kind, str(seconds), "1"], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True) for _ in range(nproc)]
procs.append(subprocess.Popen([sys.executable, "-c", CHILD, background, str(seconds), "0"], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True))
ok = True
for p in procs:
    try:
        out, _ = p.communicate(timeout=seconds * 6 + 120)
    except subprocess.TimeoutExpired:
        p.kill()
        out, ok = "TIMEOUT", False
    print(out.strip().splitlines()[-1] if out.strip() else "(no output)", "rc", p.returncode)
    ok = ok and p.returncode == 0
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
