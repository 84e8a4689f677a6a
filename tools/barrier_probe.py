"""How long does each wave of an AIS batch wait at the workgroup barriers?
KABC_ABLATE=128: every wave writes (lifetime, time inside __syncthreads) in s_memtime
ticks (100 MHz) into the debug records; printed as means over the 512 batches of the
C3 launch for the consumer (wave 0) and the producers (waves 1-3)."""
import os
import sys

os.environ["KABC_PROBES"] = "1"   # the library variant with the probes compiled in

os.environ["KABC_ABLATE"] = "128"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import kissabc_jl_amd as k  # noqa: E402

nt = int(os.environ.get("KABC_NT", "16"))
e = k.AisEnsemble(bench.build_model(k), 65536, seed=1).init()
e.advance(2, nt)
e.set_debug(nt)
e.advance(1, nt)
d = e.get_debug(nt).reshape(65536, -1)
t = d[0:32768:64, 8:16].astype(np.float64).reshape(-1, 4, 2)   # [block][wave][life, barrier]
print("nt", nt, "ticks are s_memtime units")
for w in range(4):
    life, bar = t[:, w, 0].mean(), t[:, w, 1].mean()
    print(f"wave {w} ({'consumer' if w == 0 else 'producer'}): life {life:9.1f}  at barriers {bar:9.1f}  = {100 * bar / life:5.1f} %")
