"""Timing probe for the AIS half-generation kernel under KABC_ABLATE (1 = no
consumer, 2 = no producers after the prologue, 4 = no prologue).  Results are
WRONG under ablation; only the kernel duration is of interest."""
import os
import sys

os.environ["KABC_PROBES"] = "1"   # the library variant with the probes compiled in

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import kissabc_jl_amd as k  # noqa: E402

nt = int(os.environ.get("KABC_NT", "16"))
m = bench.build_model(k)
e = k.AisEnsemble(m, 65536, seed=1).init()
e.advance(5, nt)
ts = []
for _ in range(5):
    e.set_timing(800, stride=8)   # (a pair per launch adds ~2.5 us of marker packets to the figure)
    e.advance(400, nt)
    ts.append(e.kernel_ms()[0] * 1e3)
print("ablate", os.environ.get("KABC_ABLATE", "0"), "nt", nt, "kernel us (hipEvent, median of 5 x 800 launches)",
      round(sorted(ts)[2], 3))
