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
e.set_timing(100)
e.advance(50, nt)
print("ablate", os.environ.get("KABC_ABLATE"), "nt", nt, "kernel ms (hipEvent)", e.kernel_ms())
