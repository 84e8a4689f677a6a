"""README.md:31-57 as AIS(10) (BASELINE.json configs[0]) under KABC_ABLATE (1 = no consumer,
2 = no producers after the prologue): which role's chain bounds the tiny-ensemble launch.
Results are WRONG under ablation; only the kernel duration is of interest."""
import os
import sys

os.environ["KABC_PROBES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import kissabc_jl_amd as k  # noqa: E402

nt = int(os.environ.get("KABC_NT", "100"))
N = int(os.environ.get("KABC_N", "10"))
e = k.AisEnsemble(bench.readme_problem(k), N, seed=1).init()
e.advance(5, nt)
e.set_timing(200, stride=1)
e.advance(100, nt)
print("N", N, "ablate", os.environ.get("KABC_ABLATE"), "nt", nt, "kernel ms (hipEvent)", e.kernel_ms())
