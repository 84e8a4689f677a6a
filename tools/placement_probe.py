"""Where do the four waves of an AIS batch land?  KABC_ABLATE=64 makes every wave of
the half-generation kernel write its HW_ID into the debug records; this prints, for
the C3 launch, how consumers (wave 0) and producers (waves 1-3) share SIMDs."""
import collections
import os
import sys

os.environ["KABC_PROBES"] = "1"   # the library variant with the probes compiled in

os.environ["KABC_ABLATE"] = "64"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import kissabc_jl_amd as k  # noqa: E402

nt = 16
e = k.AisEnsemble(bench.build_model(k), 65536, seed=1).init()
e.advance(2, nt)
e.set_debug(nt)
e.advance(1, nt)
d = e.get_debug(nt)            # [N][nt][6]; rows 0..32767 = half 0
hw = d[0:32768:64, 0, 0:4].astype(np.uint32)     # [block][wave]
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
se = (hw >> 13) & 7
wave_id = hw & 15
print("blocks", hw.shape[0])
print("SIMD of (wave0, wave1, wave2, wave3), most common patterns:")
for pat, n in collections.Counter(map(tuple, simd.tolist())).most_common(8):
    print("  ", pat, n)
print("wave0 SIMD histogram:", np.bincount(simd[:, 0], minlength=4).tolist())
print("HW wave slot ids of wave0:", np.bincount(wave_id[:, 0], minlength=16).tolist())
