"""Compile (hipRTC, no GPU needed) the specialised kernels bench.py's `by_prior_class` leg asks for, so
that their code objects sit in the on-disk cache (kissabc.jl_amd/lib/rtc_cache, which travels with
the tree) and a GPU run loads them in a millisecond instead of compiling for 3-20 s each.
Prints the time of every compilation.  Called by __graft_entry__.build()."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("KABC_NO_TORCH_PRELOAD", "1")
import kissabc_jl_amd as k  # noqa: E402
import bench  # noqa: E402

# entries of earlier builds are dead weight (the key holds a fingerprint of every header): the tree
# that travels to the GPU box carries the current ones only
cache = os.path.join(os.path.dirname(k.LIB_PATH), "rtc_cache")
if os.path.isdir(cache) and not os.environ.get("KABC_RTC_CACHE_DIR"):
    for f in os.listdir(cache):
        if f.startswith("kabc_") and f.endswith(".co"):
            os.remove(os.path.join(cache, f))

out = {}
for name, model, N, D in bench.prior_class_problems(k):
    t0 = time.perf_counter()
    h = k.compile_model(model, families=1)
    out[name] = round(time.perf_counter() - t0, 2)
    if h:
        k._lib.check(k._lib.load().kabc_model_release(h))
print(json.dumps({"hiprtc_compile_or_cache_s": out}))
