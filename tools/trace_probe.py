"""Sample-trace streaming rate of kabc_ais_advance(out_samples) at C3:
generations/s and host GB/s with a pinned vs a pageable destination, beside the
no-trace rate.  Usage: python tools/trace_probe.py [--gens 128] [--nt 16]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kissabc_jl_amd as k  # noqa: E402
from kissabc_jl_amd import _lib  # noqa: E402
from kissabc_jl_amd.api import AisEnsemble  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gens", type=int, default=128)
    ap.add_argument("--nt", type=int, default=16)
    ap.add_argument("--n", type=int, default=65536)
    a = ap.parse_args()
    D = 8
    model = k.ApproxKernelizedPosterior(k.Factored(*[k.Uniform(-5, 5)] * D), k.costs.Rosenbrock(), 1.0)
    out = {"N": a.n, "D": D, "ntransitions": a.nt, "generations": a.gens}
    ref = None
    for mode in ("none", "pinned", "pageable"):
        os.environ["KABC_PINNED_TRACE"] = "0" if mode == "pageable" else "1"
        ens = AisEnsemble(model, a.n, seed=1)
        ens.init(100)
        ens.advance(8, a.nt, collect=(mode != "none"))      # warm-up, allocations
        _lib.default_context().synchronize()
        ta = time.perf_counter()
        buf = None if mode == "none" else _lib.pinned_empty((a.gens, a.n, D))
        if mode == "pageable":
            buf.fill(0.0)       # fault the pages in outside the timed region
        t0 = time.perf_counter()
        tr = ens.advance(a.gens, a.nt, out=buf)
        el = time.perf_counter() - t0
        out.setdefault("alloc_s", {})[mode] = t0 - ta
        gb = a.gens * a.n * D * 8 / 1e9
        out[mode] = {"s": el, "us_per_generation": el / a.gens * 1e6,
                     "host_GBps": None if mode == "none" else gb / el}
        if tr is not None:
            x, _, _, _ = ens.state()
            # the last generation of the trace is push_p of the final state
            assert np.array_equal(tr[-1], x), mode
            if ref is None:
                ref = tr.copy()
            else:
                assert np.array_equal(ref, tr), "pinned and pageable traces differ"
        ens.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
