#!/usr/bin/env python3
"""Generates tests/golden/user_priors_logpdf.json with scipy.stats: the run-time compiled prior
families kissabc_jl_amd ships as snippets (Poisson, Laplace, Truncated(Gamma)) -- the reference
takes any UnivariateDistribution in Factored (src/priors.jl:11) and evaluates it through
Distributions.jl, which cannot run here.  Run: python tests/golden/gen_user_priors_golden.py"""
import json
import os

import numpy as np
from scipy import stats

rng = np.random.default_rng(20261003)
out = {"generator": "scipy.stats " + __import__("scipy").__version__, "cases": []}


def add(kind, params, xs, logpdf):
    out["cases"].append({"kind": kind, "params": [float(p) for p in params],
                         "x": [float(v) for v in xs],
                         "logpdf": [float(v) if np.isfinite(v) else ("-inf" if v < 0 else "inf")
                                    for v in logpdf]})


for lam in [0.3, 3.0, 12.5, 140.0]:
    xs = np.concatenate([rng.poisson(lam, 14).astype(float), [0.0, 1.0, -1.0, 2.5, 300.0]])
    add("Poisson", (lam,), xs, stats.poisson(lam).logpmf(xs))
for mu, th in [(0.0, 1.0), (0.5, 2.0), (-3.0, 0.05)]:
    xs = np.concatenate([rng.laplace(mu, 3 * th, 16), [mu]])
    add("Laplace", (mu, th), xs, stats.laplace(mu, th).logpdf(xs))
for a, th, lo, hi in [(2.0, 1.5, 0.5, 6.0), (0.7, 2.0, 0.0, 3.0), (9.0, 0.5, 4.0, 1e9), (1.0, 1.0, 0.25, 0.75)]:
    top = min(hi, lo + 8 * a * th)
    xs = np.concatenate([rng.uniform(lo, top, 14), [lo, min(hi, 1e6), lo - 0.1, hi + 0.1]])
    g = stats.gamma(a, scale=th)
    lp = np.where((xs >= lo) & (xs <= hi), g.logpdf(xs) - np.log(g.cdf(hi) - g.cdf(lo)), -np.inf)
    add("TruncatedGamma", (a, th, lo, hi), xs, lp)

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "user_priors_logpdf.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path, len(out["cases"]), "cases")
