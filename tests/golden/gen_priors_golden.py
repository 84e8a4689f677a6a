#!/usr/bin/env python3
"""Generates tests/golden/priors_logpdf.json with scipy.stats.

The reference evaluates prior densities through Distributions.jl
(src/priors.jl:18-36), which is not in the reference tree and cannot run
here (no Julia).  These vectors pin our restatement of each family against an
independent implementation (scipy 1.15).  Run: python tests/golden/gen_priors_golden.py
"""
import json
import os

import numpy as np
from scipy import stats

rng = np.random.default_rng(20261002)
out = {"generator": "scipy.stats " + __import__("scipy").__version__, "cases": []}


def add(kind, params, xs, logpdf):
    out["cases"].append({"kind": kind, "params": [float(p) for p in params],
                         "x": [float(v) for v in xs],
                         "logpdf": [float(v) if np.isfinite(v) else ("-inf" if v < 0 else "inf")
                                    for v in logpdf]})


# Uniform(a,b)
for a, b in [(0, 1), (-5, 5), (100, 101), (1, 3), (0, 4)]:
    xs = np.concatenate([rng.uniform(a - 1, b + 1, 12), [a, b, a - 1e-9, b + 1e-9]])
    add("Uniform", (a, b), xs, stats.uniform(a, b - a).logpdf(xs))
# Normal(mu,sigma)
for mu, s in [(0, 1), (0, 5), (1, 0.2), (1, 0.5), (-3.5, 12.0)]:
    xs = rng.normal(mu, 3 * s, 16)
    add("Normal", (mu, s), xs, stats.norm(mu, s).logpdf(xs))
# Truncated(Normal(mu,sigma), lo, hi)
for mu, s, lo, hi in [(0, 0.1, 0, 100), (0, 0.05, 0, 100), (1, 2, -1, 4), (0, 1, 2, 5)]:
    xs = np.concatenate([rng.uniform(lo - 0.5, min(hi, lo + 6 * s) + 0.5, 14), [lo, hi]])
    add("TruncNormal", (mu, s, lo, hi), xs,
        stats.truncnorm((lo - mu) / s, (hi - mu) / s, loc=mu, scale=s).logpdf(xs))
# Beta(alpha,beta)
for a, b in [(15, 2), (2, 2), (0.5, 0.5), (1, 3), (4.5, 1)]:
    xs = np.concatenate([rng.uniform(0, 1, 14), [-0.1, 1.1, 1e-12, 1 - 1e-12]])
    add("Beta", (a, b), xs, stats.beta(a, b).logpdf(xs))
# DiscreteUniform(a,b)
for a, b in [(1, 2), (1, 10), (0, 0), (-3, 4)]:
    xs = np.concatenate([np.arange(a - 2, b + 3, dtype=float), [a + 0.5]])
    lp = stats.randint(a, b + 1).logpmf(xs)
    add("DiscreteUniform", (a, b), xs, lp)
# NegativeBinomial(r,p)  (socks prior: test/runtests.jl:46-50)
prior_mu, prior_sd = 30.0, 15.0
size = -prior_mu ** 2 / (prior_mu - prior_sd ** 2)
for r, p in [(size, size / (prior_mu + size)), (1, 0.5), (10, 0.9), (0.3, 0.01)]:
    xs = np.concatenate([rng.integers(0, 200, 14).astype(float), [0.0, 1.0, -1.0, 2.5]])
    add("NegativeBinomial", (r, p), xs, stats.nbinom(r, p).logpmf(xs))
# Exponential(theta), Gamma(alpha, theta), LogNormal(mu, sigma)
for th in [1.0, 0.2, 7.5]:
    xs = np.concatenate([rng.exponential(th, 12), [0.0, -1.0]])
    add("Exponential", (th,), xs, stats.expon(scale=th).logpdf(xs))
for a, th in [(1.0, 1.0), (2.5, 0.7), (0.4, 3.0), (30.0, 0.1)]:
    xs = np.concatenate([rng.gamma(a, th, 12), [-1.0]])
    add("Gamma", (a, th), xs, stats.gamma(a, scale=th).logpdf(xs))
for mu, s in [(0, 1), (1.5, 0.3)]:
    xs = np.concatenate([rng.lognormal(mu, s, 12), [0.0, -2.0]])
    add("LogNormal", (mu, s), xs, stats.lognorm(s, scale=np.exp(mu)).logpdf(xs))

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "priors_logpdf.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote", path, len(out["cases"]), "cases")
