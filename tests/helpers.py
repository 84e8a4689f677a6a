"""Shared helpers for the test-suite (model builders, golden loaders)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_prior_golden(name="priors_logpdf.json"):
    with open(os.path.join(GOLDEN, name)) as f:
        data = json.load(f)
    for c in data["cases"]:
        c["logpdf"] = np.array([(-np.inf if v == "-inf" else np.inf if v == "inf" else v)
                                for v in c["logpdf"]], dtype=float)
        c["x"] = np.array(c["x"], dtype=float)
    return data["cases"]


def make_dist(k, kind, params):
    return {
        "Uniform": k.Uniform, "Normal": k.Normal, "TruncNormal": k.TruncatedNormal,
        "Beta": k.Beta, "DiscreteUniform": k.DiscreteUniform,
        "NegativeBinomial": k.NegativeBinomial, "Exponential": k.Exponential,
        "Gamma": k.Gamma, "LogNormal": k.LogNormal,
        # run-time compiled families (kabc_compile_prior_plugin)
        "Poisson": k.Poisson, "Laplace": k.Laplace, "TruncatedGamma": k.TruncatedGamma,
    }[kind](*params)


def ulp_diff(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    sp = np.abs(np.nextafter(b, np.inf) - b)
    sp = np.where(sp == 0, 5e-324, sp)
    with np.errstate(invalid="ignore"):
        d = np.abs(a - b) / sp
    return np.where(same, 0.0, d)
