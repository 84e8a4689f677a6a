"""Generates tests/golden/mvpriors_logpdf.json: scipy.stats reference values for the joint prior families
shipped as snippets (kissabc_jl_amd.distributions.Dirichlet, Ar1Normal).  The reference leaves these to
Distributions.jl (un-vendored: SURVEY 8c), so the pins are scipy's densities, as for the univariate
families (gen_priors_golden.py).  Run here (scipy is in the build image); the JSON is what is committed."""
import json
import os

import numpy as np
from scipy import stats

rng = np.random.default_rng(20261004)
cases = []
for alpha in ([2.0, 3.0, 4.0], [0.7, 1.0, 5.5, 2.25], list(np.linspace(0.5, 3.0, 20))):
    a = np.array(alpha)
    x = rng.dirichlet(a, size=6)
    cases.append({"family": "Dirichlet", "params": {"alpha": alpha}, "x": x.tolist(),
                  "logpdf": [float(stats.dirichlet.logpdf(r, a)) for r in x]})
for D, mu, sg, rho in ((3, 0.5, 1.5, 0.6), (8, -1.0, 0.3, -0.4), (20, 0.0, 1.0, 0.9)):
    idx = np.arange(D)
    cov = sg * sg / (1 - rho * rho) * rho ** np.abs(idx[:, None] - idx[None, :])
    x = rng.multivariate_normal(np.full(D, mu), cov, size=6)
    cases.append({"family": "Ar1Normal", "params": {"D": D, "mu": mu, "sigma": sg, "rho": rho}, "x": x.tolist(),
                  "logpdf": [float(stats.multivariate_normal.logpdf(r, np.full(D, mu), cov)) for r in x]})
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mvpriors_logpdf.json"), "w") as f:
    json.dump({"generator": "gen_mvpriors_golden.py", "scipy": __import__("scipy").__version__, "cases": cases}, f, indent=1)
print(len(cases), "cases")
