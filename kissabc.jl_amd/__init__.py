"""kissabc.jl_amd -- MI355X (gfx950) walker-update path of KissABC behind the
reference's own user surface (see include/kabc.h for the C ABI it sits on).

Import name: `kissabc_jl_amd` (the directory name has a dot; the repo-root
module kissabc_jl_amd.py registers this package under that name).
"""
from . import costs
from ._cdefs import KABC_MAX_DIM
from ._lib import Context, KabcError, LIB_PATH, default_context
from .api import (ABCDE, AIS, AisEnsemble, ApproxKernelizedPosterior, ApproxPosterior, CommonLogDensity,
                  MCMCThreads, compile_model, pfilter, set_specialize,
                  Particles, sample, smc)
from . import comm
from .comm import Comm, EnsembleGroup
from .costs import DeviceCost
from .distributions import (Ar1Normal, Beta, Dirichlet, DiscreteUniform, Exponential, Factored, Gamma, InitFromSnippet,
                            LogNormal, MultivariateNormal, MvNormal, NegativeBinomial, Normal, Product,
                            Laplace, Poisson, Truncated, TruncatedGamma, TruncatedNormal, Uniform, UserInit,
                            UserMvPrior, UserPrior, truncated)

__all__ = [
    "ABCDE", "AIS", "AisEnsemble", "ApproxKernelizedPosterior", "ApproxPosterior", "CommonLogDensity",
    "MCMCThreads", "pfilter",
    "Particles", "sample", "smc", "DeviceCost", "costs", "Factored", "Uniform", "Normal",
    "Truncated", "truncated", "TruncatedNormal", "Beta", "DiscreteUniform", "NegativeBinomial",
    "Exponential", "Gamma", "LogNormal", "Product", "MvNormal", "MultivariateNormal", "Context", "KabcError", "default_context", "LIB_PATH",
    "KABC_MAX_DIM", "comm", "Comm", "EnsembleGroup", "UserInit", "InitFromSnippet",
    "UserPrior", "UserMvPrior", "Dirichlet", "Ar1Normal", "Poisson", "Laplace", "TruncatedGamma", "compile_model", "set_specialize",
]
