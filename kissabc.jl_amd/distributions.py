"""Prior surface of the path: the univariate families the reference's tests,
README and examples use with `Factored` (src/priors.jl:10-49), as plain
descriptors that lower to kabc_prior_t.  Names follow Distributions.jl.

pdf / logpdf / rand / push_p run on the device through the C ABI
(kabc_factored_*): there is no host implementation in the product.
"""
import ctypes as C
import math

import numpy as np

from . import _cdefs as cd
from . import _lib


class UnivariateDistribution:
    kind = 0
    discrete = False

    def params(self):
        raise NotImplementedError

    def to_c(self):
        p = list(self.params()) + [0.0] * 4
        return cd.Prior(self.kind, 0, (C.c_double * 4)(*[float(v) for v in p[:4]]))

    def __len__(self):
        return 1

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(repr(float(v)) for v in self.params())})"


class Uniform(UnivariateDistribution):
    kind = cd.PRIOR_UNIFORM

    def __init__(self, a=0.0, b=1.0):
        self.a, self.b = float(a), float(b)

    def params(self):
        return (self.a, self.b)


class Normal(UnivariateDistribution):
    kind = cd.PRIOR_NORMAL

    def __init__(self, mu=0.0, sigma=1.0):
        self.mu, self.sigma = float(mu), float(sigma)

    def params(self):
        return (self.mu, self.sigma)


class TruncatedNormal(UnivariateDistribution):
    kind = cd.PRIOR_TRUNCNORMAL

    def __init__(self, mu, sigma, lower, upper):
        self.mu, self.sigma, self.lower, self.upper = map(float, (mu, sigma, lower, upper))

    def params(self):
        return (self.mu, self.sigma, self.lower, self.upper)


def Truncated(d, lower, upper):
    """Truncated(Normal(mu, sigma), lower, upper) as in README.md:40."""
    if not isinstance(d, Normal):
        raise TypeError("only Truncated(Normal(...), lo, hi) is on the device path")
    return TruncatedNormal(d.mu, d.sigma, lower, upper)


truncated = Truncated


class Beta(UnivariateDistribution):
    kind = cd.PRIOR_BETA

    def __init__(self, alpha=1.0, beta=1.0):
        self.alpha, self.beta = float(alpha), float(beta)

    def params(self):
        return (self.alpha, self.beta)


class DiscreteUniform(UnivariateDistribution):
    kind = cd.PRIOR_DISCRETE_UNIFORM
    discrete = True

    def __init__(self, a=0, b=1):
        self.a, self.b = float(a), float(b)

    def params(self):
        return (self.a, self.b)


class NegativeBinomial(UnivariateDistribution):
    kind = cd.PRIOR_NEGBINOMIAL
    discrete = True

    def __init__(self, r=1.0, p=0.5):
        self.r, self.p = float(r), float(p)

    def params(self):
        return (self.r, self.p)


class Exponential(UnivariateDistribution):
    kind = cd.PRIOR_EXPONENTIAL

    def __init__(self, theta=1.0):
        self.theta = float(theta)

    def params(self):
        return (self.theta,)


class Gamma(UnivariateDistribution):
    kind = cd.PRIOR_GAMMA

    def __init__(self, alpha=1.0, theta=1.0):
        self.alpha, self.theta = float(alpha), float(theta)

    def params(self):
        return (self.alpha, self.theta)


class LogNormal(UnivariateDistribution):
    kind = cd.PRIOR_LOGNORMAL

    def __init__(self, mu=0.0, sigma=1.0):
        self.mu, self.sigma = float(mu), float(sigma)

    def params(self):
        return (self.mu, self.sigma)


class UserInit(UnivariateDistribution):
    """One coordinate of a CommonLogDensity's own `sample_init` (src/types.jl:105-113: any
    `rng -> sample` in the reference): drawn on the device by the log-density's C snippet
    (`#define KABC_USER_SAMPLE_INIT 1` + `kabc_user_sample_init`, include/kabc_costs.h).  It has
    no density of its own.  `InitFromSnippet(n)` builds the n-coordinate product."""
    kind = cd.PRIOR_USER_INIT

    def params(self):
        return ()


def InitFromSnippet(nparameters):
    return Factored(*[UserInit() for _ in range(int(nparameters))])


class Factored:
    """Factored(d1, d2, ...) -- src/priors.jl:10-13: a product of univariate
    distributions with mixed continuous / discrete support."""

    def __init__(self, *components):
        if not components:
            raise ValueError("Factored needs at least one component")
        for c in components:
            if not isinstance(c, UnivariateDistribution):
                raise TypeError("Factored components must be univariate distributions")
        if len(components) > cd.KABC_MAX_DIM_DYN:
            raise ValueError(f"the device path supports length(prior) <= {cd.KABC_MAX_DIM_DYN}")
        self.p = tuple(components)

    def __len__(self):  # length(p::Factored) = N, src/priors.jl:49
        return len(self.p)

    def __repr__(self):
        return "Factored(" + ", ".join(map(repr, self.p)) + ")"

    def to_c(self):
        arr = (cd.Prior * len(self.p))()
        for i, c in enumerate(self.p):
            arr[i] = c.to_c()
        return arr

    @property
    def discrete_mask(self):
        return np.array([c.discrete for c in self.p], dtype=bool)

    # -- device-evaluated utilities ------------------------------------------
    def _rows(self, x):
        a = np.ascontiguousarray(np.asarray(x, dtype=np.float64))
        single = a.ndim == 1
        a = a.reshape(1, -1) if single else a
        if a.shape[1] != len(self):
            raise ValueError(f"expected {len(self)} coordinates, got {a.shape[1]}")
        return a, single

    def logpdf(self, x, ctx=None):
        """logpdf(d::Factored, x), src/priors.jl:30-36"""
        a, single = self._rows(x)
        out = np.empty(a.shape[0])
        ctx = ctx or _lib.default_context()
        _lib.check(_lib.load().kabc_factored_logpdf(
            ctx.handle, self.to_c(), len(self), a.shape[0],
            a.ctypes.data_as(cd.c_double_p), out.ctypes.data_as(cd.c_double_p)))
        return float(out[0]) if single else out

    def pdf(self, x, ctx=None):
        """pdf(d::Factored, x), src/priors.jl:18-24"""
        return np.exp(self.logpdf(x, ctx))

    def push_p(self, x, ctx=None):
        """push_p(density::Factored, p), src/types.jl:29-32"""
        a, single = self._rows(x)
        out = np.empty_like(a)
        ctx = ctx or _lib.default_context()
        _lib.check(_lib.load().kabc_factored_push_p(
            ctx.handle, self.to_c(), len(self), a.shape[0],
            a.ctypes.data_as(cd.c_double_p), out.ctypes.data_as(cd.c_double_p)))
        return out[0] if single else out

    def rand(self, n=None, seed=0, ctx=None):
        """rand(rng, d::Factored), src/priors.jl:42-43 (n draws; stream `seed`)"""
        m = 1 if n is None else int(n)
        out = np.empty((m, len(self)))
        ctx = ctx or _lib.default_context()
        _lib.check(_lib.load().kabc_factored_rand(
            ctx.handle, self.to_c(), len(self), int(seed), cd.DOM_AIS_INIT, 0, m, 0,
            out.ctypes.data_as(cd.c_double_p)))
        out = self.push_p(out, ctx)
        return out[0] if n is None else out


class Product(Factored):
    """Product([d1, d2, ...]) of Distributions.jl (test/runtests.jl:30): a vector-valued
    product of univariate distributions.  push_p(density::Distribution, p) broadcasts the
    per-component projection (src/types.jl:30), logpdf is the sum of the components' --
    i.e. exactly the Factored kernels; only the walker is a Vector instead of a Tuple."""
    vector_valued = True

    def __init__(self, components):
        super().__init__(*list(components))

    def __repr__(self):
        return "Product([" + ", ".join(map(repr, self.p)) + "])"


class _MvNormalComponent(UnivariateDistribution):
    """Component k of a full-covariance MvNormal: (handle, k) of kabc_mvnormal_register."""
    kind = cd.PRIOR_MVNORMAL

    def __init__(self, owner, k):
        self.owner, self.k = owner, int(k)

    def params(self):
        return (self.owner.handle(), self.k)

    def __repr__(self):
        return f"MvNormal[{self.k}]"


class MvNormal(Product):
    """MultivariateNormal of Distributions.jl as a prior (the reference takes any Distribution:
    src/types.jl:30, :34-35, :52; src/smc.jl:92).

    MvNormal(d, σ) (zero mean, isotropic -- the form test/runtests.jl:186 uses) and MvNormal(μ, σ)
    with σ a scalar or a vector of standard deviations are products of Normals and run on the
    Factored kernels as such.  MvNormal(μ, Σ) with a covariance MATRIX is registered with the
    library (kabc_mvnormal_register: Cholesky factor, its inverse, constants) and travels as D
    components of kind KABC_PRIOR_MVNORMAL (include/kabc_mvnormal.h); a diagonal Σ is lowered to
    Normal(μ_k, sqrt(Σ_kk)) components."""

    def __init__(self, mu_or_dim, sigma=1.0):
        if np.isscalar(mu_or_dim):
            mu = np.zeros(int(mu_or_dim))
        else:
            mu = np.asarray(mu_or_dim, dtype=float).ravel()
        sig = np.asarray(sigma, dtype=float)
        self.mu, self.cov, self._h = mu, None, {}
        if sig.ndim == 2:
            if sig.shape != (mu.size, mu.size):
                raise ValueError(f"MvNormal: Σ must be {mu.size} x {mu.size}")
            if np.count_nonzero(sig - np.diag(np.diagonal(sig))) == 0:
                sig = np.sqrt(np.diagonal(sig))          # diagonal Σ: a product of Normals
            else:
                if mu.size > cd.KABC_MAX_DIM:
                    raise ValueError(f"a full-covariance MvNormal prior supports length(prior) <= {cd.KABC_MAX_DIM}")
                self.cov = np.ascontiguousarray(sig)
                super().__init__([_MvNormalComponent(self, k) for k in range(mu.size)])
                self.handle()                            # validates Σ now (symmetric, positive definite)
                return
        sig = np.broadcast_to(sig, mu.shape)
        super().__init__([Normal(m, s) for m, s in zip(mu, sig)])

    def handle(self):
        """the library's handle of the prepared block (kabc_mvnormal_register), taken once"""
        if "lib" not in self._h:
            h = C.c_int32()
            _lib.check(_lib.load().kabc_mvnormal_register(
                self.mu.ctypes.data_as(cd.c_double_p), self.cov.ctypes.data_as(cd.c_double_p),
                self.mu.size, C.byref(h)))
            self._h["lib"] = int(h.value)
        return self._h["lib"]

    def __repr__(self):
        return f"MvNormal(dim={len(self)}{', full covariance' if self.cov is not None else ''})"


MultivariateNormal = MvNormal


def as_factored(prior):
    """A bare univariate prior (e.g. Normal(1, 0.2), test/runtests.jl:78) is a
    1-component Factored on this path; Product / MvNormal already are Factored."""
    if isinstance(prior, Factored):
        return prior
    if isinstance(prior, UnivariateDistribution):
        return Factored(prior)
    raise TypeError(f"unsupported prior {prior!r}")
