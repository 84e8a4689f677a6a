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
    """Truncated(Normal(mu, sigma), lower, upper) as in README.md:40 (a built-in family);
    Truncated(Gamma(alpha, theta), lower, upper) is a run-time compiled family (UserPrior)."""
    if isinstance(d, Normal):
        return TruncatedNormal(d.mu, d.sigma, lower, upper)
    if isinstance(d, Gamma):
        return TruncatedGamma(d.alpha, d.theta, lower, upper)
    raise TypeError("Truncated(Normal(...), lo, hi) and Truncated(Gamma(...), lo, hi) are on the device "
                    "path; any other family: write it as a UserPrior snippet")


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


# ---- families compiled at run time ---------------------------------------------------
USER_PRIOR_SIGNATURE = (
    "KABC_HD double kabc_user_prior_logpdf(double x, const double* p, const double* tab);\n"
    "KABC_HD double kabc_user_prior_rand(const double* p, const kabc_slotwin_t* w);")
_user_kinds = {}


class UserPrior(UnivariateDistribution):
    """A prior family from a C snippet -- the device-path counterpart of "Factored takes any
    UnivariateDistribution" (src/priors.jl:11; logpdf :31-33, rand :43, push_p src/types.jl:30-32).

    `source` defines

        KABC_HD double kabc_user_prior_logpdf(double x, const double* p, const double* tab);
        KABC_HD double kabc_user_prior_rand(const double* p, const kabc_slotwin_t* w);

    x: the coordinate after push_p (rounded when `discrete`); p: `params` (at most four doubles);
    -Inf outside the support; tab: what kabc_log_t / kabc_log1p_t / kabc_lgamma_t (kabc_math.h)
    take.  rand draws from the component's window of the counter stream (kabc_slot(w, j), the
    helpers of include/kabc_sampling_base.h).  Derived constants (a truncation's log-mass, a
    normaliser) go into the snippet text as hexadecimal floating literals (`float.hex`): the host
    computes them once, as Distributions.jl does when the distribution object is built.
    The snippet is compiled by hipRTC at once (kabc_compile_prior_plugin, include/kabc.h); the
    kernels of a prior that contains the family are compiled at first use."""

    def __init__(self, source, params=(), discrete=False, name="user_prior"):
        self.source, self.discrete, self.name = str(source), bool(discrete), name
        self._params = tuple(float(v) for v in params)
        if len(self._params) > 4:
            raise ValueError("a prior component carries at most four parameters (kabc_prior_t.p[4]); "
                             "put derived constants into the snippet as literals")
        key = (self.source, self.discrete)
        kind = _user_kinds.get(key)
        if kind is None:
            out = C.c_int32()
            _lib.check(_lib.load().kabc_compile_prior_plugin(self.source.encode(), int(self.discrete),
                                                             C.byref(out)))
            kind = _user_kinds[key] = int(out.value)
        self.kind = kind

    def params(self):
        return self._params

    def __repr__(self):
        return f"{self.name}({', '.join(repr(v) for v in self._params)})"


def _hx(v):
    """a double as a C literal with exactly its bits"""
    v = float(v)
    if math.isinf(v):
        return "(-KABC_INF)" if v < 0 else "KABC_INF"
    return v.hex()


class Poisson(UserPrior):
    """Poisson(lambda) of Distributions.jl: logpdf = x log(lambda) - lambda - lgamma(x + 1) on the
    non-negative integers; p = (lambda, log lambda)."""
    SOURCE = """
KABC_HD double kabc_user_prior_logpdf(double x, const double* p, const double* tab) {
    if (!(x >= 0.0) || x != kabc_rint(x)) return -KABC_INF;
    return x * p[1] - p[0] - kabc_lgamma_t(x + 1.0, tab);
}
KABC_HD double kabc_user_prior_rand(const double* p, const kabc_slotwin_t* w) {
    return kabc_sample_poisson(w, 0u, p[0]);
}
"""

    def __init__(self, lam=1.0):
        if not lam > 0:
            raise ValueError("Poisson: lambda must be > 0")
        self.lam = float(lam)
        super().__init__(self.SOURCE, (self.lam, math.log(self.lam)), discrete=True, name="Poisson")

    def __repr__(self):
        return f"Poisson({self.lam!r})"


class Laplace(UserPrior):
    """Laplace(mu, theta): logpdf = -|x - mu| / theta - log(2 theta); p = (mu, theta, 1/theta,
    log(2 theta)); rand by inversion of the CDF."""
    SOURCE = """
KABC_HD double kabc_user_prior_logpdf(double x, const double* p, const double* tab) {
    (void)tab;
    return -kabc_div_rc(kabc_fabs(x - p[0]), p[1], p[2]) - p[3];
}
KABC_HD double kabc_user_prior_rand(const double* p, const kabc_slotwin_t* w) {
    const kabc_u128_t b = kabc_slot(w, 0);
    const double u = kabc_u01(kabc_lo64(b));            /* (0, 1] */
    const double e = -p[1] * kabc_log(u);               /* Exponential(theta) */
    return (kabc_hi64(b) & 1ull) ? p[0] + e : p[0] - e;
}
"""

    def __init__(self, mu=0.0, theta=1.0):
        if not theta > 0:
            raise ValueError("Laplace: theta must be > 0")
        self.mu, self.theta = float(mu), float(theta)
        super().__init__(self.SOURCE, (self.mu, self.theta, 1.0 / self.theta, math.log(2.0 * self.theta)),
                         name="Laplace")

    def __repr__(self):
        return f"Laplace({self.mu!r}, {self.theta!r})"


def _gamma_p(a, x):
    """regularised lower incomplete gamma P(a, x) (series / Lentz continued fraction)"""
    if x <= 0.0:
        return 0.0
    if math.isinf(x):
        return 1.0
    lg = math.lgamma(a)
    if x < a + 1.0:
        term = summ = 1.0 / a
        n = a
        for _ in range(10000):
            n += 1.0
            term *= x / n
            summ += term
            if abs(term) < abs(summ) * 1e-17:
                break
        return summ * math.exp(-x + a * math.log(x) - lg)
    tiny = 1e-300
    b = x + 1.0 - a
    c = 1.0 / tiny
    d = 1.0 / b
    h = d
    for i in range(1, 10000):
        an = -i * (i - a)
        b += 2.0
        d = an * d + b
        d = tiny if abs(d) < tiny else d
        c = b + an / c
        c = tiny if abs(c) < tiny else c
        d = 1.0 / d
        delta = d * c
        h *= delta
        if abs(delta - 1.0) < 1e-16:
            break
    return 1.0 - math.exp(-x + a * math.log(x) - lg) * h


class TruncatedGamma(UserPrior):
    """Truncated(Gamma(alpha, theta), lower, upper): the Gamma log-density minus the log-mass of
    [lower, upper] inside it, -Inf outside; p = (alpha, theta, lower, upper), the normaliser
    lgamma(alpha) + alpha log(theta) + logtp is a literal of the snippet.

    rand is a rejection sampler over the component's window of the counter stream, with the
    envelope chosen HERE, once, by acceptance rate (a literal of the snippet):
      * the parent Gamma (Marsaglia-Tsang; for alpha < 1 with the boost U^(1/alpha) -- a FRESH U
        per proposal, so that (U, G) are rejected jointly), accepted when inside [lower, upper]:
        rate ~ 0.95 x the window's mass;
      * a uniform on [lower, upper] against the density's maximum there: rate = mass / ((upper -
        lower) max pdf) -- what serves the narrow windows of little mass.
    A window neither envelope fills with probability 1 - 1e-12 within the slots is refused at
    construction (Distributions.jl switches to quantile inversion there; the incomplete-gamma
    inverse is not part of the arithmetic contract)."""
    TEMPLATE = """
KABC_HD double kabc_user_prior_logpdf(double x, const double* p, const double* tab) {
    if (!(x >= p[2] && x <= p[3]) || !(x >= 0.0)) return -KABC_INF;
    const double t1 = (p[0] == 1.0) ? 0.0 : (p[0] - 1.0) * kabc_log_t(x, tab);
    return t1 - kabc_div_rc(x, p[1], %(rtheta)s) - %(norm)s;
}
KABC_HD double kabc_user_prior_rand(const double* p, const kabc_slotwin_t* w) {
    const double a0 = p[0];
#if %(uniform_envelope)d
    /* uniform proposals on [lo, upper] against the (unnormalised) log-density's maximum there */
    const double lo = %(lo)s, width = p[3] - lo;
    for (uint32_t j = 0; j < KABC_SLOTS_PER_DIM; ++j) {
        const kabc_u128_t b = kabc_slot(w, j);
        const double x = lo + width * kabc_u01(kabc_lo64(b));
        if (!(x >= lo && x <= p[3]) || !(x > 0.0)) continue;
        const double lf = ((a0 == 1.0) ? 0.0 : (a0 - 1.0) * kabc_log(x)) - kabc_div_rc(x, p[1], %(rtheta)s);
        if (kabc_log(kabc_u01(kabc_hi64(b))) < lf - %(logfmax)s) return x;
    }
    return %(xmax)s;
#else
    /* Marsaglia-Tsang proposals of the parent, two per group of blocks (normals, accept uniforms,
     * and for alpha < 1 the boost uniforms), until one lands in [lower, upper] */
    const int small = a0 < 1.0;
    const double a = small ? a0 + 1.0 : a0;
    const double d = a - 1.0 / 3.0, c = 1.0 / kabc_sqrt(9.0 * d);
    const uint32_t step = small ? 3u : 2u;
    for (uint32_t j = 0; j + step <= KABC_SLOTS_PER_DIM; j += step) {
        const kabc_u128_t bn = kabc_slot(w, j), bu = kabc_slot(w, j + 1u);
        double z0, z1;
        kabc_normal_pair(kabc_lo64(bn), kabc_hi64(bn), &z0, &z1);
        const double us[2] = {kabc_u01(kabc_lo64(bu)), kabc_u01(kabc_hi64(bu))};
        const double zs[2] = {z0, z1};
        double boost[2] = {1.0, 1.0};
        if (small) {
            const kabc_u128_t bb = kabc_slot(w, j + 2u);
            boost[0] = kabc_exp(kabc_log(kabc_u01(kabc_lo64(bb))) / a0);
            boost[1] = kabc_exp(kabc_log(kabc_u01(kabc_hi64(bb))) / a0);
        }
        for (int i = 0; i < 2; ++i) {
            double v = 1.0 + c * zs[i];
            if (v <= 0.0) continue;
            v = v * v * v;
            if (kabc_log(us[i]) < 0.5 * zs[i] * zs[i] + d - d * v + d * kabc_log(v)) {
                const double x = d * v * boost[i] * p[1];
                if (x >= p[2] && x <= p[3]) return x;
            }
        }
    }
    return %(xmax)s;  /* (probability < 1e-12 by construction: the point of highest density) */
#endif
}
"""
    SLOTS = 128   # KABC_SLOTS_PER_DIM (include/kabc_sampling_base.h)

    def __init__(self, alpha, theta, lower, upper):
        alpha, theta, lower, upper = map(float, (alpha, theta, lower, upper))
        if not (alpha > 0 and theta > 0 and upper > lower):
            raise ValueError("Truncated(Gamma): alpha, theta > 0 and upper > lower")
        lo = max(lower, 0.0)
        tp = _gamma_p(alpha, upper / theta) - _gamma_p(alpha, lo / theta)
        if not tp > 0:
            raise ValueError("Truncated(Gamma): the interval has no mass")
        self.alpha, self.theta, self.lower, self.upper = alpha, theta, lower, upper
        self.logtp = math.log(tp)
        lognorm0 = math.lgamma(alpha) + alpha * math.log(theta)
        norm = lognorm0 + self.logtp
        # the point of highest density inside the window, and log of the unnormalised density there
        xmax = min(max((alpha - 1.0) * theta, lo), upper) if alpha >= 1.0 else lo
        logf = lambda x: ((alpha - 1.0) * math.log(x) if alpha != 1.0 else 0.0) - x / theta   # noqa: E731
        # acceptance rate per proposal and proposals per window of either envelope
        rate_parent = 0.95 * tp
        n_parent = 2 * (self.SLOTS // (3 if alpha < 1.0 else 2))
        rate_unif, logfmax = 0.0, 0.0
        if math.isfinite(upper) and xmax > 0.0:
            logfmax = logf(xmax)
            rate_unif = tp / ((upper - lo) * math.exp(logfmax - lognorm0))
        fail_parent = (1.0 - min(rate_parent, 1.0)) ** n_parent
        fail_unif = (1.0 - min(rate_unif, 1.0)) ** self.SLOTS if rate_unif > 0 else 1.0
        self.envelope = "uniform" if fail_unif < fail_parent else "parent"
        if min(fail_parent, fail_unif) > 1e-12:
            raise ValueError(
                f"Truncated(Gamma({alpha}, {theta}), {lower}, {upper}): the window holds {tp:.3g} of the mass and "
                "neither rejection envelope of the device sampler fills it reliably (Distributions.jl "
                "inverts the quantile there, which the arithmetic contract does not provide)")
        super().__init__(self.TEMPLATE % {"rtheta": _hx(1.0 / theta), "norm": _hx(norm),
                                          "uniform_envelope": int(self.envelope == "uniform"),
                                          "lo": _hx(lo), "logfmax": _hx(logfmax), "xmax": _hx(xmax)},
                         (alpha, theta, lower, upper), name="TruncatedGamma")

    def __repr__(self):
        return f"Truncated(Gamma({self.alpha!r}, {self.theta!r}), {self.lower!r}, {self.upper!r})"


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


_user_mv_kinds = {}


class _JointComponent(UnivariateDistribution):
    """component k of a joint user prior: the family's kind and the component's (at most three) parameters"""

    def __init__(self, kind, params, source):
        self.kind, self._params, self.joint_source = int(kind), tuple(float(v) for v in params), source

    def params(self):
        return self._params


class UserMvPrior(Factored):
    """A JOINT prior from a C snippet -- the device-path counterpart of handing any multivariate
    `Distribution` to ApproxKernelizedPosterior / ApproxPosterior / smc (the reference calls
    rand(rng, prior) and logpdf(prior, x) on whatever it is given: src/types.jl:30,34-35,52;
    src/smc.jl:92-93).  `source` defines

        KABC_HD double kabc_user_mvprior_logpdf(const double* x, int D, const double* p, int pstride,
                                                const double* tab);
        KABC_HD void   kabc_user_mvprior_rand(double* out, int D, const double* p, int pstride,
                                              const kabc_slotwin_t* w);

    x: the whole vector; component k's parameters: p[k * pstride + 0..2] (`params`: D rows of at most
    three doubles); -Inf outside the support; rand fills out[0..D) from the walker's window
    (kabc_slot(w, j), j < D * KABC_SLOTS_PER_DIM; helpers of include/kabc_sampling_base.h).
    kabc_compile_mvprior_plugin (include/kabc.h) compiles the snippet at once; the kernels of a model
    with such a prior are compiled at first use, like those of a univariate user family."""

    def __init__(self, source, params, name="user_mvprior"):
        self.source, self.name = str(source), name
        rows = [tuple(float(v) for v in np.atleast_1d(r)) for r in params]
        if not rows or len(rows) > cd.KABC_MAX_DIM_DYN:
            raise ValueError(f"a joint prior has 1..{cd.KABC_MAX_DIM_DYN} components")
        if any(len(r) > 3 for r in rows):
            raise ValueError("a component of a joint prior carries at most three parameters (kabc_prior_t.p[0..2])")
        kind = _user_mv_kinds.get(self.source)
        if kind is None:
            out = C.c_int32()
            _lib.check(_lib.load().kabc_compile_mvprior_plugin(self.source.encode(), C.byref(out)))
            kind = _user_mv_kinds[self.source] = int(out.value)
        self.kind = kind
        super().__init__(*[_JointComponent(kind, r, self.source) for r in rows])

    def __repr__(self):
        return f"{self.name}(D={len(self)})"


class Dirichlet(UserMvPrior):
    """Dirichlet(alpha) of Distributions.jl as a prior: logpdf = sum (alpha_k - 1) log x_k - log B(alpha) on
    the probability simplex (all x_k >= 0 and |sum x - 1| <= D * 2^-52 * 4, the tolerance of isprobvec's
    isapprox at this scale), -Inf elsewhere; rand: independent Gamma(alpha_k, 1) draws, normalised.
    p_k = (alpha_k, -, log B(alpha))."""
    SOURCE = """
KABC_HD double kabc_user_mvprior_logpdf(const double* x, int D, const double* p, int pstride, const double* tab) {
    double sx = 0.0, s = 0.0;
    for (int k = 0; k < D; ++k) {
        if (!(x[k] >= 0.0)) return -KABC_INF;
        sx += x[k];
    }
    if (!(kabc_fabs(sx - 1.0) <= (double)D * 0x1p-50)) return -KABC_INF;
    for (int k = 0; k < D; ++k) {
        const double a = p[k * pstride];
        if (a != 1.0) s += (a - 1.0) * kabc_log_t(x[k], tab);
    }
    return s - p[2];
}
KABC_HD void kabc_user_mvprior_rand(double* out, int D, const double* p, int pstride, const kabc_slotwin_t* w) {
    double sm = 0.0;
    for (int k = 0; k < D; ++k) {
        kabc_slotwin_t wk = *w;
        wk.base = w->base + (uint32_t)k * KABC_SLOTS_PER_DIM;
        out[k] = kabc_sample_gamma1(&wk, 0u, p[k * pstride]);
        sm += out[k];
    }
    for (int k = 0; k < D; ++k) out[k] = out[k] / sm;
}
"""

    def __init__(self, alpha):
        a = [float(v) for v in np.atleast_1d(alpha)]
        if len(a) < 2 or min(a) <= 0:
            raise ValueError("Dirichlet(alpha): at least two positive concentrations")
        logB = sum(math.lgamma(v) for v in a) - math.lgamma(sum(a))
        self.alpha = np.array(a)
        super().__init__(self.SOURCE, [(v, 0.0, logB) for v in a], name="Dirichlet")


class Ar1Normal(UserMvPrior):
    """A stationary Gaussian AR(1) process as a joint prior: x_1 ~ N(mu, sigma / sqrt(1 - rho^2)),
    x_k | x_{k-1} ~ N(mu + rho (x_{k-1} - mu), sigma) -- a latent time series; not a product of
    univariate densities.  p_k = (mu, sigma, rho)."""
    SOURCE = """
KABC_HD double kabc_user_mvprior_logpdf(const double* x, int D, const double* p, int pstride, const double* tab) {
    const double mu = p[0], sg = p[1], rho = p[2];
    const double s0 = sg / kabc_sqrt(1.0 - rho * rho);
    double z = (x[0] - mu) / s0;
    double s = -(z * z + KABC_LOG_2PI) / 2.0 - kabc_log_t(s0, tab);
    const double lsg = kabc_log_t(sg, tab);
    for (int k = 1; k < D; ++k) {
        z = (x[k] - (mu + rho * (x[k - 1] - mu))) / sg;
        s += -(z * z + KABC_LOG_2PI) / 2.0 - lsg;
    }
    (void)pstride;
    return s;
}
KABC_HD void kabc_user_mvprior_rand(double* out, int D, const double* p, int pstride, const kabc_slotwin_t* w) {
    const double mu = p[0], sg = p[1], rho = p[2];
    double prev = 0.0;
    for (int k = 0; k < D; ++k) {
        const kabc_u128_t b = kabc_slot(w, (uint32_t)k * KABC_SLOTS_PER_DIM);
        double z0, z1;
        kabc_normal_pair(kabc_lo64(b), kabc_hi64(b), &z0, &z1);
        prev = (k == 0) ? mu + sg / kabc_sqrt(1.0 - rho * rho) * z0 : mu + rho * (prev - mu) + sg * z0;
        out[k] = prev;
    }
    (void)pstride;
}
"""

    def __init__(self, D, mu=0.0, sigma=1.0, rho=0.5):
        if not (sigma > 0 and abs(rho) < 1 and int(D) >= 1):
            raise ValueError("Ar1Normal(D, mu, sigma, rho): sigma > 0, |rho| < 1")
        self.mu, self.sigma, self.rho = float(mu), float(sigma), float(rho)
        super().__init__(self.SOURCE, [(mu, sigma, rho)] * int(D), name="Ar1Normal")


def as_factored(prior):
    """A bare univariate prior (e.g. Normal(1, 0.2), test/runtests.jl:78) is a
    1-component Factored on this path; Product / MvNormal already are Factored."""
    if isinstance(prior, Factored):
        return prior
    if isinstance(prior, UnivariateDistribution):
        return Factored(prior)
    raise TypeError(f"unsupported prior {prior!r}")
