# KissABCHip.jl -- thin `ccall` layer that puts the MI355X walker-update path
# (libkabc_hip.so, C ABI in include/kabc.h) behind KissABC.jl's OWN surface.  Loading it
# adds methods, it introduces no sampler type of its own: a model whose `cost` is a
# DeviceCost runs on the GPU through the reference's calls, unchanged --
#
#     model = ApproxKernelizedPosterior(prior, Rosenbrock(), 1.0)       # KissABC's type
#     sample(model, AIS(65536), 10^6; ntransitions = 100, discard_initial = 10^5)
#     sample(model, AIS(12), MCMCThreads(), 100, 50)                    # chains = one batch handle
#     smc(prior, cost; nparticles = 32768)      ABCDE(prior, cost, ϵ)      pfilter(prior, cost, N)
#     sample(CommonLogDensity(2, InitFrom(Factored(Normal(), Normal())), lπ::DeviceCost), AIS(50), 1000)
#     sample_sharded(model, AIS(524288), [0,1,2,3,4,5,6,7], 10^6)       # one process, 8 GPUs, RCCL
#
# -- and a model with an ordinary Julia closure keeps running through KissABC's CPU path.
# Dispatch: `AbstractMCMC.step(rng, model::DeviceModel, spl::AIS; ...)` is more specific than
# the reference's `step(rng, model::AbstractDensity, spl::AIS; ...)` (src/KissABC.jl:35-41,66-72).
#
# It contains no numerics: it lowers `Factored` / `Product` / `MvNormal` (diagonal or full) /
# univariate Distributions to `kabc_prior_t`, a DeviceCost to `kabc_cost_t`, calls the library
# and wraps the result in `Particles` exactly as src/KissABC.jl:82-104 and src/smc.jl:200-205,
# :334-340, :425-430 do.  It mirrors kissabc.jl_amd/api.py call for call.
#
# STATUS: EXPERIMENTAL -- Julia is not available in the build image of this repository, so
# this file has been checked by eye only; the identical call sequences are exercised through
# the Python ctypes mirror (api.py, comm.py) and from plain C (examples/abi_demo.c) by tests/.
module KissABCHip

using KissABC, Random, LinearAlgebra, Statistics, Printf
import AbstractMCMC
import AbstractMCMC: MCMCThreads
import KissABC: Factored, ApproxKernelizedPosterior, ApproxPosterior, CommonLogDensity, AIS, Particles
using Distributions

const libkabc = get(ENV, "KABC_LIB", joinpath(@__DIR__, "..", "lib", "libkabc_hip.so"))

# ---- struct mirrors of include/kabc.h ---------------------------------------
struct KabcPrior
    kind::Int32
    reserved::Int32
    p::NTuple{4,Float64}
end
struct KabcCost
    id::Int32
    nparams::Int32
    params::Ptr{Float64}
    ndata::Int64
    data::Ptr{Float64}
end
struct KabcModel
    prior::Ptr{KabcPrior}
    D::Int32
    posterior::Int32
    eps::Float64
    cost::KabcCost
end
mutable struct KabcStats
    proposals::UInt64
    cost_evals::UInt64
    accepted::UInt64
end
mutable struct KabcSmcOpts
    nparticles::Int64
    alpha::Float64
    mcmc_retrys::Int32
    verbose::Int32
    mcmc_tol::Float64
    epstol::Float64
    r_epstol::Float64
    min_r_ess::Float64
    max_stretch::Float64
    seed::UInt64
    max_iterations::Int64
end
struct KabcSmcIter       # kabc_smc_iter_t (one entry of the optional iteration log)
    eps::Float64
    ess::Int64
    accepted::Int64
    resampled::Int32
    flag::Int32
    mcmc_passes::Int32
    reserved::Int32
end
mutable struct KabcSmcResult
    theta::Ptr{Float64}
    cost::Ptr{Float64}
    alive::Ptr{UInt8}
    eps::Float64
    iterations::Int64
    n_alive::Int64
    cost_evals::UInt64
    proposals::UInt64
    iter_log::Ptr{Cvoid}
    iter_log_cap::Int64
    kernel_ms_mcmc::Float64
    mcmc_launches::Int64
end
mutable struct KabcAbcdeOpts
    nparticles::Int64
    generations::Int64
    eps_target::Float64
    alpha::Float64
    proposal_width::Float64
    earlystop::Int32
    verbose::Int32
    seed::UInt64
end
mutable struct KabcAbcdeResult
    theta::Ptr{Float64}
    cost::Ptr{Float64}
    reached_eps::Int32
    reserved::Int32
    generations_run::Int64
    nsims::UInt64
end
mutable struct KabcPfilterOpts
    nparticles::Int64
    q::Float64
    eff_tol::Float64
    epstol::Float64
    proposal_width::Float64
    max_iters::Int64      # Inf -> -1
    verbose::Int32
    reserved::Int32
    seed::UInt64
end
mutable struct KabcPfilterResult
    theta::Ptr{Float64}
    cost::Ptr{Float64}
    eps::Float64
    eff::Float64
    iterations::Int64
    nreps::UInt64
    cost_evals::UInt64
end

check(st) = st == 0 ? nothing :
    error(unsafe_string(ccall((:kabc_last_error, libkabc), Cstring, ())))   # reference's text

# ---- DeviceCost: the `cost` argument on the device path ----------------------
# A DeviceCost is also callable on the CPU (`cpu` holds the same formula as a Julia closure,
# written as the reference's tests write it), so ONE model object runs through KissABC's own
# AIS/smc and through the HIP path.  ids/formulas: include/kabc_costs.h.
struct DeviceCost{F}
    id::Int32
    params::Vector{Float64}
    data::Vector{Float64}
    cpu::F
end
(c::DeviceCost)(x) = c.cpu(x)
devcost(id, params, data, f) = DeviceCost(Int32(id), collect(Float64, params), collect(Float64, data), f)

GaussDist(c) = devcost(1, c, Float64[], x -> sqrt(sum(abs2, collect(x) .- c)))
Rosenbrock() = devcost(2, Float64[], Float64[],
    x -> sqrt(sum(100 * (x[k+1] - x[k]^2)^2 + (1 - x[k])^2 for k in 1:length(x)-1)))
# θ = (m, s, z_1..z_G): ȳ_g = m + s z_g + randn/√8, cost = RMS(ȳ − ȳ_obs)   (SURVEY 8d, C4)
HierGaussSim(ybar) = devcost(3, Float64[], ybar,
    x -> sqrt(sum(abs2, (x[1] + x[2] * x[2+g] + randn() / sqrt(8) - ybar[g]) for g in 1:length(ybar)) / length(ybar)))
# README.md:43-49: n draws N(μ, σ); hypot(mean − mean(tdata), 50 (std − std(tdata)))
NormalMeanStdSim(n, m_obs, s_obs) = devcost(4, [n, m_obs, s_obs], Float64[],
    ((μ, σ),) -> (y = μ .+ σ .* randn(Int(n)); hypot(mean(y) - m_obs, 50 * (std(y) - s_obs))))
DiracSq(t = 1.5) = devcost(5, [t], Float64[], x -> abs(x[1]^2 + 1 - t))            # runtests.jl:79-80
AbsDiff(t) = devcost(6, [t], Float64[], x -> abs(x[1] - t))                        # runtests.jl:178
NormShell(t) = devcost(7, [t], Float64[], x -> abs(sqrt(sum(abs2, x)) - t))        # runtests.jl:186
NoisyQuadDU(t = 5.5) = devcost(8, [t], Float64[],                                  # runtests.jl:108-109
    ((n, du),) -> abs((n * n + du) * (n + randn() * 0.01) - t))
Mixture(t = 0.0) = devcost(9, [t], Float64[],                                      # runtests.jl:145-146
    μ -> abs(μ[1] + rand((randn() * 0.1, randn())) - t))
NoisyBanana(p = 0.0) = devcost(10, [p], Float64[],                                 # runtests.jl:242,248
    ((x, y),) -> rand() < p ? Inf : 50 * (x + randn() * 0.01 - y^2)^2 + (y - 1 + randn() * 0.01)^2)
WienerRms(tdata) = devcost(11, Float64[], tdata,                                   # runtests.jl:116-126
    ((μ, σ),) -> (t = 0:length(tdata)-1;
                  sum(abs, sqrt.(μ^2 .* t .^ 2 .+ σ^2 .* t) .* (0.95 + 0.1 * rand()) .- tdata) / length(tdata)))

"""
    UserCost(csrc, dims; params, data, cpu, posteriors)
A DeviceCost from a C snippet defining `kabc_user_cost` (include/kabc_costs.h,
KABC_COST_USER), compiled IN PROCESS by hipRTC through `kabc_compile_cost_plugin`
(include/kabc.h): the snippet is checked here, each kernel family is compiled when a model
first uses it.  `cpu` is the Julia closure with the same formula (used when the model runs
through KissABC's own AIS/smc).  With posterior kind CommonLogDensity the snippet returns the
log-density.  `posteriors`: bit mask of the posterior kinds to build AIS kernels for
(1 kernelized, 2 threshold, 4 common; 0 = all).
"""
function UserCost(csrc::String, dims; params = Float64[], data = Float64[], cpu = x -> NaN,
                  posteriors::Integer = 0)
    d = Int32[Int32(x) for x in dims]
    id = Ref{Int32}(0)
    check(ccall((:kabc_compile_cost_plugin, libkabc), Cint,
                (Cstring, Ptr{Int32}, Int32, Int32, Ref{Int32}), csrc, d, length(d), posteriors, id))
    devcost(id[], params, data, cpu)
end

# the struct mirrors above against the library's own sizeof (kabc_abi_sizeof, include/kabc.h):
# a layout drift between this file and kabc.h fails at load time, not as memory corruption
function __init__()
    mirrors = (KabcPrior, KabcCost, KabcModel, KabcStats, KabcSmcOpts, KabcSmcIter, KabcSmcResult,
               KabcAbcdeOpts, KabcAbcdeResult, KabcPfilterOpts, KabcPfilterResult)
    for (i, T) in enumerate(mirrors)
        want = ccall((:kabc_abi_sizeof, libkabc), Int32, (Int32,), Int32(i - 1))
        want == sizeof(T) || error("KissABCHip: sizeof($T) = $(sizeof(T)) but libkabc_hip has $want " *
                                   "(include/kabc.h changed; update the struct mirrors)")
        for f in 1:fieldcount(T)                 # field by field: kabc_abi_offsetof
            off = ccall((:kabc_abi_offsetof, libkabc), Int32, (Int32, Int32), Int32(i - 1), Int32(f - 1))
            off == fieldoffset(T, f) || error("KissABCHip: $T.$(fieldname(T, f)) sits at $(fieldoffset(T, f)), " *
                                              "libkabc_hip has it at $off")
        end
        ccall((:kabc_abi_offsetof, libkabc), Int32, (Int32, Int32), Int32(i - 1), Int32(fieldcount(T))) == -1 ||
            error("KissABCHip: $T has fewer fields than the library's struct")
    end
end

# ---- Factored / Distributions -> kabc_prior_t --------------------------------
lower(d::Uniform) = KabcPrior(1, 0, (d.a, d.b, 0.0, 0.0))
lower(d::Normal) = KabcPrior(2, 0, (d.μ, d.σ, 0.0, 0.0))
lower(d::Truncated{<:Normal}) = KabcPrior(3, 0, (d.untruncated.μ, d.untruncated.σ, d.lower, d.upper))
lower(d::Beta) = KabcPrior(4, 0, (d.α, d.β, 0.0, 0.0))
lower(d::DiscreteUniform) = KabcPrior(5, 0, (Float64(d.a), Float64(d.b), 0.0, 0.0))
lower(d::NegativeBinomial) = KabcPrior(6, 0, (Float64(d.r), d.p, 0.0, 0.0))
lower(d::Exponential) = KabcPrior(7, 0, (d.θ, 0.0, 0.0, 0.0))
lower(d::Gamma) = KabcPrior(8, 0, (d.α, d.θ, 0.0, 0.0))
lower(d::LogNormal) = KabcPrior(9, 0, (d.μ, d.σ, 0.0, 0.0))
# ---- any other UnivariateDistribution: a prior family compiled at run time -----------------
# Factored takes ANY UnivariateDistribution (src/priors.jl:11).  A family outside the built-in
# kinds is a C snippet (kabc_user_prior_logpdf / kabc_user_prior_rand, include/kabc.h "user prior
# families") registered with kabc_compile_prior_plugin; derived constants that are expensive to
# form (a truncation's log-mass) go into the snippet text as hexadecimal literals, as
# Distributions.jl keeps them in the distribution object.  The snippets below are the ones
# kissabc.jl_amd/distributions.py ships (same text => same kind, same kernels).
const user_prior_kinds = Dict{Tuple{String,Bool},Int32}()
function user_prior_kind(src::String, discrete::Bool)
    get!(user_prior_kinds, (src, discrete)) do
        k = Ref{Int32}(0)
        check(ccall((:kabc_compile_prior_plugin, libkabc), Cint, (Cstring, Int32, Ref{Int32}), src, Int32(discrete), k))
        k[]
    end
end
"UserPrior(csrc, params; discrete): a KabcPrior of a run-time compiled family (at most four parameters)"
UserPrior(src::String, params; discrete::Bool = false) =
    KabcPrior(user_prior_kind(src, discrete), 0, ntuple(i -> i <= length(params) ? Float64(params[i]) : 0.0, 4))
hexlit(v::Float64) = isinf(v) ? (v < 0 ? "(-KABC_INF)" : "KABC_INF") : @sprintf("%a", v)   # the bits, as a C literal

const POISSON_SRC = """
KABC_HD double kabc_user_prior_logpdf(double x, const double* p, const double* tab) {
    if (!(x >= 0.0) || x != kabc_rint(x)) return -KABC_INF;
    return x * p[1] - p[0] - kabc_lgamma_t(x + 1.0, tab);
}
KABC_HD double kabc_user_prior_rand(const double* p, const kabc_slotwin_t* w) {
    return kabc_sample_poisson(w, 0u, p[0]);
}
"""
lower(d::Poisson) = UserPrior(POISSON_SRC, (d.λ, log(d.λ)); discrete = true)
const LAPLACE_SRC = """
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
lower(d::Laplace) = UserPrior(LAPLACE_SRC, (d.μ, d.θ, 1 / d.θ, log(2 * d.θ)))
# Truncated(Gamma(α, θ), lo, hi): the normaliser lgamma(α) + α log θ + logtp is a literal of the snippet;
# so is the rejection envelope of rand, chosen here by acceptance rate exactly as
# kissabc.jl_amd/distributions.py (TruncatedGamma) chooses it: the parent Gamma (a fresh boost
# uniform per proposal when α < 1) or a uniform on the window against the density's maximum there
function lower(d::Truncated{<:Gamma})
    g = d.untruncated
    lognorm0 = Distributions.loggamma(g.α) + g.α * log(g.θ)
    norm = lognorm0 + d.logtp
    lo = max(Float64(d.lower), 0.0)
    hi = Float64(d.upper)
    tp = exp(d.logtp)
    xmax = g.α >= 1 ? min(max((g.α - 1) * g.θ, lo), hi) : lo
    logf(x) = (g.α != 1 ? (g.α - 1) * log(x) : 0.0) - x / g.θ
    slots = 128                                   # KABC_SLOTS_PER_DIM
    n_parent = 2 * (slots ÷ (g.α < 1 ? 3 : 2))
    rate_unif, logfmax = 0.0, 0.0
    if isfinite(hi) && xmax > 0
        logfmax = logf(xmax)
        rate_unif = tp / ((hi - lo) * exp(logfmax - lognorm0))
    end
    fail_parent = (1 - min(0.95 * tp, 1.0))^n_parent
    fail_unif = rate_unif > 0 ? (1 - min(rate_unif, 1.0))^slots : 1.0
    uniform_envelope = fail_unif < fail_parent
    min(fail_parent, fail_unif) > 1e-12 &&
        error("Truncated(Gamma): neither rejection envelope of the device sampler fills this window reliably")
    src = """
KABC_HD double kabc_user_prior_logpdf(double x, const double* p, const double* tab) {
    if (!(x >= p[2] && x <= p[3]) || !(x >= 0.0)) return -KABC_INF;
    const double t1 = (p[0] == 1.0) ? 0.0 : (p[0] - 1.0) * kabc_log_t(x, tab);
    return t1 - kabc_div_rc(x, p[1], $(hexlit(1 / g.θ))) - $(hexlit(norm));
}
KABC_HD double kabc_user_prior_rand(const double* p, const kabc_slotwin_t* w) {
    const double a0 = p[0];
#if $(Int(uniform_envelope))
    /* uniform proposals on [lo, upper] against the (unnormalised) log-density's maximum there */
    const double lo = $(hexlit(lo)), width = p[3] - lo;
    for (uint32_t j = 0; j < KABC_SLOTS_PER_DIM; ++j) {
        const kabc_u128_t b = kabc_slot(w, j);
        const double x = lo + width * kabc_u01(kabc_lo64(b));
        if (!(x >= lo && x <= p[3]) || !(x > 0.0)) continue;
        const double lf = ((a0 == 1.0) ? 0.0 : (a0 - 1.0) * kabc_log(x)) - kabc_div_rc(x, p[1], $(hexlit(1 / g.θ)));
        if (kabc_log(kabc_u01(kabc_hi64(b))) < lf - $(hexlit(logfmax))) return x;
    }
    return $(hexlit(xmax));
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
    return $(hexlit(xmax));  /* (probability < 1e-12 by construction: the point of highest density) */
#endif
}
"""
    UserPrior(src, (g.α, g.θ, Float64(d.lower), Float64(d.upper)))
end

lower_prior(d::UnivariateDistribution) = KabcPrior[lower(d)]
lower_prior(d::Factored) = KabcPrior[lower(c) for c in d.p]
# vector-valued walkers (test/runtests.jl:30,186): products of univariate components run on the
# Factored kernels; only the shape of the emitted sample differs (Vector instead of Tuple)
lower_prior(d::Product) = KabcPrior[lower(c) for c in d.v]
# a diagonal Σ is a product of Normals; a full Σ is registered with the library (Cholesky factor,
# its inverse and the constants are kept there: include/kabc_mvnormal.h) and travels as D
# components of kind KABC_PRIOR_MVNORMAL carrying (handle, index)
const mvnormal_handles = IdDict{Any,Int32}()
function lower_prior(d::AbstractMvNormal)
    Σ = Matrix{Float64}(cov(d))
    μ = Vector{Float64}(mean(d))
    isdiag(Σ) && return KabcPrior[KabcPrior(2, 0, (m, sqrt(v), 0.0, 0.0)) for (m, v) in zip(μ, diag(Σ))]
    h = get!(mvnormal_handles, d) do
        r = Ref{Int32}(0)
        # row-major D x D: Σ is symmetric, so Julia's column-major storage is the same matrix
        check(ccall((:kabc_mvnormal_register, libkabc), Cint, (Ptr{Float64}, Ptr{Float64}, Int32, Ref{Int32}),
                    μ, Σ, Int32(length(μ)), r))
        r[]
    end
    KabcPrior[KabcPrior(11, 0, (Float64(h), Float64(k - 1), 0.0, 0.0)) for k in 1:length(μ)]
end
# ---- any other multivariate Distribution: a JOINT prior compiled at run time -----------------
# The reference hands whatever `prior` is to rand / logpdf (src/types.jl:30,34-35,52; src/smc.jl:92-93).  A
# joint density that is not a product is a C snippet (kabc_user_mvprior_logpdf / kabc_user_mvprior_rand over
# the whole vector, include/kabc.h "joint user priors") registered with kabc_compile_mvprior_plugin; all D
# components of the lowered prior carry its kind, component k its (at most three) parameters.
const user_mvprior_kinds = Dict{String,Int32}()
function user_mvprior_kind(src::String)
    get!(user_mvprior_kinds, src) do
        k = Ref{Int32}(0)
        check(ccall((:kabc_compile_mvprior_plugin, libkabc), Cint, (Cstring, Ref{Int32}), src, k))
        k[]
    end
end
"UserMvPrior(csrc, rows): KabcPrior[] of a run-time compiled joint prior (one row of at most three parameters per component)"
UserMvPrior(src::String, rows) =
    KabcPrior[KabcPrior(user_mvprior_kind(src), 0, ntuple(i -> i <= length(r) ? Float64(r[i]) : 0.0, 4)) for r in rows]
# (the snippet kissabc.jl_amd/distributions.py ships: same text => same kind, same kernels)
const DIRICHLET_SRC = """
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
function lower_prior(d::Dirichlet)
    UserMvPrior(DIRICHLET_SRC, [(a, 0.0, d.lmnB) for a in d.alpha])   # lmnB = log B(alpha), kept by Distributions.jl
end
vector_valued(d) = d isa MultivariateDistribution && !(d isa Factored)

"InitFrom(d): a `sample_init` for CommonLogDensity that is callable (reference path, src/types.jl:112-113) and lowers to a prior (device path)."
struct InitFrom{D<:Distribution}
    d::D
end
(s::InitFrom)(rng) = (x = rand(rng, s.d); x isa Tuple ? collect(Float64, x) : x)

"""
    InitFromSnippet(n, cpu)
A `sample_init` drawn by the log-density's own C snippet on the device (`#define
KABC_USER_SAMPLE_INIT 1` + `kabc_user_sample_init`, include/kabc_costs.h; prior kind
KABC_PRIOR_USER_INIT = 10); `cpu` is the Julia closure `rng -> sample` with the same law, used
when the model runs through KissABC's own path.
"""
struct InitFromSnippet{F}
    n::Int
    cpu::F
end
(s::InitFromSnippet)(rng) = s.cpu(rng)
lower_prior(s::InitFromSnippet) = [KabcPrior(10, 0, (0.0, 0.0, 0.0, 0.0)) for _ in 1:s.n]

const KernelizedDev = ApproxKernelizedPosterior{<:Distribution,<:DeviceCost}
const ThresholdDev = ApproxPosterior{<:Distribution,<:DeviceCost}
const CommonDev = CommonLogDensity{<:Any,<:Union{InitFrom,InitFromSnippet},<:DeviceCost}
const DeviceModel = Union{KernelizedDev,ThresholdDev,CommonDev}

posterior_kind(::ApproxKernelizedPosterior) = Int32(1)
posterior_kind(::ApproxPosterior) = Int32(2)
posterior_kind(::CommonLogDensity) = Int32(3)
eps_of(m::ApproxKernelizedPosterior) = Float64(m.scale)
eps_of(m::ApproxPosterior) = Float64(m.maxcost)
eps_of(::CommonLogDensity) = 1.0
prior_of(m::CommonLogDensity) = m.sample_init isa InitFromSnippet ? m.sample_init : m.sample_init.d
prior_of(m) = m.prior
cost_of(m::CommonLogDensity) = m.lπ
cost_of(m) = m.cost

# ---- context -----------------------------------------------------------------
const CTX = Dict{Int,Ptr{Cvoid}}()
function context(device::Integer = 0)
    get!(CTX, device) do
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:kabc_ctx_create, libkabc), Cint, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, C_NULL, h))
        h[]
    end
end

"with_model(f, model): lowers the model to a kabc_model_t that stays valid while `f` runs"
function with_model(f, model)
    pri = lower_prior(prior_of(model))
    c = cost_of(model)::DeviceCost
    GC.@preserve pri c begin
        cm = KabcModel(pointer(pri), length(pri), posterior_kind(model), eps_of(model),
                       KabcCost(c.id, length(c.params), pointer(c.params), length(c.data), pointer(c.data)))
        f(cm)
    end
end

# ---- AIS: sample(model, AIS(N), Ns; ...) -------------------------------------
mutable struct AISHipState           # AISState of src/KissABC.jl:25-33, device resident
    handle::Ptr{Cvoid}
    cache::Matrix{Float64}           # D x N samples (push_p'ed) of the last generation
    i::Int                           # next column to emit; N + 1 = cache exhausted
end
destroy!(s::AISHipState) = (s.handle != C_NULL && ccall((:kabc_ais_destroy, libkabc), Cint, (Ptr{Cvoid},), s.handle);
                            s.handle = C_NULL)

function generation!(st::AISHipState, ntransitions)
    check(ccall((:kabc_ais_advance, libkabc), Cint, (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Ptr{Cvoid}),
                st.handle, 1, ntransitions, st.cache, C_NULL))
    st.i = 1
end

# the sample a walker is emitted as: scalar (univariate prior), Tuple (Factored) or Vector
function wrap(model, col)
    pr = prior_of(model)
    x = pr isa UnivariateDistribution ? col[1] : vector_valued(pr) || model isa CommonLogDensity ? collect(col) : Tuple(col)
    KissABC.Particle(model isa CommonLogDensity ? x : KissABC.push_p(pr, x))   # Int for discrete components
end

# step(rng, model, spl; retry_sampling) -- replaces src/KissABC.jl:35-64.  As there, the first
# sample is the LAST initial walker and no transition has run yet.
function AbstractMCMC.step(rng::Random.AbstractRNG, model::DeviceModel, spl::AIS;
                           retry_sampling::Int = 100, kwargs...)
    N, D = spl.nparticles, length(model)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    with_model(model) do cm
        check(ccall((:kabc_ais_create, libkabc), Cint, (Ptr{Cvoid}, Ref{KabcModel}, Int64, UInt64, Ref{Ptr{Cvoid}}),
                    context(), cm, N, rand(rng, UInt64), h))   # N < D+5 -> the reference's message
    end
    st = AISHipState(h[], Matrix{Float64}(undef, D, N), N + 1)
    finalizer(destroy!, st)
    check(ccall((:kabc_ais_init, libkabc), Cint, (Ptr{Cvoid}, Int32), st.handle, retry_sampling))
    x = Matrix{Float64}(undef, D, N)
    check(ccall((:kabc_ais_get_ensemble, libkabc), Cint, (Ptr{Cvoid}, Ptr{Float64}), st.handle, x))
    wrap(model, view(x, :, N)), st
end

# step(rng, model, spl, state; ntransitions) -- replaces src/KissABC.jl:66-80: every N-th call
# advances one device generation (every walker `ntransitions` transitions = N reference
# step() calls), the calls in between are served from the host cache.
function AbstractMCMC.step(rng::Random.AbstractRNG, model::DeviceModel, spl::AIS, st::AISHipState;
                           ntransitions::Int = 1, kwargs...)
    st.i > spl.nparticles && generation!(st, ntransitions)
    s = wrap(model, view(st.cache, :, st.i))
    st.i += 1
    s, st
end

# bundle_samples / chainsstack are the reference's own (src/KissABC.jl:82-104): the samples are
# ordinary `Particle`s

# sample(model, AIS(N), MCMCThreads(), Ns, Nc) -- src/KissABC.jl:96-104,108: the chains are ONE
# batch handle (chain = a grid dimension of every launch), not Nc tasks
function AbstractMCMC.sample(rng::Random.AbstractRNG, model::DeviceModel, spl::AIS, ::MCMCThreads,
                             Ns::Integer, Nc::Integer; ntransitions::Int = 1, discard_initial::Int = 0,
                             retry_sampling::Int = 100, chain_type::Type = Any, kwargs...)
    N, D = spl.nparticles, length(model)
    seeds = rand(rng, UInt64, Nc)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    with_model(model) do cm
        check(ccall((:kabc_ais_create_batch, libkabc), Cint,
                    (Ptr{Cvoid}, Ref{KabcModel}, Int64, Int32, Ptr{UInt64}, Ref{Ptr{Cvoid}}),
                    context(), cm, N, Nc, seeds, h))
    end
    try
        check(ccall((:kabc_ais_init, libkabc), Cint, (Ptr{Cvoid}, Int32), h[], retry_sampling))
        gd, gk = cld(discard_initial, N), max(1, cld(Ns, N))
        gd > 0 && check(ccall((:kabc_ais_advance, libkabc), Cint, (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Ptr{Cvoid}),
                              h[], gd, ntransitions, C_NULL, C_NULL))
        tr = Array{Float64,4}(undef, D, N, Nc, gk)             # [gen][chain][N][D], column-major
        check(ccall((:kabc_ais_advance, libkabc), Cint, (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Ptr{Cvoid}),
                    h[], gk, ntransitions, tr, C_NULL))
        chains = map(1:Nc) do c
            cols = reshape(permutedims(view(tr, :, :, c, :), (1, 2, 3)), D, N * gk)
            samples = [wrap(model, view(cols, :, j)) for j in 1:Ns]
            AbstractMCMC.bundle_samples(samples, model, spl, nothing, chain_type)
        end
        return AbstractMCMC.chainsstack(AbstractMCMC.tighten_eltype(chains))
    finally
        ccall((:kabc_ais_destroy, libkabc), Cint, (Ptr{Cvoid},), h[])
    end
end

# ---- walker-sharded AIS over several GPUs (include/kabc.h "multi-GPU") ---------------------
# One process, n GPUs: kabc_comm_init_all + kabc_ais_create_dist + the *_multi drivers; the
# all-gather after every half-generation is issued inside the library (RCCL, or the P2P pull
# kernel with backend = :p2p).  For one process per GPU (Distributed / MPI launch) use
# `unique_id()` on rank 0, ship the 128 bytes, `comm_init_rank(id, rank, world; device)` on
# every rank and pass the communicator to `sample_sharded(model, spl, comm, Ns)`.
unique_id() = (id = Vector{UInt8}(undef, 128); check(ccall((:kabc_comm_unique_id, libkabc), Cint, (Ptr{UInt8},), id)); id)
function comm_init_rank(id::Vector{UInt8}, rank::Integer, world::Integer; device::Integer = rank)
    c = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:kabc_comm_init_rank, libkabc), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32, Ref{Ptr{Cvoid}}),
                context(device), id, rank, world, c))
    c[]
end

function sample_sharded(model::DeviceModel, spl::AIS, devices::AbstractVector{<:Integer}, Ns::Integer;
                        rng = Random.GLOBAL_RNG, ntransitions::Int = 1, discard_initial::Int = 0,
                        retry_sampling::Int = 100, backend::Symbol = :rccl)
    n, N, D = length(devices), spl.nparticles, length(model)
    ctxs, comms = fill(C_NULL, n), fill(C_NULL, n)
    check(ccall((:kabc_comm_init_all, libkabc), Cint, (Int32, Ptr{Int32}, Int32, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}),
                n, Int32.(devices), backend == :p2p ? 2 : 1, ctxs, comms))
    hs, seed = fill(C_NULL, n), rand(rng, UInt64)
    try
        with_model(model) do cm
            for r in 1:n
                h = Ref{Ptr{Cvoid}}(C_NULL)
                check(ccall((:kabc_ais_create_dist, libkabc), Cint, (Ptr{Cvoid}, Ref{KabcModel}, Int64, UInt64, Ref{Ptr{Cvoid}}),
                            comms[r], cm, N, seed, h))
                hs[r] = h[]
            end
        end
        check(ccall((:kabc_ais_init_multi, libkabc), Cint, (Ptr{Ptr{Cvoid}}, Int32, Int32), hs, n, retry_sampling))
        adv(g) = check(ccall((:kabc_ais_advance_multi, libkabc), Cint, (Ptr{Ptr{Cvoid}}, Int32, Int64, Int32, Ptr{Cvoid}),
                             hs, n, g, ntransitions, C_NULL))
        adv(cld(discard_initial, N))
        gk = max(1, cld(Ns, N))
        out = Array{Float64,3}(undef, D, N, gk)
        for g in 1:gk                                             # the ensemble after each generation
            adv(1)
            check(ccall((:kabc_ais_get_ensemble, libkabc), Cint, (Ptr{Cvoid}, Ptr{Float64}), hs[1], view(out, :, :, g)))
        end
        cols = reshape(out, D, N * gk)
        samples = [wrap(model, view(cols, :, j)) for j in 1:Ns]
        return AbstractMCMC.bundle_samples(samples, model, spl, nothing, Any)
    finally
        foreach(h -> h != C_NULL && ccall((:kabc_ais_destroy, libkabc), Cint, (Ptr{Cvoid},), h), hs)
        foreach(c -> c != C_NULL && ccall((:kabc_comm_destroy, libkabc), Cint, (Ptr{Cvoid},), c), comms)
    end
end

# one process per GPU: every rank calls this with its communicator (comm_init_rank); the
# returned samples are identical on all ranks
function sample_sharded(model::DeviceModel, spl::AIS, comm::Ptr{Cvoid}, Ns::Integer; seed::UInt64,
                        ntransitions::Int = 1, discard_initial::Int = 0, retry_sampling::Int = 100)
    N, D = spl.nparticles, length(model)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    with_model(model) do cm
        check(ccall((:kabc_ais_create_dist, libkabc), Cint, (Ptr{Cvoid}, Ref{KabcModel}, Int64, UInt64, Ref{Ptr{Cvoid}}),
                    comm, cm, N, seed, h))                      # the same seed on every rank
    end
    try
        check(ccall((:kabc_ais_init, libkabc), Cint, (Ptr{Cvoid}, Int32), h[], retry_sampling))
        adv(g) = check(ccall((:kabc_ais_advance, libkabc), Cint, (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Ptr{Cvoid}),
                             h[], g, ntransitions, C_NULL, C_NULL))
        adv(cld(discard_initial, N))
        gk = max(1, cld(Ns, N))
        out = Array{Float64,3}(undef, D, N, gk)
        for g in 1:gk
            adv(1)
            check(ccall((:kabc_ais_get_ensemble, libkabc), Cint, (Ptr{Cvoid}, Ptr{Float64}), h[], view(out, :, :, g)))
        end
        cols = reshape(out, D, N * gk)
        return AbstractMCMC.bundle_samples([wrap(model, view(cols, :, j)) for j in 1:Ns], model, spl, nothing, Any)
    finally
        ccall((:kabc_ais_destroy, libkabc), Cint, (Ptr{Cvoid},), h[])
    end
end

# ---- smc / ABCDE / pfilter ---------------------------------------------------
particles_of(prior, theta, keep) = begin
    D = size(theta, 1)
    P = [Particles(theta[k, keep]) for k in 1:D]                   # src/smc.jl:203
    length(P) == 1 ? first(P) : P
end
kcost(c::DeviceCost) = KabcCost(c.id, length(c.params), pointer(c.params), length(c.data), pointer(c.data))

# smc(prior, cost::DeviceCost; ...) -- replaces src/smc.jl:92-206, same keywords and defaults
function KissABC.smc(prior::Distribution, cost::DeviceCost; rng = Random.GLOBAL_RNG,
                     nparticles::Int = 100, alpha = 0.95, mcmc_retrys::Int = 0, mcmc_tol = 0.015,
                     epstol = 0.0, r_epstol = (1 - alpha)^1.5 / 50, min_r_ess = alpha^2,
                     max_stretch = 2.0, verbose::Bool = false, parallel::Bool = false,
                     comm::Ptr{Cvoid} = C_NULL, shard::Symbol = :cost_loop)
    # comm (a communicator of init_rank): collective over its ranks, every rank passes the same arguments
    # and an `rng` in the same state, and receives the same result -- smc's, bit for bit.
    # shard = :cost_loop  the propose / accept pass is shared out (the reference's `parallel = true` leg);
    #         :particles  the ranks own their particles, the ε-selection is sharded as well (SURVEY §8e)
    pri = lower_prior(prior)
    D, N = length(pri), nparticles
    theta = Matrix{Float64}(undef, D, max(N, 1))
    C = Vector{Float64}(undef, max(N, 1))
    alive = zeros(UInt8, max(N, 1))
    o = KabcSmcOpts(N, alpha, mcmc_retrys, verbose, mcmc_tol, epstol, r_epstol, min_r_ess,
                    max_stretch, rand(rng, UInt64), 0)
    r = KabcSmcResult(pointer(theta), pointer(C), pointer(alive), 0.0, 0, 0, 0, 0, C_NULL, 0, 0.0, 0)
    GC.@preserve pri cost theta C alive begin
        if comm == C_NULL
            check(ccall((:kabc_smc_run, libkabc), Cint,
                        (Ptr{Cvoid}, Ptr{KabcPrior}, Int32, Ref{KabcCost}, Ref{KabcSmcOpts}, Ref{KabcSmcResult}),
                        context(), pri, D, kcost(cost), o, r))      # argument errors: the reference's messages
        else
            check(ccall((:kabc_smc_run_dist_mode, libkabc), Cint,
                        (Ptr{Cvoid}, Ptr{KabcPrior}, Int32, Ref{KabcCost}, Ref{KabcSmcOpts}, Int32, Ref{KabcSmcResult}),
                        comm, pri, D, kcost(cost), o, Int32(shard == :particles ? 1 : 0), r))
        end
    end
    (P = particles_of(prior, theta, findall(!=(0x00), alive)), C = C, ϵ = r.eps)   # src/smc.jl:205
end

# how this thread's last smc was driven (kabc_smc_dist_stats): iterations, collectives issued, host looks,
# selections decided by the one exchange / phase by phase, passes, batched
function smc_dist_stats()
    out = zeros(Int64, 8)
    ccall((:kabc_smc_dist_stats, libkabc), Cvoid, (Ptr{Int64},), out)
    (iterations = out[1], collectives = out[2], host_looks = out[3], one_exchange_selections = out[4],
     phase_by_phase_selections = out[5], passes = out[6], batched = out[7] != 0)
end

# ABCDE(prior, cost::DeviceCost, ϵ_target; ...) -- replaces src/smc.jl:347-430
function KissABC.ABCDE(prior::Distribution, cost::DeviceCost, ϵ_target; nparticles = 50, generations = 20,
                       α = 0, parallel = false, earlystop = false, verbose = true,
                       rng = Random.GLOBAL_RNG, proposal_width = 1.0)
    pri = lower_prior(prior)
    D, N = length(pri), nparticles
    theta = Matrix{Float64}(undef, D, N)
    C = Vector{Float64}(undef, N)
    o = KabcAbcdeOpts(N, generations, ϵ_target, α, proposal_width, earlystop, verbose, rand(rng, UInt64))
    r = KabcAbcdeResult(pointer(theta), pointer(C), 0, 0, 0, 0)
    GC.@preserve pri cost theta C begin
        check(ccall((:kabc_abcde_run, libkabc), Cint,
                    (Ptr{Cvoid}, Ptr{KabcPrior}, Int32, Ref{KabcCost}, Ref{KabcAbcdeOpts}, Ref{KabcAbcdeResult}),
                    context(), pri, D, kcost(cost), o, r))
    end
    (P = particles_of(prior, theta, 1:N), C = Particles(C), reached_ϵ = r.reached_eps != 0)   # src/smc.jl:425-430
end

# pfilter(prior, cost::DeviceCost, N; ...) -- replaces src/smc.jl:275-340
function KissABC.pfilter(prior::Distribution, cost::DeviceCost, N; rng = Random.GLOBAL_RNG, q = 0.7,
                         eff_tol = 0.1, epstol = -Inf, max_iters = Inf, proposal_width = 0.75,
                         verbose = false, parallel = false)
    pri = lower_prior(prior)
    D = length(pri)
    Neff = ccall((:kabc_pfilter_nparticles, libkabc), Int64, (Int64, Float64, Int32), N, q, D)   # :276-279
    theta = Matrix{Float64}(undef, D, Neff)
    C = Vector{Float64}(undef, Neff)
    o = KabcPfilterOpts(N, q, eff_tol, epstol, proposal_width, isinf(max_iters) ? -1 : floor(Int64, max_iters),
                        verbose, 0, rand(rng, UInt64))
    r = KabcPfilterResult(pointer(theta), pointer(C), 0.0, 0.0, 0, 0, 0)
    GC.@preserve pri cost theta C begin
        check(ccall((:kabc_pfilter_run, libkabc), Cint,
                    (Ptr{Cvoid}, Ptr{KabcPrior}, Int32, Ref{KabcCost}, Ref{KabcPfilterOpts}, Ref{KabcPfilterResult}),
                    context(), pri, D, kcost(cost), o, r))
    end
    (P = particles_of(prior, theta, 1:Neff), C = Particles(C))     # src/smc.jl:334-340
end

"""
    compile_model(model; families = 0) / compile_model(prior, cost; families = 0)
Kernels specialised for ONE model (kabc_compile_model, include/kabc.h): the prior tuple's
families and parameters become compile-time constants of a translation unit compiled by hipRTC.
Afterwards `sample` / `smc` / `ABCDE` / `pfilter` on the same prior components and cost use them;
results keep their bits.  Returns the registration's handle (0: left to the prebuilt kernels).
"""
function compile_model(model::DeviceModel; families::Integer = 0)
    h = Ref{Int32}(0)
    with_model(model) do cm
        check(ccall((:kabc_compile_model, libkabc), Cint, (Ref{KabcModel}, Int32, Ref{Int32}), cm, Int32(families), h))
    end
    h[]
end
function compile_model(prior::Distribution, cost::DeviceCost; families::Integer = 2)
    pri = lower_prior(prior)
    h = Ref{Int32}(0)
    GC.@preserve pri cost begin
        cm = KabcModel(pointer(pri), length(pri), 0, 1.0, kcost(cost))
        check(ccall((:kabc_compile_model, libkabc), Cint, (Ref{KabcModel}, Int32, Ref{Int32}), cm, Int32(families), h))
    end
    h[]
end
release_model(h::Integer) = check(ccall((:kabc_model_release, libkabc), Cint, (Int32,), Int32(h)))
"""
    prefetch_model(model; families = 0)
The library specialises an eligible model ON ITS OWN without ever waiting for the compiler (a
detached worker process; include/kabc.h "THE DEFAULT" -- what Julia's per-type compilation of
`logpdf(::Factored)` gives the reference, src/priors.jl:11,30-36).  This starts that compilation
ahead of the first `sample` / `smc`; it returns at once.
"""
function prefetch_model(model::DeviceModel; families::Integer = 0)
    with_model(model) do cm
        check(ccall((:kabc_prefetch_model, libkabc), Cint, (Ref{KabcModel}, Int32), cm, Int32(families)))
    end
end
"""
    set_specialize(mode)

`:env` (KABC_SPECIALIZE decides), `:off` (never specialise, never start the compiler worker),
`:blocking` (compile at first sight), `:background` (the worker process) -- `kabc_set_specialize`.
"""
function set_specialize(mode::Symbol)
    m = mode === :env ? -1 : mode === :off ? 0 : mode === :blocking ? 1 : mode === :background ? 2 :
        error("set_specialize: :env, :off, :blocking or :background")
    check(ccall((:kabc_set_specialize, libkabc), Cint, (Int32,), Int32(m)))
end
"the code-object cache directory in use (\"\": none -- disabled, or no candidate nobody else can write to)"
function rtc_cache_dir()
    buf = Vector{Cchar}(undef, 4096)
    ccall((:kabc_rtc_cache_dir, libkabc), Int32, (Ptr{Cchar}, Int32), buf, Int32(length(buf)))
    GC.@preserve buf unsafe_string(pointer(buf))
end
"(started, loaded, failed, cache_hits): process-wide counters of the background specialisations"
function spec_counters()
    out = zeros(UInt64, 4)
    check(ccall((:kabc_spec_counters, libkabc), Cint, (Ptr{UInt64},), out))
    (started = out[1], loaded = out[2], failed = out[3], cache_hits = out[4])
end

export DeviceCost, UserCost, UserPrior, compile_model, release_model, prefetch_model, spec_counters, set_specialize, rtc_cache_dir, UserMvPrior, InitFrom, InitFromSnippet, GaussDist, Rosenbrock, HierGaussSim, NormalMeanStdSim, DiracSq,
       AbsDiff, NormShell, NoisyQuadDU, Mixture, NoisyBanana, WienerRms, sample_sharded, unique_id,
       comm_init_rank
end # module
