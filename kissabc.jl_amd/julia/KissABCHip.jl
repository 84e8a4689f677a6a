# KissABCHip.jl -- thin `ccall` layer that puts the MI355X walker-update path
# (libkabc_hip.so, C ABI in include/kabc.h) behind KissABC.jl's own surface:
#
#     ApproxKernelizedPosterior(prior, cost::DeviceCost, scale)   # KissABC's type, unchanged
#     sample(model, AISHip(N), Ns; ntransitions, discard_initial, retry_sampling)
#     smc(prior, cost::DeviceCost; kwargs...)                      # same keywords/defaults
#
# It contains no numerics: it lowers `Factored`/Distributions objects to
# `kabc_prior_t`, a `DeviceCost` to `kabc_cost_t`, calls the library and wraps
# the result in `Particles` exactly as src/KissABC.jl:82-104 and src/smc.jl:200-205 do.
#
# NOTE: Julia is not available in the build image of this repository, so this file
# has been checked by eye only; the identical call sequence is exercised through
# the Python ctypes mirror (kissabc.jl_amd/api.py) by tests/.
module KissABCHip

using KissABC, Random
import AbstractMCMC
import KissABC: Factored, ApproxKernelizedPosterior, ApproxPosterior, Particles
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

check(st) = st == 0 ? nothing :
    error(unsafe_string(ccall((:kabc_last_error, libkabc), Cstring, ())))   # reference's text

# ---- DeviceCost: the `cost` argument on the device path ----------------------
# A DeviceCost is also callable on the CPU (`cpu` holds the same formula as a
# Julia closure), so ONE model object runs through KissABC's own AIS/smc and
# through the HIP path.  ids/formulas: include/kabc_costs.h.
struct DeviceCost{F}
    id::Int32
    params::Vector{Float64}
    data::Vector{Float64}
    cpu::F
end
(c::DeviceCost)(x) = c.cpu(x)
GaussDist(c) = DeviceCost(Int32(1), collect(Float64, c), Float64[], x -> sqrt(sum(abs2, x .- c)))
Rosenbrock() = DeviceCost(Int32(2), Float64[], Float64[],
    x -> sqrt(sum(100 * (x[k+1] - x[k]^2)^2 + (1 - x[k])^2 for k in 1:length(x)-1)))
DiracSq(t = 1.5) = DeviceCost(Int32(5), [Float64(t)], Float64[], x -> abs(x[1]^2 + 1 - t))
AbsDiff(t) = DeviceCost(Int32(6), [Float64(t)], Float64[], x -> abs(x[1] - t))
NormShell(t) = DeviceCost(Int32(7), [Float64(t)], Float64[], x -> abs(sqrt(sum(abs2, x)) - t))
NoisyBanana(p = 0.0) = DeviceCost(Int32(10), [Float64(p)], Float64[],
    ((x, y),) -> rand() < p ? Inf : 50 * (x + randn() * 0.01 - y^2)^2 + (y - 1 + randn() * 0.01)^2)
# ... the remaining ids (hier_gauss_sim, normal_meanstd_sim, noisy_quad_du, mixture,
# wiener_rms) follow the same pattern.

"""
    UserCost(csrc, dims; params, data, cpu)
A DeviceCost from a C snippet defining `kabc_user_cost` (include/kabc_costs.h,
KABC_COST_USER): compiled with hipcc for gfx950 together with csrc/user_plugin.inc and
registered with `kabc_register_cost_plugin`.  `cpu` is the Julia closure with the same
formula (used when the model runs through KissABC's own AIS/smc).
"""
function UserCost(csrc::String, dims; params = Float64[], data = Float64[], cpu = x -> NaN)
    root = normpath(joinpath(@__DIR__, "..", ".."))
    cond = join(("(D) == $d" for d in dims), " || ")
    text = "#define KABC_USER_DIM_OK(D) ($cond)\n#include <hip/hip_runtime.h>\n" *
           "#include \"kabc_philox.h\"\n" * csrc *
           "\n#define KABC_USER_COST_DEFINED 1\n#include \"user_plugin.inc\"\n"
    dir = mktempdir(); src = joinpath(dir, "user.hip"); so = joinpath(dir, "libkabc_user.so")
    write(src, text)
    run(`/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950
         -I $(joinpath(root, "include")) -I $(joinpath(root, "kissabc.jl_amd", "csrc"))
         -shared -o $so $src`)
    id = Ref{Int32}(0)
    check(ccall((:kabc_register_cost_plugin, libkabc), Cint, (Cstring, Ref{Int32}), so, id))
    DeviceCost(id[], collect(Float64, params), collect(Float64, data), cpu)
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
lower(d::Factored) = KabcPrior[lower(c) for c in d.p]
lower_prior(d::UnivariateDistribution) = KabcPrior[lower(d)]
lower_prior(d::Factored) = lower(d)

posterior_kind(::ApproxKernelizedPosterior) = Int32(1)
posterior_kind(::ApproxPosterior) = Int32(2)
eps_of(m::ApproxKernelizedPosterior) = Float64(m.scale)
eps_of(m::ApproxPosterior) = Float64(m.maxcost)

# ---- context -----------------------------------------------------------------
const CTX = Ref{Ptr{Cvoid}}(C_NULL)
function context(device = 0)
    if CTX[] == C_NULL
        check(ccall((:kabc_ctx_create, libkabc), Cint, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}),
                    device, C_NULL, CTX))
    end
    CTX[]
end

# ---- AIS ---------------------------------------------------------------------
"AISHip(N): AIS(N) executed on the GPU (src/KissABC.jl:21-23)."
struct AISHip <: AbstractMCMC.AbstractSampler
    nparticles::Int
end

mutable struct AISHipState           # AISState of src/KissABC.jl:25-33, device resident
    handle::Ptr{Cvoid}
    cache::Matrix{Float64}           # D x N samples of the last generation
    i::Int
end

function create(model, spl::AISHip, seed::UInt64)
    pri = lower_prior(model.prior)
    c = model.cost::DeviceCost
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve pri c begin
        cm = KabcModel(pointer(pri), length(pri), posterior_kind(model), eps_of(model),
                       KabcCost(c.id, length(c.params), pointer(c.params), length(c.data),
                                pointer(c.data)))
        check(ccall((:kabc_ais_create, libkabc), Cint,
                    (Ptr{Cvoid}, Ref{KabcModel}, Int64, UInt64, Ref{Ptr{Cvoid}}),
                    context(), cm, spl.nparticles, seed, h))
    end
    h[]
end

function generation!(st::AISHipState, ntransitions)
    check(ccall((:kabc_ais_advance, libkabc), Cint,
                (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Ptr{Cvoid}),
                st.handle, 1, ntransitions, st.cache, C_NULL))
    st.i = 1
end

wrap(model, col) = KissABC.Particle(length(col) == 1 ? col[1] : Tuple(col))

# step(rng, model, spl; retry_sampling) -- replaces src/KissABC.jl:35-64
function AbstractMCMC.step(rng::Random.AbstractRNG, model::AbstractMCMC.AbstractModel,
                           spl::AISHip; retry_sampling::Int = 100, ntransitions::Int = 1, kwargs...)
    h = create(model, spl, rand(rng, UInt64))
    check(ccall((:kabc_ais_init, libkabc), Cint, (Ptr{Cvoid}, Int32), h, retry_sampling))
    st = AISHipState(h, Matrix{Float64}(undef, length(model), spl.nparticles), 1)
    finalizer(s -> ccall((:kabc_ais_destroy, libkabc), Cint, (Ptr{Cvoid},), s.handle), st)
    generation!(st, ntransitions)
    wrap(model, view(st.cache, :, spl.nparticles)), st
end

# step(rng, model, spl, state; ntransitions) -- replaces src/KissABC.jl:66-80:
# every N-th call advances one device generation (N x ntransitions transitions),
# the calls in between are served from the host cache.
function AbstractMCMC.step(rng::Random.AbstractRNG, model::AbstractMCMC.AbstractModel,
                           spl::AISHip, st::AISHipState; ntransitions::Int = 1, kwargs...)
    st.i > spl.nparticles && generation!(st, ntransitions)
    s = wrap(model, view(st.cache, :, st.i))
    st.i += 1
    s, st
end

AbstractMCMC.bundle_samples(samples::Vector{<:KissABC.Particle}, m::AbstractMCMC.AbstractModel,
                            ::AISHip, state, T::Type; kwargs...) =
    AbstractMCMC.bundle_samples(samples, m, KissABC.AIS(1), state, T; kwargs...)

# ---- smc ---------------------------------------------------------------------
# smc(prior, cost::DeviceCost; ...) -- replaces src/smc.jl:92-206
function KissABC.smc(prior::Distribution, cost::DeviceCost; rng = Random.GLOBAL_RNG,
                     nparticles::Int = 100, alpha = 0.95, mcmc_retrys::Int = 0, mcmc_tol = 0.015,
                     epstol = 0.0, r_epstol = (1 - alpha)^1.5 / 50, min_r_ess = alpha^2,
                     max_stretch = 2.0, verbose::Bool = false, parallel::Bool = false)
    pri = lower_prior(prior)
    D, N = length(pri), nparticles
    theta = Matrix{Float64}(undef, D, max(N, 1))
    C = Vector{Float64}(undef, max(N, 1))
    alive = zeros(UInt8, max(N, 1))
    o = KabcSmcOpts(N, alpha, mcmc_retrys, verbose, mcmc_tol, epstol, r_epstol, min_r_ess,
                    max_stretch, rand(rng, UInt64), 0)
    r = KabcSmcResult(pointer(theta), pointer(C), pointer(alive), 0.0, 0, 0, 0, 0, C_NULL, 0, 0.0, 0)
    GC.@preserve pri cost theta C alive begin
        kc = KabcCost(cost.id, length(cost.params), pointer(cost.params), length(cost.data),
                      pointer(cost.data))
        check(ccall((:kabc_smc_run, libkabc), Cint,
                    (Ptr{Cvoid}, Ptr{KabcPrior}, Int32, Ref{KabcCost}, Ref{KabcSmcOpts},
                     Ref{KabcSmcResult}), context(), pri, D, kc, o, r))
    end
    keep = findall(!=(0x00), alive)
    P = [Particles(theta[k, keep]) for k in 1:D]      # src/smc.jl:203
    length(P) == 1 && (P = first(P))
    (P = P, C = C, ϵ = r.eps)
end

export AISHip, DeviceCost, UserCost, GaussDist, Rosenbrock, DiracSq, AbsDiff, NormShell, NoisyBanana
end # module
