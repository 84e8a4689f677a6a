# bench/cpu_reference.jl -- BASELINE.md §3 (i): the reference Julia CPU path, timed on the same
# box as bench.py, IF a `julia` with KissABC installed happens to be present (it is not in the
# build image; bench.py's `cpu_baseline` leg -- the C restatement of the same serial schedule --
# is what is always reported).
#
#   julia -t 1 bench/cpu_reference.jl [N=65536] [D=8] [ntransitions=100] [seconds=12]
#
# Workload = bench.py's (SURVEY 8d C3): prior Factored(Uniform(-5,5))^D, cost
# sqrt(sum 100 (x[k+1]-x[k]^2)^2 + (1-x[k])^2), ApproxKernelizedPosterior scale 1.0.  The state
# comes from the init `step` (src/KissABC.jl:35-64); then KissABC.transition! (src/transition.jl:67-82)
# is driven directly over i = 1:N for as many sweeps as fit the time budget, so AbstractMCMC's
# per-sample overhead is excluded.  Prints one JSON line comparable with bench.py's
# cpu_baseline: {"value": evals/s, "unit": "evals/s", "cores": 1, "kind": "reference", ...}.
using KissABC, Random, Distributions
import AbstractMCMC

N = length(ARGS) >= 1 ? parse(Int, ARGS[1]) : 65536
D = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 8
nt = length(ARGS) >= 3 ? parse(Int, ARGS[3]) : 100
budget = length(ARGS) >= 4 ? parse(Float64, ARGS[4]) : 12.0

prior = Factored((Uniform(-5, 5) for _ in 1:D)...)
cost(x) = sqrt(sum(100 * (x[k+1] - x[k]^2)^2 + (1 - x[k])^2 for k in 1:D-1))
model = ApproxKernelizedPosterior(prior, cost, 1.0)
rng = Random.MersenneTwister(1)
_, state = AbstractMCMC.step(rng, model, AIS(N))                       # prior draws + loglike
X, LD = state.sample, state.loglikelihood

function sweep!(model, X, LD, rng, nt, steps)                           # `steps` reference step() calls
    i = 1
    for _ in 1:steps
        for _ in 1:nt
            KissABC.transition!(model, X, LD, i, rng)
        end
        i = 1 + (i % length(X))
    end
end
sweep!(model, X, LD, rng, nt, 256)                                      # compile
steps, done = 4096, 0
t0 = time()
while time() - t0 < budget
    sweep!(model, X, LD, rng, nt, steps)
    global done += steps
end
el = time() - t0
println("{\"value\": $(done * nt / el), \"unit\": \"evals/s\", \"cores\": 1, \"kind\": \"reference\", ",
        "\"sample\": \"KissABC.transition! driven over $(done) step() calls x ntransitions=$(nt), ",
        "N=$(N) D=$(D) rosenbrock, $(round(el, digits = 1))s on 1 host core (julia -t 1)\"}")
