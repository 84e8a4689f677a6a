"""ctypes mirror of include/kabc.h (structs, enums, prototypes).

Used by the product binding (_lib.py) and, for the struct layouts only, by the
oracle's test binding (oracle/oracle.py).  Nothing here computes anything.
"""
import ctypes as C

KABC_MAX_DIM = 16
KABC_MAX_DIM_DYN = 256   # AIS only: run-time-dimension kernels beyond KABC_MAX_DIM
KABC_VERSION = 321   # include/kabc.h
KABC_COMM_ID_BYTES = 128
KABC_MAX_EXCHANGE_CHUNKS = 16
KABC_COMM_MAX_WORLD = 16
COMM_RCCL, COMM_P2P = 1, 2

# kabc_status_t
KABC_OK, KABC_ERR_INVALID_ARG, KABC_ERR_RETRY_EXHAUSTED, KABC_ERR_INVALID_STATE, \
    KABC_ERR_DEVICE, KABC_ERR_UNSUPPORTED, KABC_ERR_NAN_COST = range(7)

# kabc_prior_kind_t
PRIOR_UNIFORM, PRIOR_NORMAL, PRIOR_TRUNCNORMAL, PRIOR_BETA, PRIOR_DISCRETE_UNIFORM, \
    PRIOR_NEGBINOMIAL, PRIOR_EXPONENTIAL, PRIOR_GAMMA, PRIOR_LOGNORMAL = range(1, 10)
PRIOR_USER_INIT = 10
PRIOR_MVNORMAL = 11
PRIOR_USER = 100         # kinds >= this: families compiled at run time (kabc_compile_prior_plugin)
FAMILY_AIS, FAMILY_SMC, FAMILY_ABCDE, FAMILY_PFILTER, FAMILY_AIS_SMALL = 1, 2, 4, 8, 16

POSTERIOR_KERNELIZED, POSTERIOR_THRESHOLD, POSTERIOR_COMMON = 1, 2, 3

# DeviceCost ids (include/kabc_costs.h)
COST_GAUSS_DIST, COST_ROSENBROCK, COST_HIER_GAUSS_SIM, COST_NORMAL_MEANSTD_SIM, COST_DIRAC_SQ, \
    COST_ABS_DIFF, COST_NORM_SHELL, COST_NOISY_QUAD_DU, COST_MIXTURE, COST_NOISY_BANANA, \
    COST_WIENER_RMS = range(1, 12)

# stream domains (include/kabc_philox.h)
DOM_AIS_INIT, DOM_AIS_INIT_COST, DOM_AIS_MOVE, DOM_AIS_COST, DOM_SMC_INIT, DOM_SMC_INIT_COST, \
    DOM_SMC_MOVE, DOM_SMC_COST = range(1, 9)

c_double_p = C.POINTER(C.c_double)


class Prior(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("p", C.c_double * 4)]


class Cost(C.Structure):
    _fields_ = [("id", C.c_int32), ("nparams", C.c_int32), ("params", c_double_p),
                ("ndata", C.c_int64), ("data", c_double_p)]


class Model(C.Structure):
    _fields_ = [("prior", C.POINTER(Prior)), ("D", C.c_int32), ("posterior", C.c_int32),
                ("eps", C.c_double), ("cost", Cost)]


class Stats(C.Structure):
    _fields_ = [("proposals", C.c_uint64), ("cost_evals", C.c_uint64), ("accepted", C.c_uint64)]


class SmcOpts(C.Structure):
    _fields_ = [("nparticles", C.c_int64), ("alpha", C.c_double), ("mcmc_retrys", C.c_int32),
                ("verbose", C.c_int32), ("mcmc_tol", C.c_double), ("epstol", C.c_double),
                ("r_epstol", C.c_double), ("min_r_ess", C.c_double), ("max_stretch", C.c_double),
                ("seed", C.c_uint64), ("max_iterations", C.c_int64)]


class SmcIter(C.Structure):
    _fields_ = [("eps", C.c_double), ("ess", C.c_int64), ("accepted", C.c_int64),
                ("resampled", C.c_int32), ("flag", C.c_int32), ("mcmc_passes", C.c_int32),
                ("reserved", C.c_int32)]


class SmcResult(C.Structure):
    _fields_ = [("theta", c_double_p), ("cost", c_double_p), ("alive", C.POINTER(C.c_uint8)),
                ("eps", C.c_double), ("iterations", C.c_int64), ("n_alive", C.c_int64),
                ("cost_evals", C.c_uint64), ("proposals", C.c_uint64),
                ("iter_log", C.POINTER(SmcIter)), ("iter_log_cap", C.c_int64),
                ("kernel_ms_mcmc", C.c_double), ("mcmc_launches", C.c_int64)]


class AbcdeOpts(C.Structure):
    _fields_ = [("nparticles", C.c_int64), ("generations", C.c_int64), ("eps_target", C.c_double),
                ("alpha", C.c_double), ("proposal_width", C.c_double), ("earlystop", C.c_int32),
                ("verbose", C.c_int32), ("seed", C.c_uint64)]


class AbcdeResult(C.Structure):
    _fields_ = [("theta", c_double_p), ("cost", c_double_p), ("reached_eps", C.c_int32),
                ("reserved", C.c_int32), ("generations_run", C.c_int64), ("nsims", C.c_uint64)]


class PfilterOpts(C.Structure):
    _fields_ = [("nparticles", C.c_int64), ("q", C.c_double), ("eff_tol", C.c_double),
                ("epstol", C.c_double), ("proposal_width", C.c_double), ("max_iters", C.c_int64),
                ("verbose", C.c_int32), ("reserved", C.c_int32), ("seed", C.c_uint64)]


class PfilterResult(C.Structure):
    _fields_ = [("theta", c_double_p), ("cost", c_double_p), ("eps", C.c_double),
                ("eff", C.c_double), ("iterations", C.c_int64), ("nreps", C.c_uint64),
                ("cost_evals", C.c_uint64)]


# every symbol include/kabc.h declares: name -> (restype, argtypes)
VP = C.c_void_p
PROTOTYPES = {
    "kabc_version": (C.c_int32, []),
    "kabc_abi_sizeof": (C.c_int32, [C.c_int32]),
    "kabc_abi_offsetof": (C.c_int32, [C.c_int32, C.c_int32]),
    "kabc_mvnormal_register": (C.c_int, [c_double_p, c_double_p, C.c_int32, C.POINTER(C.c_int32)]),
    "kabc_last_error": (C.c_char_p, []),
    "kabc_device_count": (C.c_int32, []),
    "kabc_ctx_create": (C.c_int, [C.c_int32, VP, C.POINTER(VP)]),
    "kabc_ctx_destroy": (C.c_int, [VP]),
    "kabc_ctx_synchronize": (C.c_int, [VP]),
    "kabc_math_probe": (C.c_int, [VP, C.c_int32, C.c_int64, c_double_p, c_double_p]),
    "kabc_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(VP)]),
    "kabc_host_free": (C.c_int, [VP]),
    "kabc_factored_logpdf": (C.c_int, [VP, C.POINTER(Prior), C.c_int32, C.c_int64, c_double_p,
                                       c_double_p]),
    "kabc_factored_push_p": (C.c_int, [VP, C.POINTER(Prior), C.c_int32, C.c_int64, c_double_p,
                                       c_double_p]),
    "kabc_factored_rand": (C.c_int, [VP, C.POINTER(Prior), C.c_int32, C.c_uint64, C.c_uint32,
                                     C.c_int64, C.c_int64, C.c_uint64, c_double_p]),
    "kabc_register_cost_plugin": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32)]),
    "kabc_plugin_precompile": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "kabc_compile_cost_plugin": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                                         C.POINTER(C.c_int32)]),
    "kabc_compile_prior_plugin": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]),
    "kabc_compile_mvprior_plugin": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32)]),
    "kabc_compile_model": (C.c_int, [C.POINTER(Model), C.c_int32, C.POINTER(C.c_int32)]),
    "kabc_model_release": (C.c_int, [C.c_int32]),
    "kabc_prefetch_model": (C.c_int, [C.POINTER(Model), C.c_int32]),
    "kabc_spec_counters": (C.c_int, [C.POINTER(C.c_uint64)]),
    "kabc_set_specialize": (C.c_int, [C.c_int32]),
    "kabc_rtc_cache_dir": (C.c_int32, [C.c_char_p, C.c_int32]),
    "kabc_rtc_worker_main": (C.c_int32, [C.c_char_p]),
    "kabc_ais_spec_state": (C.c_int, [VP, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "kabc_ais_driver": (C.c_int32, [VP]),
    "kabc_ais_create": (C.c_int, [VP, C.POINTER(Model), C.c_int64, C.c_uint64, C.POINTER(VP)]),
    "kabc_ais_create_batch": (C.c_int, [VP, C.POINTER(Model), C.c_int64, C.c_int32,
                                        C.POINTER(C.c_uint64), C.POINTER(VP)]),
    "kabc_ais_create_sharded": (C.c_int, [VP, C.POINTER(Model), C.c_int64, C.c_int32, C.c_int32,
                                          C.c_uint64, VP, VP, C.POINTER(VP)]),
    "kabc_ais_init": (C.c_int, [VP, C.c_int32]),
    "kabc_ais_half_generation": (C.c_int, [VP, C.c_int32, C.c_int32, VP]),
    "kabc_ais_end_generation": (C.c_int, [VP, C.c_int32]),
    "kabc_ais_advance": (C.c_int, [VP, C.c_int64, C.c_int32, c_double_p, C.POINTER(Stats)]),
    "kabc_ais_get_state": (C.c_int, [VP, c_double_p, c_double_p, c_double_p,
                                     C.POINTER(C.c_uint64)]),
    "kabc_ais_set_state": (C.c_int, [VP, c_double_p, c_double_p, c_double_p, C.c_uint64]),
    "kabc_ais_get_stats": (C.c_int, [VP, C.POINTER(Stats)]),
    "kabc_ais_owned": (C.c_int64, [VP, C.c_int32]),
    "kabc_ais_owned_segments": (C.c_int32, [VP, C.c_int32, C.POINTER(C.c_int64),
                                            C.POINTER(C.c_int64), C.c_int32]),
    "kabc_ais_set_timing": (C.c_int, [VP, C.c_int32]),
    "kabc_ais_set_timing_stride": (C.c_int, [VP, C.c_int32]),
    "kabc_ais_kernel_ms": (C.c_double, [VP, C.POINTER(C.c_int64)]),
    "kabc_ais_exchange_us": (C.c_int, [VP, c_double_p]),
    "kabc_ais_set_debug": (C.c_int, [VP, C.c_int32]),
    "kabc_ais_get_debug": (C.c_int, [VP, C.POINTER(C.c_int32), C.c_int64]),
    "kabc_ais_destroy": (C.c_int, [VP]),
    "kabc_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "kabc_comm_init_rank": (C.c_int, [VP, C.POINTER(C.c_uint8), C.c_int32, C.c_int32,
                                      C.POINTER(VP)]),
    "kabc_comm_init_all": (C.c_int, [C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(VP),
                                     C.POINTER(VP)]),
    "kabc_comm_rank": (C.c_int32, [VP]),
    "kabc_comm_world": (C.c_int32, [VP]),
    "kabc_comm_ctx": (VP, [VP]),
    "kabc_comm_allreduce_sum_u64": (C.c_int, [VP, C.POINTER(C.c_uint64), C.c_int32]),
    "kabc_comm_allreduce_max_f64": (C.c_int, [VP, c_double_p, C.c_int32]),
    "kabc_comm_barrier": (C.c_int, [VP]),
    "kabc_comm_destroy": (C.c_int, [VP]),
    "kabc_ais_create_dist": (C.c_int, [VP, C.POINTER(Model), C.c_int64, C.c_uint64,
                                       C.POINTER(VP)]),
    "kabc_ais_get_ensemble": (C.c_int, [VP, c_double_p]),
    "kabc_ais_init_multi": (C.c_int, [C.POINTER(VP), C.c_int32, C.c_int32]),
    "kabc_ais_advance_multi": (C.c_int, [C.POINTER(VP), C.c_int32, C.c_int64, C.c_int32,
                                         C.POINTER(Stats)]),
    "kabc_smc_default_opts": (None, [C.POINTER(SmcOpts)]),
    "kabc_abcde_default_opts": (None, [C.POINTER(AbcdeOpts)]),
    "kabc_abcde_run": (C.c_int, [VP, C.POINTER(Prior), C.c_int32, C.POINTER(Cost),
                                 C.POINTER(AbcdeOpts), C.POINTER(AbcdeResult)]),
    "kabc_pfilter_default_opts": (None, [C.POINTER(PfilterOpts)]),
    "kabc_pfilter_nparticles": (C.c_int64, [C.c_int64, C.c_double, C.c_int32]),
    "kabc_pfilter_run": (C.c_int, [VP, C.POINTER(Prior), C.c_int32, C.POINTER(Cost),
                                   C.POINTER(PfilterOpts), C.POINTER(PfilterResult)]),
    "kabc_smc_run": (C.c_int, [VP, C.POINTER(Prior), C.c_int32, C.POINTER(Cost),
                               C.POINTER(SmcOpts), C.POINTER(SmcResult)]),
    "kabc_smc_run_dist": (C.c_int, [VP, C.POINTER(Prior), C.c_int32, C.POINTER(Cost),
                               C.POINTER(SmcOpts), C.POINTER(SmcResult)]),
    "kabc_smc_run_dist_mode": (C.c_int, [VP, C.POINTER(Prior), C.c_int32, C.POINTER(Cost),
                                    C.POINTER(SmcOpts), C.c_int32, C.POINTER(SmcResult)]),
    "kabc_smc_dist_stats": (None, [C.POINTER(C.c_int64)]),
}


def bind(lib, prototypes=PROTOTYPES):
    """Attach restype/argtypes; raises AttributeError if a symbol is missing."""
    for name, (res, args) in prototypes.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
