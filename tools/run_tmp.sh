cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_dyn_dim.py tests/test_gpu_user_priors.py tests/test_user_mvpriors.py tests/test_user_cost.py tests/test_gpu_smc_parity.py -x -q -m gpu 2>&1 | tail -5
KABC_SMC_DYN_TEAM=8 timeout 900 python -m pytest tests/test_gpu_dyn_dim.py tests/test_gpu_user_priors.py tests/test_user_mvpriors.py -x -q -m gpu -k smc 2>&1 | tail -3
KABC_SMC_DYN_TEAM=64 timeout 900 python -m pytest tests/test_gpu_dyn_dim.py tests/test_gpu_user_priors.py tests/test_user_mvpriors.py -x -q -m gpu -k smc 2>&1 | tail -3
