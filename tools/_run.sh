for x in 1 0; do echo "GRL_GEMM_TN_PIPE=$x"; GRL_GEMM_TN_PIPE=$x timeout 200 python tools/bench_gemm_tn.py 2>&1 | grep -v amdgpu.ids; done
timeout 300 python -m pytest -q -m gpu tests/test_gpu_train.py -k "gemm_tn or linear or conv3x3" 2>&1 | tail -2
for t in 0 1 0 1; do echo "PIPE=$t"; GRL_GEMM_TN_PIPE=$t timeout 200 python tools/train_steps.py --graph --steps 20 2>&1 | grep "graphed:"; done
