timeout 900 python -m pytest -q -m gpu tests/test_gpu_train.py tests/test_gpu_train_step.py -x 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^ROCm\|^HIP version\|^Hostname\|^Librccl\|Warning\|warn" | tail -5
for t in 0 1 0 1; do echo "F16_HANDOVER=$t"; GRL_F16_HANDOVER=$t timeout 200 python tools/train_steps.py --graph --steps 10 2>&1 | grep "graphed:"; done
