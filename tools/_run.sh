timeout 300 python -m pytest -q -m gpu tests/test_gpu_train.py -k "pack_conv or conv3x3_fn" 2>&1 | grep -v "amdgpu.ids" | tail -4
for i in 1 2; do timeout 200 python tools/train_steps.py --graph --steps 10 2>&1 | grep "graphed:"; done
timeout 600 python -m pytest -q -m gpu tests/test_gpu_train_step.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^ROCm\|^HIP version\|^Hostname\|^Librccl" | tail -3
