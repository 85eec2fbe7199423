timeout 600 python -m pytest -q -m gpu tests/test_gpu_train.py -x 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^ROCm\|^HIP version\|^Hostname\|^Librccl" | tail -25
for rw in 1 0; do GRL_REAL_WIDTHS=$rw timeout 200 python tools/train_steps.py --graph --steps 10 2>&1 | grep "graphed:\|Error\|error" | head -5; done
