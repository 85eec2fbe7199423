timeout 300 python -m pytest -q -m gpu tests/test_gpu_train.py -k "layernorm or cpb" -s 2>&1 | grep -v "amdgpu.ids" | tail -8
for i in 1 2; do timeout 200 python tools/train_steps.py --graph --steps 10 2>&1 | grep "graphed:"; done
timeout 600 python -m pytest -q -m gpu tests/test_gpu_train_step.py tests/test_gpu_train_graph.py 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^ROCm\|^HIP version\|^Hostname\|^Librccl" | tail -4
timeout 300 python tools/grad_budget.py --runs 2 --variants hip 2>&1 | grep "run "
