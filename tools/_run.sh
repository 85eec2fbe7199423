timeout 900 python -m pytest -q -m gpu tests/test_gpu_train.py tests/test_gpu_train_step.py tests/test_gpu_train_graph.py tests/test_gpu_train_replicas.py -x 2>&1 | grep "passed\|failed\|Error" | tail -3
for t in 0 1 0 1; do echo "ZERO_ARENA=$t"; GRL_ZERO_ARENA=$t timeout 200 python tools/train_steps.py --graph --steps 20 2>&1 | grep "graphed:"; done
