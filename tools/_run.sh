timeout 900 python -m pytest -q -m gpu tests/test_gpu_train_step.py tests/test_gpu_train_graph.py -x 2>&1 | grep "passed\|failed\|Error" | tail -3
for t in 0 1 0 1; do echo "FAN_OUT=$t"; GRL_FAN_OUT=$t timeout 200 python tools/train_steps.py --graph --steps 20 2>&1 | grep "graphed:"; done
