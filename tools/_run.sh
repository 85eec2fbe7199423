timeout 900 python -m pytest -q -m gpu tests/test_gpu_train.py -x -k "se_" 2>&1 | grep "passed\|failed\|Error" | tail -3
GRL_DETERMINISTIC=1 timeout 900 python -m pytest -q -m gpu tests/test_gpu_train.py -x -k "se_" 2>&1 | grep "passed\|failed\|Error" | tail -3
for i in 1 2; do timeout 200 python tools/train_steps.py --graph --steps 20 2>&1 | grep "graphed:"; done
