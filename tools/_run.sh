timeout 900 python -m pytest -q -m gpu tests/test_gpu_train.py -x 2>&1 | tail -2
timeout 200 python tools/train_steps.py --graph --steps 10 2>&1 | grep "graphed:\|Error\|error" | head -5
