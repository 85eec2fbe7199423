timeout 900 python -m pytest -q -m gpu tests/test_gpu_train.py -x -k "planes" 2>&1 | tail -2
for i in 1 2; do timeout 200 python tools/train_steps.py --graph --steps 10 2>&1 | grep "graphed:"; done
