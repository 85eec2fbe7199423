timeout 600 python -m pytest -q -m gpu tests/test_gpu_train.py -k "planes" 2>&1 | tail -2
GRL_PLANES_ORDER=0 timeout 600 python -m pytest -q -m gpu tests/test_gpu_train.py -k "planes" 2>&1 | tail -2
for t in 0 1 0 1; do echo "PLANES_ORDER=$t"; GRL_PLANES_ORDER=$t timeout 200 python tools/train_steps.py --graph --steps 20 2>&1 | grep "graphed:"; done
