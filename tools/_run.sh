timeout 600 python -m pytest -q -m gpu tests/test_gpu_train.py -k "attention" 2>&1 | tail -2
for t in 0 1 0 1; do echo "PREFETCH=$t"; GRL_ATTN_BWD_PREFETCH=$t timeout 200 python tools/train_steps.py --graph --steps 10 2>&1 | grep "graphed:"; done
