timeout 900 python -m pytest -q -m gpu tests/test_gpu_train.py tests/test_gpu_train_graph.py tests/test_gpu_train_step.py tests/test_gpu_train_replicas.py -x 2>&1 | grep "passed\|failed" | tail -2
