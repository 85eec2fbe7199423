cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5i
timeout 1500 python -m pytest tests/test_gpu_model.py -q -s 2>&1 | grep -E "max\||passed|failed|Error|error|assert" | tail -60 > gpurun_out/r5i/pytest_model.log
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -k "lr_schedule or graphed" 2>&1 | tail -8 > gpurun_out/r5i/pytest_train.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "split_precision" -s 2>&1 | grep -E "max|passed|failed" | tail -12 > gpurun_out/r5i/pytest_split.log
timeout 900 python bench.py --steps 10 --warmup 3 --no-train --no-cpu-baseline > gpurun_out/r5i/bench_quick.json 2> gpurun_out/r5i/bench_quick.err
cat gpurun_out/r5i/pytest_model.log gpurun_out/r5i/pytest_train.log gpurun_out/r5i/pytest_split.log; tail -c 2500 gpurun_out/r5i/bench_quick.json; tail -3 gpurun_out/r5i/bench_quick.err
