mkdir -p gpurun_out/prof
python bench.py > gpurun_out/prof/bench_line.json 2> gpurun_out/prof/bench_stderr.txt
tail -c 200 gpurun_out/prof/bench_line.json
