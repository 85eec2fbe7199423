python bench.py > gpurun_out/prof/bench_line.json 2> gpurun_out/prof/bench_stderr.txt
bash tools/train_profile.sh > /dev/null 2>&1
bash tools/train_profile_demangled.sh > /dev/null 2>&1
timeout 300 python tools/train_glue_sites.py 8 60 2>&1 | grep -v "amdgpu.ids" > gpurun_out/prof/train_glue_sites.txt
for i in 1 2; do timeout 200 python tools/train_steps.py --graph --steps 20 2>&1 | grep "graphed:"; done
tail -c 300 gpurun_out/prof/bench_line.json
