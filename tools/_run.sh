mkdir -p gpurun_out
timeout 1100 python -m pytest -q -m gpu tests 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^ROCm\|^HIP version\|^Hostname\|^Librccl" | tail -15 > gpurun_out/r6_suite_i.log
tail -3 gpurun_out/r6_suite_i.log
