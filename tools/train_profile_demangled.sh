#!/bin/bash
# rocprofv3 kernel statistics of three eager training steps with torch's element-wise kernels demangled to their functors:
#   gpurun_out/prof/train_kernel_stats_demangled.txt
set -u
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out/prof; mkdir -p $OUT
cd /tmp; rm -rf /tmp/prof_trd
rocprofv3 --kernel-trace --stats -d /tmp/prof_trd -o r -- python $ROOT/tools/train_steps.py --steps 3 > /tmp/train_profd.log 2>&1
DB=$(find /tmp/prof_trd -name "*.db" | head -1)
echo "# rocprofv3 --kernel-trace --stats -- python tools/train_steps.py --steps 3   (calls / 3 = kernel nodes per step)" > $OUT/train_kernel_stats_demangled.txt
python $ROOT/tools/rocprof_summary.py "$DB" 60 --demangle >> $OUT/train_kernel_stats_demangled.txt 2>&1
cd $ROOT
