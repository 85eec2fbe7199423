#!/bin/bash
# Produces the profile artefacts of a round on the GPU box (run from the repo root):
#   gpurun_out/prof_stats.txt   rocprofv3 --kernel-trace --stats summary of the bench command
#   gpurun_out/prof_pmc.txt     FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes) for the attention launches
#   gpurun_out/bench_line.json  the default bench.py line
set -u
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=$ROOT/gpurun_out
cd /tmp
rm -rf /tmp/prof_kt /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_bench_stdout.txt 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline" > $OUT/prof_stats.txt
tail -1 $OUT/prof_bench_stdout.txt >> $OUT/prof_stats.txt
python $ROOT/tools/rocprof_summary.py "$DB" 30 >> $OUT/prof_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/prof_$c
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $c -d $d -o r -- python $ROOT/tools/bench_kernels.py --tiles 4 --iters 2 --only attn_window,attn_a2w,attn_w2a,qkv_stream,block_tail > /dev/null 2>&1
  DBP=$(find $d -name "*.db" | head -1)
  python $ROOT/tools/pmc_summary.py "$DBP" $c >> $OUT/prof_pmc.txt 2>&1
done
cd $ROOT
python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt
tail -c 1500 $OUT/bench_line.json
