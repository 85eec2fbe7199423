#!/bin/bash
# Produces the profile artefacts of a round on the GPU box (run from the repo root), all under gpurun_out/prof/:
#   kernel_stats.txt     rocprofv3 --kernel-trace --stats summary of the bench command (per-kernel calls / avg / share)
#   pmc_hbm.txt          FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes) over the per-kernel micro-benchmark
#   pmc_sq.txt           SQ / GRBM counters (MFMA-busy, VALU, wait buckets) for attention, QKV, block tail, CAB convs
#   bench_line.json      the default bench.py line;  bench_config2.json / bench_config4.json: BASELINE configs[1] / [3]
#   step_breakdown_configN.txt   tools/step_breakdown.py: share of a forward per C-ABI entry point, random-init and trained scales
set -u
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o r -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --no-other-scales --no-precision-legs --no-traffic > $OUT/prof_bench_stdout.txt 2>&1
DB=$(find /tmp/prof_kt -name "*.db" | head -1)
echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --no-other-scales --no-precision-legs --no-traffic" > $OUT/kernel_stats.txt
tail -1 $OUT/prof_bench_stdout.txt >> $OUT/kernel_stats.txt
python $ROOT/tools/rocprof_summary.py "$DB" 30 >> $OUT/kernel_stats.txt 2>&1
KERN=attn_window,attn_a2w,attn_w2a,qkv_anchor,block_tail_regs,cab_conv1,cab_conv2_regs,se,stage_conv
: > $OUT/pmc_hbm.txt
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/prof_$c
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $c -d $d -o r -- python $ROOT/tools/bench_kernels.py --tiles 4 --iters 2 --only $KERN > /dev/null 2>&1
  DBP=$(find $d -name "*.db" | head -1)
  echo "# --pmc $c -- python tools/bench_kernels.py --tiles 4 --iters 2 --only $KERN   (mean per dispatch; FETCH_SIZE / WRITE_SIZE count units of" >> $OUT/pmc_hbm.txt
  echo "#   the guide's HBM section: KiB, FETCH_SIZE x2 on gfx950)" >> $OUT/pmc_hbm.txt
  python $ROOT/tools/pmc_summary.py "$DBP" $c | grep -v "at6native\|rocclr" >> $OUT/pmc_hbm.txt 2>&1
done
cd $ROOT
tools/pmc_kernels.sh gpurun_out/prof/pmc_sq_all.txt --tiles 4 --iters 3 --only $KERN > /dev/null 2>&1
grep -v "at6native\|rocclr" gpurun_out/prof/pmc_sq_all.txt > $OUT/pmc_sq.txt; rm -f gpurun_out/prof/pmc_sq_all.txt
python tools/bench_kernels.py --tiles 4 2>&1 | grep -v amdgpu.ids > $OUT/bench_kernels.txt
python tools/bench_kernels.py --tiles 4 --logit-scale 100 --only attn_window,attn_a2w,attn_w2a,qkv_anchor,qkv_split 2>&1 | grep -v amdgpu.ids > $OUT/bench_kernels_scale100.txt
# per-entry-point share of one forward (single stream, HIP events): the three bench configurations at random-init and at
# checkpoint-like logit scales
for c in "8 3" "16 2" "4 4"; do
  set -- $c
  { python tools/step_breakdown.py $1 $2; python tools/step_breakdown.py $1 $2 --trained; } 2>&1 | grep -v amdgpu.ids > $OUT/step_breakdown_config$2.txt
done
python bench.py --config 2 --tiles 16 --no-cpu-baseline --no-traffic > $OUT/bench_config2.json 2> $OUT/bench_config2.err
python bench.py --config 4 --tiles 4 --no-cpu-baseline --no-traffic > $OUT/bench_config4.json 2> $OUT/bench_config4.err
python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt
tail -c 600 $OUT/bench_line.json
