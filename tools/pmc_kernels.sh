#!/bin/bash
# SQ / GRBM counter passes over the per-kernel micro-benchmark (tools/bench_kernels.py): MFMA-busy, VALU, wait buckets.
# usage (GPU box, repo root): tools/pmc_kernels.sh <out.txt> [bench_kernels args...]      counters in separate passes (8 SQ slots)
set -u
ROOT=$(pwd)
export TMPDIR=/tmp
OUT=$ROOT/$1; shift
ARGS="$@"
: > $OUT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VALU_TRANS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  d=/tmp/pmc_pass_$i
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $set -d $d -o r -- python $ROOT/tools/bench_kernels.py $ARGS > /tmp/pmc_pass_$i.log 2>&1
  DBP=$(find $d -name "*.db" | head -1)
  echo "# pass $i: --pmc $set  -- python tools/bench_kernels.py $ARGS" >> $OUT
  if [ -n "$DBP" ]; then python $ROOT/tools/pmc_summary.py "$DBP" >> $OUT 2>&1; else tail -5 /tmp/pmc_pass_$i.log >> $OUT; fi
done
cd $ROOT
