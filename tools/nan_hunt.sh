#!/bin/bash
# Round-5's order-dependent failure: the spawned-replica tests first, then the LR-schedule test, in ONE pytest process.
# usage: tools/nan_hunt.sh <runs> [env assignments...]   logs under gpurun_out/nan_hunt/
runs=${1:-4}; shift
out=gpurun_out/nan_hunt; mkdir -p $out
T=tests/test_gpu_train_replicas.py; G=tests/test_gpu_train_graph.py
ORDER="$T::test_graphed_step_data_parallel_two_replicas $T::test_ddp_two_replicas_on_the_gpu $G::test_graphed_train_step_follows_lr_schedule_and_resume"
tag=$(echo "$*" | tr ' =' '__'); tag=${tag:-plain}
for i in $(seq 1 $runs); do
  env "$@" timeout 300 python -m pytest -q -x -m gpu $ORDER > $out/${tag}_$i.log 2>&1; rc=$?
  echo "[$tag] run $i: rc $rc  $(grep -o 'non-finite weights after a replay.*' $out/${tag}_$i.log | head -1 | cut -c1-600)"
done
