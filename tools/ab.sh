#!/bin/bash
# usage: tools/ab.sh VAR v1 v2 ...   -> bench.py value / ms_per_step for each setting of the environment variable
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', d['value'], d['ms_per_step'])"
done
