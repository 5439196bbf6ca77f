#!/bin/bash
# A/B of one MaxViT switch: tools/mv_ab.sh ENVVAR  (runs the maxvit leg of bench.py with ENVVAR=0 and =1)
v=$1
for e in 0 1; do
  env $v=$e timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-steps 0 --maxvit-steps 5 > gpurun_out/ab_$e.log 2>&1
  grep "^{" gpurun_out/ab_$e.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())['maxvit']
print('$v=$e', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})
"
done
