#!/bin/bash
# two ranks of tests/multi_gpu_worker.py on ONE GPU through gloo, with logs (debugging aid for the one-GPU box)
out=${1:-gpurun_out/multi}
mkdir -p $out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 WORLD_SIZE=2 BTSBOT_TEST_BACKEND=gloo BTSBOT_TEST_OUT=$out/res.json BTSBOT_TEST_WATCHDOG=100
RANK=0 LOCAL_RANK=0 timeout -k 10 150 python -u tests/multi_gpu_worker.py > $out/r0.log 2>&1 &
p0=$!
RANK=1 LOCAL_RANK=1 timeout -k 10 150 python -u tests/multi_gpu_worker.py > $out/r1.log 2>&1 &
p1=$!
wait $p0; c0=$?
wait $p1; c1=$?
echo "rank codes $c0 $c1"
tail -25 $out/r0.log
echo ---- ; tail -25 $out/r1.log
cat $out/res.json 2>/dev/null
