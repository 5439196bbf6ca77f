O=gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 200 python tools/mv_bench.py 1024 f16 5 2>&1 | grep -v amdgpu
timeout -k 10 200 python tools/mv_bench.py 1024 bf16 5 2>&1 | grep -v amdgpu
timeout -k 10 200 python tools/stamps_maxvit.py 1024 2>&1 | grep -v amdgpu
BTSBOT_BENCH_REHEARSAL=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench2.log 2>&1; echo "bench2 rc=$?"; tail -c 700 $O/bench2.log
