O=gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -x -k "nano or alternative or schedules or ragged" > $O/pt39.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pt39.log
timeout -k 10 120 python tools/nano_bench.py 1024 bf16 2>&1 | grep -v amdgpu
timeout -k 10 120 python tools/train_bench.py 1024 bf16 40 2>&1 | grep -v amdgpu | tail -3
BTSBOT_AMD_MV_NO_PART=1 timeout -k 10 200 python tools/mv_bench.py 1024 bf16 5 2>&1 | grep -v amdgpu
