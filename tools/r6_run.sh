O=gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "maxvit" > $O/pt21.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pt21.log
timeout -k 10 200 python tools/mv_bench.py 1024 bf16 5 2>&1 | grep -v amdgpu
BTSBOT_AMD_MV_NO_SMLP=1 timeout -k 10 200 python tools/mv_bench.py 1024 bf16 5 2>&1 | grep -v amdgpu
timeout -k 10 200 python tools/mv_bench.py 1024 bf16 5 2>&1 | grep -v amdgpu
BTSBOT_AMD_MV_NO_SMLP=1 timeout -k 10 200 python tools/mv_bench.py 1024 bf16 5 2>&1 | grep -v amdgpu
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mv21 -- python3 tools/mv_bench.py 1024 bf16 3 > $O/mv21.log 2>&1
python3 tools/kstats.py $O/mv21 4 14
