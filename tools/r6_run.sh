O=gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp
BTSBOT_AMD_NO_SIDE_STREAM=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tp20 -- python3 tools/train_bench.py 1024 bf16 20 > $O/tp20.log 2>&1
grep -v amdgpu $O/tp20.log | tail -1
python3 tools/kstats.py $O/tp20 25 40
