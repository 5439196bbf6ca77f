O=gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_train.py -q -x -k "full_backward_16bit or full_batch" > $O/pt16.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pt16.log
for i in 1 2; do timeout -k 10 100 python tools/train_bench.py 1024 bf16 60 2>&1 | grep -v amdgpu.ids; BTSBOT_AMD_WGRAD_W4=1 timeout -k 10 100 python tools/train_bench.py 1024 bf16 60 2>&1 | grep -v amdgpu.ids; done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tp16 -- python3 tools/train_bench.py 1024 bf16 20 > $O/tp16.log 2>&1
python3 tools/kstats.py $O/tp16 25 8
