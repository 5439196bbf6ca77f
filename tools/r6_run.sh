O=gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_train.py -q -x -k "maxvit" > $O/pt43.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pt43.log
timeout -k 10 300 python tools/mv_train_bench.py 64 bf16 5 2>&1 | grep -v amdgpu
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mvt43 -- python3 tools/mv_train_bench.py 64 bf16 3 > $O/mvt43.log 2>&1
python3 tools/kstats.py $O/mvt43 4 16
