O=gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 300 python tools/mv_train_bench.py 64 bf16 5 2>&1 | grep -v amdgpu
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mvt36 -- python3 tools/mv_train_bench.py 64 bf16 3 > $O/mvt36.log 2>&1
python3 tools/kstats.py $O/mvt36 4 45
