O=gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "maxvit_forward_matches_oracle or maxvit_chunking" > $O/pt34.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pt33.log
timeout -k 10 200 python tools/mv_bench.py 1024 bf16 5 2>&1 | grep -v amdgpu
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mv34 -- python3 tools/mv_bench.py 1024 bf16 3 > $O/mv34.log 2>&1
python3 tools/kstats.py $O/mv34 4 12
