O=gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp
echo "== bf16 x24"; timeout -k 10 200 python tools/det_debug.py bf16 160 24 2>&1 | grep -v amdgpu.ids | grep "pass" | grep -v "0.000e+00" | head -5
echo "== f16 x24"; timeout -k 10 200 python tools/det_debug.py f16 160 24 2>&1 | grep -v amdgpu.ids | grep "pass" | grep -v "0.000e+00" | head -5
echo "== bf16 B=163 (ragged) x12"; timeout -k 10 200 python tools/det_debug.py bf16 163 12 2>&1 | grep -v amdgpu.ids | grep "pass" | grep -v "0.000e+00" | head -5
timeout -k 10 700 python -m pytest tests/test_gpu_train.py -q -x > $O/pt14.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pt14.log
for i in 1 2; do timeout -k 10 100 python tools/train_bench.py 1024 bf16 60 2>&1 | grep -v amdgpu.ids; BTSBOT_AMD_NO_S2P_TRAIN=1 timeout -k 10 100 python tools/train_bench.py 1024 bf16 60 2>&1 | grep -v amdgpu.ids; done
