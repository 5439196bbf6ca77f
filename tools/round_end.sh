#!/bin/bash
# Round-end measurement recipe (run on the MI355X box through gpurun from the repo root).
# Writes everything under gpurun_out/final/; copy what should be judged into profiles/ (tools/collect_profiles.py).
# Every step has its own time limit and the chain stops at the first step that fails or is killed.
# usage: tools/round_end.sh [measure|profile|all]   (two gpurun calls of <= 20 minutes: measure, then profile)
set -u
O=gpurun_out/final; mkdir -p $O
export TMPDIR=/tmp
PHASE=${1:-all}
PROF="--pipeline-depth 1 --no-extra-legs"   # profiled runs: one stream, so a kernel's duration is its own
if [ "$PHASE" != profile ]; then
timeout -k 10 700 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1 &&
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 &&
( time timeout -k 10 400 python bench.py ) > $O/bench_default.log 2>&1 &&
timeout -k 10 120 python tools/nano_bench.py 1024 bf16 > $O/nano.log 2>&1 &&
timeout -k 10 120 python tools/stamps_nano.py > $O/stamps_nano.log 2>&1 &&
timeout -k 10 120 python tools/stamps.py 1024 > $O/stamps.log 2>&1 &&
timeout -k 10 120 python tools/stamps_train.py 1024 bf16 > $O/stamps_train.log 2>&1 &&
timeout -k 10 200 python tools/stamps_maxvit.py 1024 > $O/stamps_maxvit.log 2>&1 &&
timeout -k 10 120 python tools/train_bench.py 1024 bf16 40 > $O/train_s2_ab.log 2>&1 &&
BTSBOT_AMD_NO_S2P_TRAIN=1 timeout -k 10 120 python tools/train_bench.py 1024 bf16 40 >> $O/train_s2_ab.log 2>&1 &&
timeout -k 10 120 python tools/train_bench.py 1024 f32 10 > $O/train_f32_ab.log 2>&1 &&
BTSBOT_AMD_WGRAD_F32_OLD=1 timeout -k 10 120 python tools/train_bench.py 1024 f32 10 >> $O/train_f32_ab.log 2>&1 &&
timeout -k 10 120 python tools/train_bench.py 1024 bf16 40 > $O/train_ab.log 2>&1 &&
BTSBOT_AMD_NO_SIDE_STREAM=1 BTSBOT_AMD_NO_DWLN=1 timeout -k 10 120 python tools/train_bench.py 1024 bf16 40 >> $O/train_ab.log 2>&1
echo "measure rc=$?" > $O/chain_measure.log
tail -3 $O/pytest_gpu.log; head -1 $O/nano.log; cat $O/train_ab.log; tail -2 $O/smoke.log; tail -5 $O/bench_default.log | cut -c1-400; cat $O/chain_measure.log
fi
if [ "$PHASE" = measure ]; then exit 0; fi
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --train-steps 0 --maxvit-train-steps 0 $PROF > $O/trace.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 0 --maxvit-train-steps 0 $PROF > $O/pmc_fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 0 --maxvit-train-steps 0 $PROF > $O/pmc_write.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 0 --maxvit-steps 1 --maxvit-batch 256 --maxvit-train-steps 0 $PROF > $O/mfma.log 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-steps 20 --maxvit-steps 0 --maxvit-train-steps 0 $PROF > $O/train_trace.log 2>&1
rc=$?
if [ $rc = 0 ]; then
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_train -- python3 tools/train_bench.py 1024 bf16 5 > $O/pmc_fetch_train.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_train -- python3 tools/train_bench.py 1024 bf16 5 > $O/pmc_write_train.log 2>&1
rc=$?
fi
if [ $rc = 0 ]; then
# one training step's two-queue timeline, and the MaxViT forward's per-kernel statistics and HBM-side traffic
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/train_tl -- python3 tools/train_bench.py 1024 bf16 8 > $O/train_tl.log 2>&1 &&
python3 tools/train_timeline.py $O/train_tl > $O/train_timeline.txt 2>&1 &&
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mv_trace -- python3 tools/mv_bench.py 1024 bf16 3 > $O/mv_trace.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/mv_fetch -- python3 tools/mv_bench.py 1024 bf16 3 > $O/mv_fetch.log 2>&1 &&
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/mv_write -- python3 tools/mv_bench.py 1024 bf16 3 > $O/mv_write.log 2>&1 &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mvt_trace -- python3 tools/mv_train_bench.py 64 bf16 3 > $O/mvt_trace.log 2>&1
rc=$?
fi
echo "chain rc=$rc" > $O/chain.log
cat $O/chain.log; for f in trace pmc_fetch pmc_write mfma train_trace pmc_fetch_train pmc_write_train train_tl mv_trace mv_fetch mv_write mvt_trace; do tail -n 1 $O/$f.log | cut -c1-200; done
