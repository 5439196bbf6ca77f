#!/bin/bash
# Developer tool (GPU box): kernel-trace one command and print the per-kernel averages of the kernels matching a pattern.
# usage: tools/ktrace.sh <out-name> <grep-pattern> -- <program> <args...>     (the program itself after --, no env/bash hops)
set -u
name=$1; pat=$2; shift 3
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5/$name -- "$@" > gpurun_out/r5/$name.log 2>&1
cp gpurun_out/r5/$name/*/*kernel_stats.csv gpurun_out/r5/${name}_stats.csv 2>/dev/null
rm -rf gpurun_out/r5/$name
grep -E "$pat" gpurun_out/r5/${name}_stats.csv | cut -d, -f1-4 | cut -c1-160
