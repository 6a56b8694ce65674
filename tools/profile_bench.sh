#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of a bench.py invocation.
# usage: tools/profile_bench.sh <tag> [bench args...]; writes gpurun_out/prof_<tag>/ and gpurun_out/prof_<tag>_stats.csv
set -u
tag=$1; shift
repo=$(pwd)
export TMPDIR=/tmp
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $out -o trace -- python $repo/bench.py "$@" > $out/bench_stdout.txt 2>&1
cd $repo
f=$(find $out -name '*kernel_stats.csv' | head -1)
if [ -n "$f" ]; then cp $f gpurun_out/prof_${tag}_stats.csv; head -12 $f; else echo "no stats file"; find $out | head; fi
tail -1 $out/bench_stdout.txt | cut -c1-400
