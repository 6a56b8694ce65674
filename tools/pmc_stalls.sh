#!/bin/bash
# Where the issue slots of the VALU-bound kernels go: one rocprofv3 --pmc pass (SQ counters only, no tracing domains) over
# tools/kernel_suite.py — Poseidon2 tree 2^23 x 93, LDE, iNTT, FRI fold.  SQ_WAVE_CYCLES ~ SQ_ACTIVE_INST_ANY + SQ_WAIT_INST_ANY +
# SQ_WAIT_ANY (quad-cycles summed over waves, MI355X_MICROARCH.md "rocprofv3 PMC slots"); SQ_ACTIVE_INST_VALU / _SCA split the
# active part, SQ_INSTS_SALU / SQ_INSTS_SMEM count the scalar side (the out-of-line branches and round-constant loads of p2_asm.inc).
# usage: tools/pmc_stalls.sh <tag>   ->  gpurun_out/pmc_<tag>_stalls.csv
set -u
tag=$1
repo=$(pwd)
export TMPDIR=/tmp
cd /tmp
out=$repo/gpurun_out/pmc_${tag}_stalls
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM \
  -f csv -d $out -o pmc -- python $repo/tools/kernel_suite.py --cols 93 > $out/stdout.txt 2>&1
f=$(find $out -name '*counter_collection.csv' | head -1)
if [ -n "$f" ]; then python3 $repo/tools/pmc_summarize.py $f > $repo/gpurun_out/pmc_${tag}_stalls.csv; cat $repo/gpurun_out/pmc_${tag}_stalls.csv; else echo "no counter file"; tail -20 $out/stdout.txt; fi
