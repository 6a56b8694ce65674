#!/bin/bash
# PMC counter passes (separate rocprofv3 runs, no tracing domains mixed in) over tools/kernel_suite.py.
# usage: tools/pmc_suite.sh <tag>   -> gpurun_out/pmc_<tag>_{sq,fetch,write}.csv
set -u
tag=$1
repo=$(pwd)
export TMPDIR=/tmp
cd /tmp
for pass in "sq:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  out=$repo/gpurun_out/pmc_${tag}_$name
  rm -rf $out; mkdir -p $out
  rocprofv3 --pmc $ctrs -f csv -d $out -o pmc -- python $repo/tools/kernel_suite.py --cols 93 > $out/stdout.txt 2>&1
  f=$(find $out -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 $repo/tools/pmc_summarize.py $f > $repo/gpurun_out/pmc_${tag}_$name.csv; head -20 $repo/gpurun_out/pmc_${tag}_$name.csv; else echo "no counter file for $name"; tail -5 $out/stdout.txt; fi
done
