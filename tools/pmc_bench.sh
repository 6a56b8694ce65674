#!/bin/bash
# PMC passes over the SAME command bench.py's headline comes from (separate rocprofv3 runs per counter group, no tracing
# domains mixed in).  usage: tools/pmc_bench.sh <tag> [bench args]  ->  gpurun_out/pmc_bench_<tag>.json (+ per-pass csv)
set -u
tag=$1; shift
repo=$(pwd)
export TMPDIR=/tmp
cd /tmp
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  out=$repo/gpurun_out/pmcb_${tag}_$name
  rm -rf $out; mkdir -p $out
  rocprofv3 --pmc $ctrs -f csv -d $out -o pmc -- python $repo/bench.py --no-ntt --no-cpu-baseline --no-scale-replay --no-two-in-flight "$@" > $out/stdout.txt 2>&1
  f=$(find $out -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then cp $f $repo/gpurun_out/pmcb_${tag}_$name.raw.csv; python3 $repo/tools/pmc_summarize.py $f > $repo/gpurun_out/pmcb_${tag}_$name.csv; else echo "no counter file for $name"; tail -5 $out/stdout.txt; fi
  rm -rf $out
done
cd $repo
python3 tools/pmc_leaf_traffic.py gpurun_out/pmcb_${tag}_fetch.raw.csv gpurun_out/pmcb_${tag}_write.raw.csv gpurun_out/pmcb_${tag}_sq.raw.csv > gpurun_out/pmc_bench_${tag}.json
cat gpurun_out/pmc_bench_${tag}.json
rm -f gpurun_out/pmcb_${tag}_*.raw.csv
