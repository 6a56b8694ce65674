#!/bin/bash
# rocprofv3 kernel stats + SQ counters of the cfg2 NTT command (tools/cfg2_ntt.py); outputs under gpurun_out/prof_cfg2_<tag>*
set -u
tag=${1:-x}
quick=${2:-}          # "quick": kernel stats + the first SQ pass only (A/B builds selected through BOOJUM_HIP_LIB: tools/ntt_wait_ab.sh)
repo=$(pwd)
export TMPDIR=/tmp
cd /tmp
out=$repo/gpurun_out/prof_cfg2_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats -f csv -d $out/kt -o t -- python $repo/tools/cfg2_ntt.py --cfg2-only > $out/stdout_kt.txt 2>&1
f=$(find $out/kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $repo/gpurun_out/cfg2_${tag}_kernel_stats.csv && head -8 $f
for pass in "sq1:SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "sq2:SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" "tcc:FETCH_SIZE" "tccw:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  [ -n "$quick" ] && [ $name != sq1 ] && continue
  rocprofv3 --pmc $ctrs -f csv -d $out/$name -o p -- python $repo/tools/cfg2_ntt.py --cfg2-only > $out/stdout_$name.txt 2>&1
  f=$(find $out/$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 $repo/tools/pmc_summarize.py $f > $repo/gpurun_out/cfg2_${tag}_pmc_$name.csv
done
cd $repo
rm -rf $out
python3 tools/cfg2_summary.py $tag > gpurun_out/cfg2_${tag}_summary.json; cat gpurun_out/cfg2_${tag}_summary.json
