#!/bin/bash
# What do the parked cycles of ntt_local12 / ntt_strided8 wait for?  The cfg2 transform (2^20 x 256, coset 7) and the bench-size
# LDE timed on four builds of the library: the product, and the three A/B builds of csrc/ntt_r16.hip (no HBM loads, no HBM stores,
# no barriers — wrong results, only the durations mean something).  Build the variants first (CPU):
#   python -c "from era_boojum_amd import build; [build.build_variant('exp/libbj_%s.so' % t, ['-DBJ_R16_AB_' + t]) for t in ('NOLOAD','NOSTORE','NOBARRIER')]"
# usage (GPU box): tools/ntt_wait_ab.sh <tag>  ->  gpurun_out/ntt_wait_ab_<tag>.txt
set -u
tag=${1:-x}
out=gpurun_out/ntt_wait_ab_$tag.txt
: > $out
for v in PRODUCT NOLOAD NOSTORE NOBARRIER; do
  if [ $v = PRODUCT ]; then lib=""; else lib=$(pwd)/exp/libbj_$v.so; [ -f $lib ] || { echo "$v: variant not built" >> $out; continue; }; fi
  for i in 1 2; do echo "$v $(BOOJUM_HIP_LIB=$lib python tools/cfg2_ntt.py 2>/dev/null | tail -1)" >> $out; done
done
cat $out
# the same builds under rocprofv3: kernel durations (kernel trace) and SQ_BUSY_CYCLES / 32 = shader cycles of a launch give the
# clock the part sustains under each variant — whether the A/B gain is fewer cycles (a stall that went away) or a higher clock
for v in NOLOAD NOSTORE; do
  lib=$(pwd)/exp/libbj_$v.so; [ -f $lib ] || continue
  BOOJUM_HIP_LIB=$lib tools/prof_cfg2.sh ${tag}_$v quick > /dev/null 2>&1
  echo "== $v" >> $out; python3 -c "
import json
d=json.load(open('gpurun_out/cfg2_${tag}_${v}_summary.json'))
for k,e in d['kernels'].items(): print(k[:40], {f: e.get(f) for f in ('avg_ms','kernel_cycles','implied_clock_GHz','cycles_per_valu_instruction_per_simd','SQ_INSTS_VALU')})
" >> $out 2>&1
done
cat $out
