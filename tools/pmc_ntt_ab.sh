#!/bin/bash
# two PMC passes over tools/ntt_two_pass_ab.py: SQ stalls and TCC traffic
repo=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_a /tmp/pmc_b /tmp/pmc_c
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d /tmp/pmc_a -o pmc -- python $repo/tools/ntt_two_pass_ab.py 93 > /tmp/pmc_a.txt 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -f csv -d /tmp/pmc_b -o pmc -- python $repo/tools/ntt_two_pass_ab.py 93 > /tmp/pmc_b.txt 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -f csv -d /tmp/pmc_c -o pmc -- python $repo/tools/ntt_two_pass_ab.py 93 > /tmp/pmc_c.txt 2>&1
for d in a b c; do f=$(find /tmp/pmc_$d -name '*counter_collection.csv' | head -1); [ -n "$f" ] && python3 $repo/tools/pmc_summarize.py $f > $repo/gpurun_out/pmc_ab_$d.csv; done
cat $repo/gpurun_out/pmc_ab_a.csv $repo/gpurun_out/pmc_ab_b.csv $repo/gpurun_out/pmc_ab_c.csv | cut -c1-250
