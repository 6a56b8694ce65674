#!/bin/bash
# Every profile a round commits under profiles/, taken on ONE binary (run on the GPU box through gpurun):
#   r<tag>_prover_2p22_kernel_stats.csv        rocprofv3 --kernel-trace --stats over the default bench command's timed region
#   r<tag>_prover_2p22_leaf_launches.csv       per-launch durations of the Poseidon2 leaf kernel from the same trace
#   r<tag>_pmc_bench_2p22_{fetch,write,sq}.csv + _leaf_traffic.json     separate --pmc passes (tools/pmc_bench.sh)
#   r<tag>_cfg2_ntt_*                          kernel stats + PMC passes of the cfg2 NTT (tools/prof_cfg2.sh)
# usage: tools/round_profiles.sh 03   ->  gpurun_out/profiles_r03/
set -u
tag=$1
repo=$(pwd)
dst=$repo/gpurun_out/profiles_r$tag
mkdir -p $dst
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_kt -o p -- python $repo/bench.py --no-cpu-baseline --no-host-witness --no-ntt --no-scale-replay --no-two-in-flight --steps 6 --warmup 1 > /tmp/prof_kt_stdout.txt 2>&1
f=$(find /tmp/prof_kt -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $dst/r${tag}_prover_2p22_kernel_stats.csv
t=$(find /tmp/prof_kt -name '*kernel_trace.csv' | head -1)
[ -n "$t" ] && python3 - "$t" > $dst/r${tag}_prover_2p22_leaf_launches.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "poseidon2_leaves_kernel" in r.get("Kernel_Name", "")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("launch,grid_size,duration_ms,VGPR_Count")
for i, r in enumerate(rows):
    print("%d,%s,%.4f,%s" % (i, r.get("Grid_Size", r.get("Grid_Size_X", "")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("VGPR_Count", r.get("Arch_VGPR_Count", ""))))
PY
tail -1 /tmp/prof_kt_stdout.txt | cut -c1-200
cd $repo
bash tools/pmc_bench.sh r$tag --no-host-witness --steps 3 > $dst/pmc_bench_stdout.txt 2>&1
for k in fetch write sq; do [ -f gpurun_out/pmcb_r${tag}_$k.csv ] && cp gpurun_out/pmcb_r${tag}_$k.csv $dst/r${tag}_pmc_bench_2p22_$k.csv; done
[ -f gpurun_out/pmc_bench_r$tag.json ] && cp gpurun_out/pmc_bench_r$tag.json $dst/r${tag}_pmc_bench_2p22_leaf_traffic.json
bash tools/prof_cfg2.sh r$tag > $dst/prof_cfg2_stdout.txt 2>&1
for f in gpurun_out/cfg2_r${tag}_*; do b=$(basename $f); cp $f $dst/r${tag}_cfg2_ntt_${b#cfg2_r${tag}_}; done
# one rank of the 8-GPU proof ALONE on this GPU, its peers replayed (era_boojum_amd/scale_replay.py): the kernel table of what a
# rank executes per proof at W = 8 (the recording pass — eight ranks sharing the GPU for one proof — is in the trace too: 20 replayed
# proofs against 1 recorded one per rank)
cd /tmp
rm -rf /tmp/prof_rp
rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_rp -o p -- python $repo/tools/replay_rank.py 8 0 20 > $dst/replay_w8_rank0_stdout.txt 2>&1
t=$(find /tmp/prof_rp -name '*kernel_trace.csv' | head -1)
[ -n "$t" ] && python3 $repo/tools/replay_table.py $t 20 > $dst/r${tag}_replay_w8_rank0_per_proof.csv
cd $repo
ls -la $dst
