#!/bin/bash
# the secondary numbers of DESIGN.md §5 in one GPU call (everything lands in gpurun_out/r2_numbers_<tag>.txt); every step runs
# under its own timeout so that a regression costs seconds of box time, not the call's limit
tag=${1:-b}
out=gpurun_out/r2_numbers_$tag.txt
mkdir -p gpurun_out; : > $out
T="timeout 150"
for n in 20 23; do echo "== prove 2^$n (verified)" >> $out; $T python tools/prove_bench.py --log-n $n --reps 2 --verify 2>/dev/null | tail -1 >> $out; done
echo "== blake2s 2^22" >> $out; $T python tools/prove_bench.py --log-n 22 --transcript blake2s --reps 2 2>/dev/null | tail -1 >> $out
echo "== keccak256 2^22" >> $out; $T python tools/prove_bench.py --log-n 22 --transcript keccak256 --reps 2 2>/dev/null | tail -1 >> $out
echo "== recursion class 2^16" >> $out; $T python tools/recursion_class_bench.py 16 2>/dev/null | tail -2 >> $out
echo "== recursion class 2^20" >> $out; $T python tools/recursion_class_bench.py 20 2>/dev/null | tail -2 >> $out
cat $out
