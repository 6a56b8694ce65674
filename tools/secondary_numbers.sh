#!/bin/bash
# The secondary numbers of DESIGN.md §5 in one GPU call (-> gpurun_out/secondary_numbers_<tag>.txt, copied to
# profiles/r<NN>_secondary_numbers.txt by hand).  Every step runs under its own timeout so that a regression costs seconds of
# box time, not the call's limit.     usage: bash tools/secondary_numbers.sh r03
tag=${1:-x}
out=gpurun_out/secondary_numbers_$tag.txt
mkdir -p gpurun_out; : > $out
T="timeout 170"
for n in 20 23; do echo "== prove 2^$n (verified)" >> $out; $T python tools/prove_bench.py --log-n $n --reps 2 --verify 2>/dev/null | tail -n 1 >> $out; done
echo "== blake2s 2^22" >> $out; $T python tools/prove_bench.py --log-n 22 --transcript blake2s --reps 2 2>/dev/null | tail -n 1 >> $out
echo "== keccak256 2^22" >> $out; $T python tools/prove_bench.py --log-n 22 --transcript keccak256 --reps 2 2>/dev/null | tail -n 1 >> $out
echo "== recursion class 2^16" >> $out; $T python tools/recursion_class_bench.py 16 2>/dev/null | tail -n 2 >> $out
echo "== recursion class 2^20" >> $out; $T python tools/recursion_class_bench.py 20 2>/dev/null | tail -n 2 >> $out
echo "== recursion class 2^20, every evaluator as a captured op list" >> $out; $T python tools/recursion_class_bench.py 20 oplists 2>/dev/null | tail -n 2 >> $out
for tr in poseidon2 blake2s keccak256; do
  echo "== reference test, SHA-256 of 8 KiB, $tr" >> $out
  $T python tools/prove_bench.py --message-bytes 8192 --transcript $tr --reps 3 --verify 2>/dev/null | tail -n 1 >> $out
done
echo "== kernel suite 2^20 x 93 x 8" >> $out; $T python tools/kernel_suite.py 2>/dev/null | tail -n 4 >> $out
echo "== narrow tree after LDEs / back to back" >> $out; $T python tools/narrow_leaf_rate.py 2>/dev/null | tail -n 1 >> $out
cat $out
