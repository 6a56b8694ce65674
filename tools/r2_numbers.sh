#!/bin/bash
# the secondary numbers of DESIGN.md §5 in one GPU call (everything lands in gpurun_out/r2_numbers.txt)
out=gpurun_out/r2_numbers.txt
: > $out
for n in 14 16 18 20 23; do echo "== prove 2^$n" >> $out; python tools/prove_bench.py --log-n $n --reps 3 2>/dev/null | tail -1 >> $out; done
for t in poseidon2 blake2s keccak256; do echo "== 8 KiB SHA-256, $t" >> $out; python tools/prove_bench.py --message-bytes 8192 --transcript $t --verify 2>/dev/null | tail -1 >> $out; done
echo "== blake2s 2^22" >> $out; python tools/prove_bench.py --log-n 22 --transcript blake2s --reps 2 2>/dev/null | tail -1 >> $out
echo "== kernel suite 2^20" >> $out; python tools/kernel_suite.py --log-n 20 2>/dev/null >> $out
echo "== cfg2" >> $out; python tools/cfg2_ntt.py 2>/dev/null >> $out
echo "== recursion class 2^16" >> $out; python tools/recursion_class_bench.py 16 2>/dev/null | tail -2 >> $out
echo "== recursion class 2^20" >> $out; python tools/recursion_class_bench.py 20 2>/dev/null | tail -2 >> $out
cat $out
