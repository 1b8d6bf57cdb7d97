#!/bin/bash
# round 3, h = 16 after the two-to-a-CU main pass: config 4 bench lines, batch sizes, stress with the cost rule off (hard robots stay in the main pass:
# working sets beyond 67 rows are then re-factorised by the in-place tail of the 128-register kernel's rebuild)
set -o pipefail
OUT=gpurun_out/${1:-r03h16}
mkdir -p $OUT
timeout -k 10 600 python bench.py --mixed --horizon 16 > $OUT/bench_config4_f32.json 2> $OUT/bench_config4_f32.err; echo cfg4 $?
timeout -k 10 600 python bench.py --mixed --horizon 16 --hessian bf16x3 --no-side > $OUT/bench_config4_bf16x3.json 2> $OUT/bench_config4_bf16x3.err; echo cfg4bf16 $?
timeout -k 10 600 python bench.py --mixed --horizon 16 --robots 2048 --no-side --no-cpu-baseline > $OUT/bench_config4_2048_robots.json 2> $OUT/bench_config4_2048.err; echo cfg4-2048 $?
timeout -k 10 600 python bench.py --mixed --horizon 16 --robots 8192 --no-side --no-cpu-baseline > $OUT/bench_config4_8192_robots.json 2> $OUT/bench_config4_8192.err; echo cfg4-8192 $?
QRGPU_H16_TWO=0 timeout -k 10 600 python bench.py --mixed --horizon 16 --no-side --no-cpu-baseline > $OUT/bench_config4_one_per_cu.json 2> $OUT/bench_config4_one.err; echo cfg4-one $?
QRGPU_H16_BIG_US=100000 timeout -k 10 800 python scratch/stress_h16.py 21,22 > $OUT/stress_h16_cost_rule_off.txt 2>&1; tail -2 $OUT/stress_h16_cost_rule_off.txt
timeout -k 10 800 python scratch/stress_h16.py 21,22,23,31,32,33 > $OUT/stress_h16.txt 2>&1; tail -1 $OUT/stress_h16.txt
timeout -k 10 300 python scratch/diag_h16_two.py > $OUT/h16_two_per_cu.txt 2>&1; tail -4 $OUT/h16_two_per_cu.txt
