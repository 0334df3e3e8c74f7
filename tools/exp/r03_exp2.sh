#!/bin/bash
# round-3 experiment batch 2: where K-cost's waves run and when (census), strips by age class
set -u
O=gpurun_out/r03d; mkdir -p $O
export UGSM_DEV=1
step() { echo "[exp2] $* ($(date +%T))"; }
timeout -k 10 120 ./tools/kbench_stamp 4928 3264 10 6 > $O/census_uniform.txt 2>&1; step "census uniform rc=$?"
head -20 $O/census_uniform.txt
timeout -k 10 120 ./tools/kbench_stamp 4928 3264 10 6 430 326 > $O/census_430_326.txt 2>&1; step "census 430/326 rc=$?"
head -20 $O/census_430_326.txt
timeout -k 10 200 ./tools/kbench 4928 3264 10 12 > $O/kb12_16mp.txt 2>&1; step "kb12 16mp rc=$?"
grep -c bit-exact $O/kb12_16mp.txt; grep -v bit-exact $O/kb12_16mp.txt | tail -26
timeout -k 10 200 ./tools/kbench 3484 2307 10 12 > $O/kb12_8mp.txt 2>&1; step "kb12 8mp rc=$?"
grep -c bit-exact $O/kb12_8mp.txt; grep -v bit-exact $O/kb12_8mp.txt | tail -13
timeout -k 10 200 ./tools/kbench 2463 1631 20 12 > $O/kb12_4mp.txt 2>&1; step "kb12 4mp rc=$?"
grep -c bit-exact $O/kb12_4mp.txt; grep -v bit-exact $O/kb12_4mp.txt | tail -13
timeout -k 10 400 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; step "tests rc=$? $(tail -1 $O/gpu_tests.log)"
tail -15 $O/gpu_tests.log
