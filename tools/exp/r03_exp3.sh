#!/bin/bash
set -u
O=gpurun_out/r03e; mkdir -p $O
export UGSM_DEV=1
step() { echo "[exp3] $* ($(date +%T))"; }
for sz in "4928 3264 10" "3484 2307 10" "2463 1631 20"; do set -- $sz; timeout -k 10 200 ./tools/kbench $1 $2 $3 12 > $O/kb12_$1.txt 2>&1; step "kb12 $1 rc=$? bitexact=$(grep -c bit-exact $O/kb12_$1.txt)"; grep -v bit-exact $O/kb12_$1.txt | tail -26; done
timeout -k 10 120 ./tools/kbench_stamp 4928 3264 10 6 470 340 > $O/census_470_340.txt 2>&1; step "census 470/340"; sed -n 1,3p $O/census_470_340.txt; grep "start rank\|stamped" $O/census_470_340.txt
b() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-events ${BARGS:-} > $O/$name.json 2> $O/$name.err; step "$name: $(python -c "import json;d=json.load(open('$O/$name.json'));print(round(d['value'],1),'pairs/s',round(d['ms_per_step'],2),'ms')" 2>&1)"; }
BARGS="" b s4_uniform UGSM_X=0
BARGS="" b s4_age450 UGSM_MARCH_AGE=450,340
BARGS="" b s4_age470 UGSM_MARCH_AGE=470,340
BARGS="" b s4_uniformb UGSM_X=0
BARGS="" b s4_age450b UGSM_MARCH_AGE=450,340
BARGS="" b s4_age470b UGSM_MARCH_AGE=470,340
BARGS="--slots 1" b s1_uniform UGSM_X=0
BARGS="--slots 1" b s1_age450 UGSM_MARCH_AGE=450,340
timeout -k 10 400 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; step "tests rc=$? $(tail -1 $O/gpu_tests.log)"
