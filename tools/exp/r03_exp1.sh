#!/bin/bash
# round-3 experiment batch 1: two-stream slots, LDS pad co-scheduling, r02 vs r03 K-smooth on the same box
set -u
O=gpurun_out/r03c; mkdir -p $O
export UGSM_DEV=1
step() { echo "[exp1] $* ($(date +%T))"; }
timeout -k 10 400 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; step "tests rc=$? $(tail -1 $O/gpu_tests.log)"
b() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-events ${BARGS:-} > $O/$name.json 2> $O/$name.err; step "$name: $(python -c "import json;d=json.load(open('$O/$name.json'));print(round(d['value'],1),'pairs/s',round(d['ms_per_step'],2),'ms')" 2>&1)"; }
BARGS="--slots 1" b s1_two0 UGSM_TWO_STREAMS=0
BARGS="--slots 1" b s1_two1 UGSM_TWO_STREAMS=1
BARGS="--slots 1" b s1_two0b UGSM_TWO_STREAMS=0
BARGS="--slots 1" b s1_two1b UGSM_TWO_STREAMS=1
BARGS="" b s4_two0 UGSM_TWO_STREAMS=0
BARGS="" b s4_two1 UGSM_TWO_STREAMS=1
BARGS="" b s4_pad4k UGSM_SMOOTH_LDS_PAD=4096
BARGS="" b s4_two0b UGSM_TWO_STREAMS=0
BARGS="" b s4_two1b UGSM_TWO_STREAMS=1
BARGS="" b s4_pad4kb UGSM_SMOOTH_LDS_PAD=4096
BARGS="--slots 3" b s3_two1 UGSM_TWO_STREAMS=1
BARGS="--slots 6" b s6_two1 UGSM_TWO_STREAMS=1
BARGS="--slots 6" b s6_pad4k UGSM_SMOOTH_LDS_PAD=4096
timeout -k 10 120 ./tools/kbench_r02 4928 3264 10 2>&1 | grep "smooth" > $O/kb_r02_16mp.txt; step "kbench r02"
timeout -k 10 120 ./tools/kbench 4928 3264 10 2>&1 | grep "smooth" > $O/kb_r03_16mp.txt; step "kbench r03"
timeout -k 10 120 ./tools/kbench_r02 3484 2307 10 2>&1 | grep "smooth" > $O/kb_r02_8mp.txt; step "kbench r02 8mp"
timeout -k 10 120 ./tools/kbench 3484 2307 10 2>&1 | grep "smooth" > $O/kb_r03_8mp.txt; step "kbench r03 8mp"
paste -d'|' $O/kb_r02_16mp.txt $O/kb_r03_16mp.txt | cut -c1-220
timeout -k 10 80 ./tools/valubench 2.0 mix > $O/valubench_mix.txt 2>&1; step "valubench mix rc=$?"
tail -12 $O/valubench_mix.txt
