#!/bin/bash
set -u
O=gpurun_out/r03g; mkdir -p $O
export UGSM_DEV=1
step() { echo "[exp5] $* ($(date +%T))"; }
timeout -k 10 500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1; step "tests rc=$? $(tail -1 $O/gpu_tests.log)"; tail -5 $O/gpu_tests.log
for sz in "154 102" "436 289" "54 36"; do timeout -k 10 100 ./tools/kbench $sz 20 13 > $O/kb13_${sz% *}.txt 2>&1; step "kb13 $sz"; cat $O/kb13_${sz% *}.txt; done
timeout -k 10 100 ./tools/kbench 1741 1153 20 > $O/kb_l3.txt 2>&1; grep "k_smooth_fused p5" $O/kb_l3.txt
timeout -k 10 100 ./tools/kbench_r02 1741 1153 20 > $O/kb_l3_r02.txt 2>&1; grep "k_smooth_fused p5" $O/kb_l3_r02.txt
timeout -k 10 100 ./tools/kbench 3484 2307 10 > $O/kb_l1.txt 2>&1; grep "k_smooth_fused p5" $O/kb_l1.txt
timeout -k 10 100 ./tools/kbench_r02 3484 2307 10 > $O/kb_l1_r02.txt 2>&1; grep "k_smooth_fused p5" $O/kb_l1_r02.txt
b() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-events ${BARGS:-} > $O/$name.json 2> $O/$name.err; step "$name: $(python -c "import json;d=json.load(open('$O/$name.json'));print(round(d['value'],1),'pairs/s',round(d['ms_per_step'],2),'ms')" 2>&1)"; }
BARGS="" b s4 UGSM_X=0
BARGS="--slots 1" b s1 UGSM_X=0
BARGS="--slots 1" b s1b UGSM_X=0
BARGS="--slots 2" b s2 UGSM_X=0
BARGS="--slots 2" b s2_two1 UGSM_TWO_STREAMS=1
BARGS="" b q2_s4 GPU_MAX_HW_QUEUES=2
