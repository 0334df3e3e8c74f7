#!/bin/bash
set -u
O=gpurun_out/r03f; mkdir -p $O
export UGSM_DEV=1
step() { echo "[exp4] $* ($(date +%T))"; }
b() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-events ${BARGS:-} > $O/$name.json 2> $O/$name.err; step "$name: $(python -c "import json;d=json.load(open('$O/$name.json'));print(round(d['value'],1),'pairs/s',round(d['ms_per_step'],2),'ms')" 2>&1)"; }
BARGS="" b q4_s4 UGSM_X=0
BARGS="" b q8_s4 GPU_MAX_HW_QUEUES=8
BARGS="--slots 6" b q8_s6 GPU_MAX_HW_QUEUES=8
BARGS="--slots 8" b q8_s8 GPU_MAX_HW_QUEUES=8
BARGS="--slots 5" b q8_s5 GPU_MAX_HW_QUEUES=8
BARGS="" b q8_s4_two1 GPU_MAX_HW_QUEUES=8 UGSM_TWO_STREAMS=1
BARGS="--slots 3" b q8_s3_two1 GPU_MAX_HW_QUEUES=8 UGSM_TWO_STREAMS=1
BARGS="--slots 6" b q16_s6 GPU_MAX_HW_QUEUES=16
BARGS="--slots 8" b q16_s8 GPU_MAX_HW_QUEUES=16
BARGS="--slots 12" b q16_s12 GPU_MAX_HW_QUEUES=16
BARGS="--slots 6" b q8_s6_age GPU_MAX_HW_QUEUES=8 UGSM_MARCH_AGE=470,340
BARGS="" b q4_s4b UGSM_X=0
BARGS="--slots 6" b q8_s6b GPU_MAX_HW_QUEUES=8
