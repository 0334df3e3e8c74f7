#!/bin/bash
# 3-slot and 1-slot throughput with the latency kernels of the coarse levels on / off / one at a time
export UGSM_DEV=1  # the UGSM_* kernel-choice overrides below are development switches (ugsm_runtime.cpp, apply_dev_env)
run() { echo -n "$1: "; env $2 python bench.py --no-events $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))"; }
for rep in 1 2; do
run "slots3 off      " UGSM_SMALL_MAX_PIXELS=-1 ""
run "slots3 both     " UGSM_SMALL_MASK=3 ""
run "slots3 cost     " UGSM_SMALL_MASK=1 ""
run "slots3 smooth   " UGSM_SMALL_MASK=2 ""
run "slots3 both rh32" "UGSM_SMALL_MASK=3 UGSM_SMALL_RH=32" ""
run "slots3 both 40k " "UGSM_SMALL_MAX_PIXELS=40000" ""
done
run "slots1 both     " UGSM_SMALL_MASK=3 "--slots 1"
run "slots1 cost     " UGSM_SMALL_MASK=1 "--slots 1"
run "slots1 smooth   " UGSM_SMALL_MASK=2 "--slots 1"
run "slots1 both rh32" "UGSM_SMALL_MASK=3 UGSM_SMALL_RH=32" "--slots 1"
run "slots1 both 300k" "UGSM_SMALL_MAX_PIXELS=300000" "--slots 1"
