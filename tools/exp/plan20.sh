#!/bin/bash
# Experiment: the driver's 20-step region under different call plans (UGSM_BENCH_PLAN), same box.  gpurun -- 'bash tools/exp/plan20.sh'
mkdir -p gpurun_out/plan20
for rep in 1 2 3; do
for plan in ${PLANS:-"" "5,5,5,5" "3,4,6,7" "2,4,6,8" "3,5,6,6" "4,5,5,6" "2,3,4,5,6" "3,4,5,6,2" "1,3,5,7,4" "4,4,4,4,4" "6,6,6,2" "2,6,6,6"}; do
  UGSM_BENCH_PLAN="$plan" timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 --no-events --no-cpu-baseline > gpurun_out/plan20/line.json 2>/dev/null || exit 1
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/plan20/line.json')); print('%-28s %7.2f' % ('$plan' or 'default', d['value']), flush=True)" | tee -a gpurun_out/plan20/result8.txt
done
done
