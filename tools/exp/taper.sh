#!/bin/bash
# Experiment: how fast the calls shrink at the end of a region (UGSM_BENCH_TAPER: 0 = full-size calls to the end, the default;
# t > 0: a call takes at most remaining / (t x slots) pairs), same box.  gpurun -- 'bash tools/exp/taper.sh [workload ...]'
mkdir -p gpurun_out/taper
for wl in ${@:-full16mp}; do
for rep in 1 2; do
for steps in 20 96 384; do
for taper in 0 0.5 1; do
  UGSM_BENCH_TAPER=$taper timeout -k 10 120 python3 bench.py --workload $wl --steps $steps --warmup 5 --no-events --no-cpu-baseline > gpurun_out/taper/line.json 2>/dev/null || exit 1
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/taper/line.json')); print('%-10s steps %4d taper %-5s %8.2f  steady %s' % ('$wl', $steps, '$taper', d['value'], d['steady_state']['value']), flush=True)" | tee -a gpurun_out/taper/result_$wl.txt
done
done
done
done
