#!/bin/bash
# Experiment: bench.py's 16 MP line with calls of up to B pairs, over the driver's 20 steps (and 384 with "long"), same box.
mkdir -p gpurun_out/bb
for rep in 1 2 3; do
for steps in ${STEPS:-20}; do
for b in ${BATCHES:-4 5 6 7 8}; do
  timeout -k 10 120 python3 bench.py --batch $b --steps $steps --warmup 5 --no-cpu-baseline --no-service --profile-pairs 0 --single-pairs 0 --repeats 1 --steady-steps 0 > gpurun_out/bb/line.json 2>/dev/null || exit 1
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/bb/line.json')); print('batch %d steps %4d  %7.2f %s' % ($b, $steps, d['value'], d['value_repeats']), flush=True)" | tee -a gpurun_out/bb/result2.txt
done
done
done
