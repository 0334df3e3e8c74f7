#!/bin/bash
# same-box A/B: round 4's harness (call planning in Python, slot-level API) against round 5's (the library's queue), the driver's protocol
# (--steps 20 --warmup 5) and a long region; bare throughput lines (--no-events), alternating.
set -e
out=${1:-gpurun_out/r05_ab_queue.txt}
: > $out
for rep in 1 2 3; do
  for h in tools/exp/bench_r04.py bench.py; do
    for steps in 20 384; do
      python $h --steps $steps --warmup 5 --no-events --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$h', 'steps', d['steps'], 'value %.2f' % d['value'], 'steady', d['steady_state']['value'], d.get('calls_formed_by_the_library'))" >> $out
    done
  done
done
cat $out
