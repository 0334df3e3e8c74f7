#!/bin/bash
# in-pipeline A/B of two builds of libugsm.so (compile-time switches): bench.py's throughput line and its one-pair-alone leg, alternating.
#   bash tools/exp/ab_lib.sh out.txt name:path/to/lib.so ...      ("base" = the library in the tree)
out=$1; shift
: > $out
cp ug_stereomatcher_amd/libugsm.so /tmp/ugsm_base.so
for rep in 1 2 3; do
  for v in base "$@"; do
    name=${v%%:*}; lib=${v#*:}
    [ "$name" = base ] && lib=/tmp/ugsm_base.so
    cp $lib ug_stereomatcher_amd/libugsm.so
    python bench.py --steps 384 --warmup 8 --repeats 0 --profile-pairs 0 --no-service --steady-steps 0 --single-pairs 24 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$name', 'pairs/s %.2f' % d['value'], 'steady %.2f' % d['steady_state']['value'], 'one pair alone %.2f pairs/s (%.3f ms)' % (d['single_pair_no_events']['pairs_per_s'], d['single_pair_no_events']['ms_per_pair_median']))" >> $out
  done
done
cp /tmp/ugsm_base.so ug_stereomatcher_amd/libugsm.so
cat $out
