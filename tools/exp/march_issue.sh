#!/bin/bash
# K-cost issue-slot study: every kbench_* variant listed, alternating, three rounds per process, two passes.
#   bash tools/exp/march_issue.sh out.txt base:tools/kbench ilv1:tools/kbench_ilv1 ...
out=$1; shift
: > $out
for pass in 1 2; do
  for sz in "4928 3264 100" "3484 2307 150" "1742 1154 300"; do
    for v in "$@"; do
      bin=${v#*:}
      timeout -k 10 120 $bin $sz 19 >> $out 2>&1 || echo "FAILED: $bin $sz" >> $out
    done
  done
done
cat $out
