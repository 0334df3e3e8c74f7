#!/bin/bash
# K-smooth: shared binary64 reciprocal (tree) against the binary32 Newton quotients (-DSMOOTH_NEWTON=1), tools/kbench mode 20, alternating
out=$1
: > $out
for rep in 1 2; do
  for b in kbench kbench_newton; do
    echo "== $b" >> $out
    ./tools/$b 4928 3264 20 20 | grep -E "p5\+box|p5 +4928|P=[1-5] no box|P=5 \+ box" | grep -v "phase shift\|occ" >> $out
  done
done
cat $out
