#!/bin/bash
# K-smooth A/B of kbench builds (tools/kbench mode 20, 16 MP, 112 x 36 tiles), alternating:  bash tools/exp/smooth_ab.sh out.txt kbench kbench_adj ...
out=$1; shift
: > $out
for rep in 1 2; do
  for b in "$@"; do
    echo "== $b" >> $out
    ./tools/$b 4928 3264 20 20 | grep -E "p5\+box|p5 +4928|P=[1-5] no box|P=5 \+ box" | grep -v "phase shift\|occ" >> $out
  done
done
cat $out
