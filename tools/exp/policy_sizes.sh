#!/bin/bash
# the two per-level kernel policies (latency / throughput) forced against the library's own choice, over frame sizes and context shapes:
# does the automatic choice pick the better one everywhere?   bash tools/exp/policy_sizes.sh out.txt
out=$1
: > $out
for size in "4928 3264 64" "3484 2307 96" "2464 1632 160" "1920 1080 256" "1280 720 384" "640 480 512"; do
  set -- $size
  for shape in "4 8" "4 1" "1 1"; do
    set -- $size $shape
    python tools/ab.py --size $1 $2 --pairs $3 --slots $4 --batch $5 --rounds 2 "library's choice:" "latency policy:UGSM_POLICY=l" "throughput policy:UGSM_POLICY=t" | grep -v "^round" >> $out || exit 1
  done
done
cat $out
