#!/bin/bash
# round 3, experiment 8: K-smooth tile heights in the pipeline
O=gpurun_out/exp8; mkdir -p $O
export UGSM_DEV=1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -3 $O/tests.txt
for r in 36 0 -2 39; do
  echo "== one slot UGSM_SMOOTH_ROWS=$r"; UGSM_SMOOTH_ROWS=$r timeout -k 10 200 python bench.py --slots 1 --no-events --no-cpu-baseline --steps 30 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done > $O/one_slot.txt 2>&1
grep -v "amdgpu.ids\|^\[bench" $O/one_slot.txt
for r in 36 0 -1 38 36 0; do
  echo "== four slots UGSM_SMOOTH_ROWS=$r"; UGSM_SMOOTH_ROWS=$r timeout -k 10 200 python bench.py --no-events --no-cpu-baseline --steps 48 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done > $O/four_slots.txt 2>&1
grep -v "amdgpu.ids\|^\[bench" $O/four_slots.txt
