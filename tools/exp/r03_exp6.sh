#!/bin/bash
# round 3, experiment 6: k_cost_march4 in the pipeline -- one-slot and four-slot contexts, ranges of levels
O=gpurun_out/exp6; mkdir -p $O
export UGSM_DEV=1
timeout -k 10 600 python -m pytest tests/test_gpu_march4.py -x -q > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -3 $O/tests.txt
for r in "0,0" "150001,3000000" "150001,5000000" "150001,1200000" "1,3000000"; do
  echo "== one slot UGSM_MARCH4=$r"; UGSM_MARCH4=$r timeout -k 10 200 python bench.py --slots 1 --no-events --no-cpu-baseline --steps 30 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done > $O/one_slot.txt 2>&1
cat $O/one_slot.txt
for r in "0,0" "150001,3000000" "200000,1200000" "200000,600000"; do
  echo "== four slots UGSM_MARCH4=$r"; UGSM_MARCH4=$r timeout -k 10 200 python bench.py --no-events --no-cpu-baseline --steps 48 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done > $O/four_slots.txt 2>&1
cat $O/four_slots.txt
for sz in "436 289" "308 204" "218 144" "154 102"; do timeout -k 10 60 ./tools/kbench $sz 200 14 | grep -v "rows=[1-9]"; done > $O/kb14_small.txt 2>&1
cat $O/kb14_small.txt
