#!/bin/bash
# round 3, experiment 10: slot streams by priority pool (own hardware queues), more slots, side streams in another pool
O=gpurun_out/exp10; mkdir -p $O
UGSM_DEV=1 timeout -k 10 600 python -m pytest tests/test_gpu_r03.py -x -q > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
python tools/ab.py --slots 4 --pairs 64 --rounds 2 "hhhh:" "nnnn:UGSM_STREAM_PRIO=nnnn" "llll:UGSM_STREAM_PRIO=llll" "hhll:UGSM_STREAM_PRIO=hhll" "side_l:UGSM_TWO_STREAMS=1;UGSM_SIDE_PRIO=l" "side_n:UGSM_TWO_STREAMS=1;UGSM_SIDE_PRIO=n" | grep -v "^round" > $O/s4.txt 2>&1; cat $O/s4.txt
python tools/ab.py --slots 5 --pairs 60 --rounds 2 "hhhhl:" "hhhll:UGSM_STREAM_PRIO=hhhll" | grep -v "^round" > $O/s5.txt 2>&1; cat $O/s5.txt
python tools/ab.py --slots 6 --pairs 60 --rounds 2 "hhhhll:" "hhhlll:UGSM_STREAM_PRIO=hhhlll" "hhnnll:UGSM_STREAM_PRIO=hhnnll" | grep -v "^round" > $O/s6.txt 2>&1; cat $O/s6.txt
python tools/ab.py --slots 8 --pairs 64 --rounds 2 "hhhhllll:" | grep -v "^round" > $O/s8.txt 2>&1; cat $O/s8.txt
python tools/ab.py --slots 1 --pairs 30 --rounds 2 "h_sideh:" "h_sidel:UGSM_SIDE_PRIO=l" "n_siden:UGSM_STREAM_PRIO=n" | grep -v "^round" > $O/s1.txt 2>&1; cat $O/s1.txt
