#!/bin/bash
# round 3, experiment 9: lanes (streams the slots are dealt onto) x slots
O=gpurun_out/exp9; mkdir -p $O
UGSM_DEV=1 UGSM_LANES=2 timeout -k 10 600 python -m pytest tests/test_gpu_r03.py tests/test_gpu_parity.py -x -q > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -2 $O/tests.txt
python tools/ab.py --slots 4 --pairs 64 --rounds 2 "lanes4:" "lanes4_idle:UGSM_IDLE_STREAMS=1" "lanes2:UGSM_LANES=2" "lanes1:UGSM_LANES=1" "lanes3:UGSM_LANES=3" | grep -v "^round" > $O/s4.txt 2>&1; cat $O/s4.txt
python tools/ab.py --slots 6 --pairs 60 --rounds 2 "lanes2:UGSM_LANES=2" "lanes3:UGSM_LANES=3" "lanes6:" | grep -v "^round" > $O/s6.txt 2>&1; cat $O/s6.txt
python tools/ab.py --slots 8 --pairs 64 --rounds 2 "lanes2:UGSM_LANES=2" "lanes4:UGSM_LANES=4" | grep -v "^round" > $O/s8.txt 2>&1; cat $O/s8.txt
python tools/ab.py --slots 3 --pairs 60 --rounds 2 "lanes3:" "lanes2:UGSM_LANES=2" "lanes1:UGSM_LANES=1" | grep -v "^round" > $O/s3.txt 2>&1; cat $O/s3.txt
python tools/ab.py --slots 2 --pairs 60 --rounds 2 "lanes2:" "lanes1:UGSM_LANES=1" | grep -v "^round" > $O/s2.txt 2>&1; cat $O/s2.txt
