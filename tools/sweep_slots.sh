# pairs in flight against throughput for the three single-GPU workloads:  gpurun -- 'bash tools/sweep_slots.sh'
export UGSM_DEV=1  # the UGSM_* kernel-choice overrides below are development switches (ugsm_runtime.cpp, apply_dev_env)
for wl in full16mp 1080p fovea16mp; do for sl in 1 2 4 6 8 12; do echo -n "$wl slots=$sl: "; python bench.py --workload $wl --no-cpu-baseline --no-events --slots $sl 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'pairs/s')"; done; done
