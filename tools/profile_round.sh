#!/bin/bash
# Collects everything profiles/make_summaries.py needs, in ONE gpurun call:
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r06 A'   (then B)
#   python profiles/make_summaries.py gpurun_out/prof_r06 r06
# Counter passes are separate rocprofv3 runs with at most 8 counters each (--kernel-trace only beside --pmc), each
# behind its own timeout; a line is printed after every step so that the call never looks hung.
export UGSM_DEV=1  # the UGSM_* kernel-choice overrides below are development switches (ugsm_runtime.cpp, apply_dev_env)
set -u
TAG=${1:-r06}
PART=${2:-ABC}   # A: the bench lines, the kernel trace, the counter passes; B: kbench, breakdowns, rehearsals, valubench (VALUBENCH=1); C: the same-box A/Bs of
                 # the batch dimension (gpurun calls of <= 20 minutes: one part per call; tools/ab_alone.sh is two more).  Order for a round's final set: B, A,
                 # then `python profiles/make_summaries.py gpurun_out/prof_rNN rNN` here, then `python bench.py` once more on the GPU (its valu_roofline then
                 # prices the kernels as built: stale == false)
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
step() { echo "[profile_round] $* ($(date +%T))"; }
if [[ $PART == *A* ]]; then
python bench.py > $O/bench_default.json 2> $O/bench_default.err; step "bench default: $(cut -c1-160 $O/bench_default.json)"
# the driver's protocol (BENCH_rNN.json) and round 3's configuration (single-pair calls), same box
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-service --profile-pairs 0 --single-pairs 0 > $O/bench_steps20.json 2>> $O/bench_default.err; step "bench --steps 20 --warmup 5: $(cut -c1-120 $O/bench_steps20.json)"
python bench.py --batch 1 --no-cpu-baseline --no-service --profile-pairs 0 --single-pairs 0 > $O/bench_batch1.json 2>> $O/bench_default.err; step "bench --batch 1: $(cut -c1-120 $O/bench_batch1.json)"
python bench.py --slots 1 --batch 1 --no-cpu-baseline --no-service > $O/bench_slots1.json 2>> $O/bench_default.err; step "bench slots1"
python bench.py --workload 1080p --no-cpu-baseline --no-service > $O/bench_1080p.json 2>> $O/bench_default.err; step "bench 1080p: $(cut -c1-120 $O/bench_1080p.json)"
python bench.py --workload fovea16mp --no-cpu-baseline --no-service > $O/bench_fovea16mp.json 2>> $O/bench_default.err; step "bench fovea16mp: $(cut -c1-120 $O/bench_fovea16mp.json)"
cd /tmp && export TMPDIR=/tmp
# the SAME command as the default bench line (minus the CPU and service legs, which launch no kernels of ours; 96 steps instead of
# 384: the kernel trace of the longer region does not fit gpurun's 64 MiB of returned files; without the one-slot leg, whose latency-policy
# launches -- k_cost_march at levels 0-2 only -- are not the timed region's and would weigh on the kernel's mean)
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/trace -o runc --output-format csv -- python3 $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-service --single-pairs 0 > $O/bench_under_rocprof.json 2> $O/trace.err; step "kernel trace"
# (UGSM_ALONE=0: a lone call would otherwise make the choices of a call alone on the chip, not the ones of the bench line's calls)
# The HBM-traffic passes run the timed region's launches (full-size calls of eight pairs, nothing else: `roofline.traffic` is per launch, like `roofline.achieved`); the
# SQ passes run single-pair launches (PMC_BATCH=1), whose grids identify the level (`valu_insts_level0`: the largest grid).
pmc() { name=$1; shift; b=${PMC_BATCH:-8}; UGSM_ALONE=0 timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o runc --output-format csv -- python3 $R/bench.py --steps $((b > 4 ? b : 4)) --warmup $((b > 1 ? 0 : 1)) --slots 1 --batch $b --no-cpu-baseline --no-events > $O/$name.json 2> $O/$name.err; step "pmc $name"; }
pmc pmc_rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum
pmc pmc_write WRITE_SIZE
pmc pmc_fetch FETCH_SIZE
PMC_BATCH=1
pmc pmc_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS
pmc pmc_sq2 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU
pmc pmc_sq3 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64
cd $R
fi
if [[ $PART == *B* ]]; then
timeout -k 10 200 ./tools/kbench 4928 3264 10 2 > $O/kbench_16mp.txt 2>&1; step "kbench (marching K-cost against libugsm_dev.so's LDS-tiled one)"
timeout -k 10 200 ./tools/kbench 4928 3264 10 5 > $O/kbench_smooth_16mp.txt 2>&1; step "kbench (K-smooth over its passes)"
{ for sz in "54 36" "154 102" "436 289" "616 408"; do timeout -k 10 100 ./tools/kbench $sz 200 7 | grep -v "P=[0-4]"; done; } > $O/kbench_small.txt 2>&1; step "kbench (latency kernels of the coarse levels)"
timeout -k 10 100 ./tools/kbench 4928 3264 20 9 > $O/kbench_aux_16mp.txt 2>&1; step "kbench (pyramid base, blur+decimate, seed, sqblur)"
{ for sz in "2464 1632" "1742 1154" "1232 816" "871 577"; do timeout -k 10 100 ./tools/kbench $sz 50 10; done; } > $O/kbench_strips.txt 2>&1; step "kbench (strip heights of the marching K-cost)"
{ for sz in "4928 3264 10" "3484 2307 10" "2463 1631 20"; do timeout -k 10 100 ./tools/kbench $sz 12 | grep -v "bit-exact" ; done; } > $O/kbench_age_16mp.txt 2>&1; step "kbench (strips by age class)"
{ for sz in "54 36" "154 102" "436 289"; do timeout -k 10 60 ./tools/kbench $sz 20 13; done; } > $O/kbench_graph.txt 2>&1; step "kbench (eager launches against a HIP graph, coarse level)"
{ for sz in "436 289" "615 407" "870 576" "1231 815" "1741 1153" "2463 1631"; do echo "== $sz"; timeout -k 10 60 ./tools/kbench $sz 100 14 | grep -v "rows=[1-9]"; done; } > $O/kbench_march4.txt 2>&1; step "kbench (k_cost_march4 against k_cost_march / split / small)"
{ for sz in "3 1" "4 1" "8 1" "4 0" "4 1 hhhh" "8 1 hhhhllll"; do timeout -k 5 60 ./tools/queue_probe $sz; done; } > $O/queue_probe.txt 2>&1; step "queue probe"
{ timeout -k 10 100 python tools/level_breakdown.py; timeout -k 10 100 python tools/level_breakdown.py --batch 4 --slots 4; timeout -k 10 100 python tools/level_breakdown.py --fovea 7; timeout -k 10 100 python tools/level_breakdown.py --fovea 7 --batch 8 --slots 4; timeout -k 10 100 python tools/level_breakdown.py --size 1920 1080 --batch 8 --slots 4; } 2>&1 | grep -v amdgpu.ids > $O/level_breakdown.txt; step "per-level breakdown: one pair, a batch of 4, the foveated stack alone and as a batch of 8, 1080p as a batch of 8"
timeout -k 10 100 ./tools/kbench 4928 3264 10 18 > $O/kbench_two_streams.txt 2>&1; step "kbench (two kernels on two streams)"
{ echo "# two gloo ranks on one MI355X (rehearsal of the N > 1 code path of the replicas: both ranks share the card, each with its own four-slot context and its own queue)"; echo "\$ UGSM_BENCH_DEVICE=0 UGSM_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 24 --warmup 4 --workload full16mp --no-cpu-baseline --no-service --profile-pairs 0 --repeats 0 --single-pairs 0"; UGSM_BENCH_DEVICE=0 UGSM_DIST_BACKEND=gloo MASTER_PORT=29611 timeout -k 10 200 python bench.py --gpus 2 --steps 24 --warmup 4 --workload full16mp --no-cpu-baseline --no-service --profile-pairs 0 --repeats 0 --single-pairs 0 2>/dev/null | cut -c1-600; echo "# the fovea shard on ONE rank: torch.distributed (nccl) for the barrier, the library's own RCCL communicator for the exchange (two ranks cannot share a GPU under RCCL: no two-rank rehearsal of this workload on a one-GPU box)"; echo "\$ UGSM_FORCE_DIST=1 python bench.py --workload fovea-shard --steps 48 --warmup 4 --no-cpu-baseline"; UGSM_FORCE_DIST=1 MASTER_PORT=29612 timeout -k 10 200 python bench.py --workload fovea-shard --steps 48 --warmup 4 --no-cpu-baseline 2>/dev/null | cut -c1-900; } > $O/rehearsal_2ranks.txt 2>&1; step "two-rank rehearsal over gloo + one-rank fovea shard over RCCL"
timeout -k 10 300 python -m pytest tests/test_gpu_dist.py tests/test_gpu_batch.py::test_two_contexts_in_one_process_with_stream_priority_pools -m gpu -q -s 2>&1 | grep -v amdgpu.ids | tail -12 > $O/rccl_and_contexts.txt; step "one-rank RCCL shard test (ugsm_shard_*) + two contexts"
# (tools/ab_alone.sh -- a call alone on one-slot and four-slot contexts, the shared choices forced, side-stream pools -- takes 25 minutes: two gpurun calls of its own, parts A and CB)
# (7 minutes; the instruction costs do not change with the kernels: only with VALUBENCH=1)
if [ "${VALUBENCH:-0}" = 1 ]; then timeout -k 10 420 ./tools/valubench > $O/valubench.txt 2>&1; step "valubench"; fi
timeout -k 10 200 python tools/service_latency.py > $O/service_latency.txt 2>&1; step "service latency"
# the default line once more when part A's counter passes have been condensed (profiles/make_summaries.py) into this round's VALU model: `valu_roofline` then prices the kernels as built
if [ -f profiles/${TAG}_valu_model.json ]; then python bench.py > $O/bench_default.json 2> $O/bench_default.err; step "bench default (with ${TAG}_valu_model.json): $(cut -c1-160 $O/bench_default.json)"; fi
fi
if [[ $PART == *C* ]]; then
{ timeout -k 10 400 python tools/ab.py --slots 4 --pairs 64 --rounds 2 "single-pair calls:" "batches of 2:BATCH=2" "batches of 4:BATCH=4" "batches of 8:BATCH=8" "batches of 4, levels <= 2.2 Mpx batched:BATCH=4;UGSM_BATCH_MAX_PIXELS=2200000" "batches of 4, every level batched:BATCH=4;UGSM_BATCH_MAX_PIXELS=20000000" "batches of 4, tiled pyramid kernel:BATCH=4;UGSM_PYR_STREAM=0" "two slots x 4:SLOTS=2;BATCH=4" | grep -v "^round"; timeout -k 10 300 python tools/ab.py --fovea 7 --slots 4 --pairs 256 --rounds 2 "single-pair calls:" "batches of 2:BATCH=2" "batches of 4:BATCH=4" "batches of 8:BATCH=8" "batches of 16:BATCH=16" "two slots x 8:SLOTS=2;BATCH=8" "batches of 8, the choices of a call alone:BATCH=8;UGSM_ALONE=1" "batches of 8, tiled pyramid kernel:BATCH=8;UGSM_PYR_STREAM=0" | grep -v "^round"; timeout -k 10 300 python tools/ab.py --size 1920 1080 --slots 4 --pairs 256 --rounds 2 "single-pair calls:" "batches of 4:BATCH=4" "batches of 8:BATCH=8" "batches of 16:BATCH=16" "two slots x 16:SLOTS=2;BATCH=16" | grep -v "^round"; timeout -k 10 200 python tools/ab.py --slots 1 --pairs 30 --rounds 2 "default:" "no k_cost_march4:UGSM_MARCH4=0,0" "tiled pyramid kernel:UGSM_PYR_STREAM=0" | grep -v "^round"; } > $O/ab_batch.txt 2>&1; step "same-box A/B: batch sizes, batch threshold, pyramid kernel"
fi
ls $O
