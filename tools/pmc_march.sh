# PMC passes over the marching K-cost alone (tools/kbench mode 3):  gpurun -- 'bash tools/pmc_march.sh <tag> <np> <rows>'
set -u
TAG=${1:-m}; NP=${2:-1}; ROWS=${3:-0}
O=$PWD/gpurun_out/pmc_$TAG
mkdir -p $O
K=$PWD/tools/kbench
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o $name --output-format csv -- $K 4928 3264 5 3 $NP $ROWS > $O/$name.log 2>&1; echo "$name done"; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run p2 SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_INSTS_SMEM
run p3 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64
run p4 GRBM_GUI_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_ACTIVE_INST_MISC
python3 - <<PY
import csv,glob,collections
for p in sorted(glob.glob('$O/p*/*counter_collection.csv')):
    acc=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(p)):
        if 'march' not in r['Kernel_Name']: continue
        acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
    print(p.split('/')[-2], {k: round(v/n[k]) for k,v in acc.items()})
for p in sorted(glob.glob('$O/p1/*kernel_trace.csv')):
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in csv.DictReader(open(p)) if 'march' in r['Kernel_Name']]
    print('durations us', [x/1000 for x in d])
PY
