# PMC passes over the pyramid / seed / sqblur kernels (tools/kbench mode 9):  gpurun -- 'bash tools/pmc_aux.sh <tag>'
set -u
TAG=${1:-aux}
O=$PWD/gpurun_out/pmc_$TAG
mkdir -p $O
K=$PWD/tools/kbench
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/$name -o $name --output-format csv -- $K 4928 3264 2 9 > $O/$name.log 2>&1; echo "$name done"; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
run p2 SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM
run p3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH
python3 - <<PY
import csv,glob,collections
for p in sorted(glob.glob('$O/p*/*counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(p)):
        k=r['Kernel_Name'].split('(')[0][:40] + '|' + r.get('Grid_Size','')
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
    for k in acc: print(p.split('/')[-2], k, {c: round(v/n[k][c]) for c,v in acc[k].items()})
PY
