#!/bin/bash
# PMC counters of one 16-bit pointwise shape / tile shape:  bash tools/pwh_pmc.sh <shape index> <cfg>   (tools/pwh_probe.py)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; cd "$R" || exit 1
O=gpurun_out/pwh_pmc; rm -rf $O; mkdir -p $O
run() { timeout 180 rocprofv3 --kernel-trace --pmc "${@:2}" -d $O/$1 -o pmc -- python tools/pwh_probe.py f16 $S $C > $O/$1.log 2>&1 || echo "$1: rc $?"; }
S=$1; C=$2
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run p2 FETCH_SIZE WRITE_SIZE
run p3 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
run p4 SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_ANY
for p in p1 p2 p3 p4; do python tools/rocpd_summary.py pmc "$(ls $O/$p/*.db 2>/dev/null | head -1)" 2>/dev/null | grep -E "pwh|counter" ; done
