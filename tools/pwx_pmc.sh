#!/bin/bash
# PMC passes over the stand-alone pointwise harness: tools/pwx_pmc.sh <binary> <s|q> <variant> <tag>
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R" || exit 1
B=$1; F=$2; V=$3; TAG=$4
O=gpurun_out/pwx_pmc_$TAG
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p1 -o pmc -- $B $F $V > /dev/null 2> $O/err1.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/p2 -o pmc -- $B $F $V > /dev/null 2> $O/err2.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_WAVES TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum -d $O/p3 -o pmc -- $B $F $V > /dev/null 2> $O/err3.txt
for p in p1 p2 p3; do python tools/rocpd_summary.py pmc "$(ls $O/$p/*.db | head -1)" > $O/$p.txt 2>&1; done
rm -rf $O/p1 $O/p2 $O/p3
cat $O/p1.txt $O/p2.txt $O/p3.txt | cut -c1-260
