#!/bin/bash
# PMC counters of the weight-streaming fused block kernel on one block:  bash tools/mbk_pmc.sh block_11   (MBK_PROBE_CFG=rows,nw)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; cd "$R" || exit 1
O=gpurun_out/mbk_pmc; rm -rf $O; mkdir -p $O
export MBK_PROBE_BATCH=64
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/p1 -o pmc -- python tools/mbk_probe.py "$@" > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $O/p2 -o pmc -- python tools/mbk_probe.py "$@" > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d $O/p3 -o pmc -- python tools/mbk_probe.py "$@" > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $O/p4 -o pmc -- python tools/mbk_probe.py "$@" > $O/p4.log 2>&1
for p in p1 p2 p3 p4; do python tools/rocpd_summary.py pmc "$(ls $O/$p/*.db | head -1)" | grep -E "mbk|counter" ; done | tee $O/summary.txt
tail -3 $O/p1.log
