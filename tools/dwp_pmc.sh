#!/bin/bash
# PMC passes over tools/dw5_probe.py (the LDS-tiled depthwise kernels stand-alone): gpurun -- bash tools/dwp_pmc.sh [dtype act]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R" || exit 1
O=gpurun_out/dwp_pmc
mkdir -p $O
db() { ls $O/$1/*.db 2>/dev/null | head -1; }
CMD="python tools/dw5_probe.py ${1:-bf16} ${2:-relu6}"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d $O/a -o pmc -- $CMD > /dev/null 2> $O/a.err
python tools/rocpd_summary.py pmc "$(db a)" > $O/sq_a.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS -d $O/b -o pmc -- $CMD > /dev/null 2> $O/b.err
python tools/rocpd_summary.py pmc "$(db b)" > $O/sq_b.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/s -o st -- $CMD > /dev/null 2> $O/s.err
python tools/rocpd_summary.py stats "$(db s)" > $O/stats.txt
grep -h "dwp_kernel" $O/sq_a.txt $O/sq_b.txt $O/stats.txt
