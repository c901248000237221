#!/bin/bash
# PMC passes over tools/dwq_probe.py (the 16-bit stride-1 depthwise forms stand-alone): gpurun -- bash tools/dwq_pmc.sh [shape filter] [se]
# e.g. bash tools/dwq_pmc.sh 26x26x672,52x52x240 ; YOLORET_DW_WALK=0 for the tile walk.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R" || exit 1
O=gpurun_out/dwq_pmc${YOLORET_DW_WALK:+_tile}
mkdir -p $O
db() { ls $O/$1/*.db 2>/dev/null | head -1; }
export YR_PROBE_ONLY=${1:-26x26x672}
CMD="python tools/dwq_probe.py $2"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT -d $O/a -o pmc -- $CMD > /dev/null 2> $O/a.err
python tools/rocpd_summary.py pmc "$(db a)" > $O/sq_a.txt
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS -d $O/b -o pmc -- $CMD > /dev/null 2> $O/b.err
python tools/rocpd_summary.py pmc "$(db b)" > $O/sq_b.txt
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o pmc -- $CMD > /dev/null 2> $O/f.err
python tools/rocpd_summary.py pmc "$(db f)" > $O/fetch.txt
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o pmc -- $CMD > /dev/null 2> $O/w.err
python tools/rocpd_summary.py pmc "$(db w)" > $O/write.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/s -o st -- $CMD > $O/probe.txt 2> $O/s.err
python tools/rocpd_summary.py stats "$(db s)" > $O/stats.txt
rm -rf $O/a $O/b $O/f $O/w $O/s
grep -h "dw[pq]_kernel" $O/sq_a.txt $O/sq_b.txt $O/fetch.txt $O/write.txt $O/stats.txt
