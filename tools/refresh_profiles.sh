#!/bin/bash
# Regenerates the evidence under gpurun_out/ on a GPU box (run through gpurun from the repo root); the summaries
# are then copied into profiles/ by hand.  PMC counters are collected in their own passes with --kernel-trace only.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R" || exit 1
O=gpurun_out/refresh
export YOLORET_TUNE_CACHE=$R/$O/tuned.json   # the first run tunes and saves; the profiled runs reuse the table (no trial launches)
rm -rf $O; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/prof_stats -o stats -- python bench.py --no-cpu-baseline --no-latency > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --profile-iters 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch -o pmc -- $B > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -o pmc -- $B > /dev/null 2> $O/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SMEM -d $O/prof_sq -o pmc -- $B > /dev/null 2> $O/pmc_sq.err
python bench.py --per-op --no-cpu-baseline --no-latency > $O/bench_perop.json 2> $O/perop.txt
db() { ls $O/$1/*.db 2>/dev/null | head -1; }
python tools/rocpd_summary.py stats "$(db prof_stats)" > $O/kernel_stats.txt
python tools/rocpd_summary.py pmc "$(db prof_fetch)" > $O/pmc_fetch.txt
python tools/rocpd_summary.py pmc "$(db prof_write)" > $O/pmc_write.txt
python tools/rocpd_summary.py pmc "$(db prof_sq)" > $O/pmc_sq.txt
python tools/rocpd_summary.py traffic "$(db prof_fetch)" "$(db prof_write)" > $O/traffic.json
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq
tail -c 1500 $O/bench.json; echo; head -12 $O/kernel_stats.txt
