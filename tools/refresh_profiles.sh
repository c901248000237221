#!/bin/bash
# Regenerates the evidence under gpurun_out/refresh on a GPU box (run through gpurun from the repo root); the summaries
# are then copied into profiles/ by hand (tools/refresh_profiles.sh && cp gpurun_out/refresh/... profiles/rNN_...).
# PMC counters are collected in their own passes with --kernel-trace only.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R" || exit 1
O=gpurun_out/refresh
rm -rf $O; mkdir -p $O
db() { ls $O/$1/*.db 2>/dev/null | head -1; }

# ---------------------------------------------------------------- headline: config 2 (MobileNetV2x0.75 @416, batch 64, fp32)
export YOLORET_TUNE_CACHE=$R/$O/tuned.json   # the first run tunes and saves; the profiled runs reuse the table (no trial launches)
python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o stats -- python bench.py --no-cpu-baseline --no-latency > $O/bench_under_rocprof.json 2> $O/rocprof_stats.err
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency --profile-iters 0"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch -o pmc -- $B > /dev/null 2> $O/pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -o pmc -- $B > /dev/null 2> $O/pmc_write.err
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SMEM -d $O/prof_sq -o pmc -- $B > /dev/null 2> $O/pmc_sq.err
python bench.py --per-op --no-cpu-baseline --no-latency > $O/bench_perop.json 2> $O/perop.txt
python tools/rocpd_summary.py stats "$(db prof_stats)" > $O/kernel_stats.txt
python tools/rocpd_summary.py pmc "$(db prof_fetch)" > $O/pmc_fetch.txt
python tools/rocpd_summary.py pmc "$(db prof_write)" > $O/pmc_write.txt
python tools/rocpd_summary.py pmc "$(db prof_sq)" > $O/pmc_sq.txt
python tools/rocpd_summary.py traffic "$(db prof_fetch)" "$(db prof_write)" > $O/traffic.json
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_sq

# ---------------------------------------------------------------- the same model with 16-bit activations (bf16 MFMA)
export YOLORET_TUNE_CACHE=$R/$O/tuned_bf16.json
python bench.py --dtype bf16 --no-cpu-baseline --per-op > $O/bench_c2_bf16.json 2> $O/perop_c2_bf16.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats16 -o stats -- python bench.py --dtype bf16 --no-cpu-baseline --no-latency > /dev/null 2> $O/rocprof_stats16.err
B16="python bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-latency --profile-iters 0"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/prof_sq16 -o pmc -- $B16 > /dev/null 2> $O/pmc_sq16.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch16 -o pmc -- $B16 > /dev/null 2> $O/pmc_fetch16.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write16 -o pmc -- $B16 > /dev/null 2> $O/pmc_write16.err
python tools/rocpd_summary.py stats "$(db prof_stats16)" > $O/kernel_stats_c2_bf16.txt
python tools/rocpd_summary.py pmc "$(db prof_sq16)" > $O/pmc_sq_c2_bf16.txt
python tools/rocpd_summary.py traffic "$(db prof_fetch16)" "$(db prof_write16)" > $O/traffic_c2_bf16.json
rm -rf $O/prof_stats16 $O/prof_sq16 $O/prof_fetch16 $O/prof_write16
unset YOLORET_TUNE_CACHE

# ---------------------------------------------------------------- the other BASELINE.json configurations (one GPU's share)
python bench.py --model efficientnetb0 --batch 128 --dtype bf16 --no-cpu-baseline --per-op > $O/bench_c3_effb0_bf16.json 2> $O/perop_c3_effb0_bf16.txt
python bench.py --model efficientnetb0-lite --batch 128 --dtype bf16 --no-cpu-baseline --per-op > $O/bench_c3_effb0lite_bf16.json 2> $O/perop_c3_effb0lite_bf16.txt
python bench.py --model efficientnetb0 --batch 128 --no-cpu-baseline > $O/bench_c3_effb0_f32.json 2> /dev/null
python bench.py --model mobilenetv2x14 --size 512 --batch 64 --no-cpu-baseline --per-op > $O/bench_c4_mbv2x14_f32.json 2> $O/perop_c4_mbv2x14_f32.txt
python bench.py --model efficientnetb3 --size 640 --batch 32 --dtype f16 --no-cpu-baseline --per-op > $O/bench_c5_effb3_f16.json 2> $O/perop_c5_effb3_f16.txt
python bench.py --model efficientnetb3-lite --size 640 --batch 32 --dtype f16 --no-cpu-baseline --per-op > $O/bench_c5_effb3lite_f16.json 2> $O/perop_c5_effb3lite_f16.txt
python bench.py --model efficientnetb3 --size 640 --batch 32 --no-cpu-baseline > $O/bench_c5_effb3_f32.json 2> /dev/null
python tools/model_sweep.py > $O/model_sweep.txt 2>&1
for f in $O/bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print('%-44s %9.1f img/s  %8.4f ms/step  p50(B=1) %s  dominant %s frac %.3f' % (sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d.get('p50_ms_b1'), d['roofline']['kernel'], d['roofline']['frac']))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done | tee $O/summary.txt
head -14 $O/kernel_stats.txt
