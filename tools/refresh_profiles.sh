#!/bin/bash
# Regenerates the round's evidence under gpurun_out/refresh on a GPU box (run through gpurun from the repo root); the
# summaries are then copied into profiles/ (tools/refresh_profiles.sh && cp gpurun_out/refresh/... profiles/rNN_...).
# Per configuration: the bench line (steps in flight as shipped), the same with --depth 1 and the per-op table (kernels timed
# one after the other: what the per-kernel roofline uses), rocprofv3 --kernel-trace --stats of the --depth 1 command (its
# per-symbol averages must agree with the live hipEvent figures) and of the shipped depth (kernels of different steps
# overlap: longer per-kernel durations, shorter steps), and PMC passes in their own runs with --kernel-trace only.
#   tools/refresh_profiles.sh [c2|c3|c4|c5|dist ...]   (default: all)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R" || exit 1
O=gpurun_out/refresh
mkdir -p $O
db() { ls $O/$1/*.db 2>/dev/null | head -1; }
WHAT=${*:-c2 c3 c4 c5 c3se c5se dist}
ROUND=${ROUND:-04}

one() {   # one <tag> <traffic-suffix> <sq: 0|1> <bench args...>
  local tag=$1 suf=$2 sq=$3; shift 3
  export YOLORET_TUNE_CACHE=$R/$O/tuned_$tag.json   # the first run tunes and saves; profiled runs reuse the table (no trial launches)
  rm -f $YOLORET_TUNE_CACHE
  python bench.py "$@" --depth 1 --steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-other-configs --profile-iters 0 > /dev/null 2> $O/tune_$tag.err
  # ---- counters first (their traffic table is what the bench lines below quote as `traffic`)
  local BP="python bench.py $* --depth 1 --steps 5 --warmup 2 --no-cpu-baseline --no-latency --no-other-configs --profile-iters 0"
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf_$tag -o pmc -- $BP > /dev/null 2> $O/pmc_$tag.err
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw_$tag -o pmc -- $BP > /dev/null 2>> $O/pmc_$tag.err
  python tools/rocpd_summary.py pmc "$(db pf_$tag)" > $O/pmc_fetch_$tag.txt
  python tools/rocpd_summary.py pmc "$(db pw_$tag)" > $O/pmc_write_$tag.txt
  python tools/rocpd_summary.py traffic "$(db pf_$tag)" "$(db pw_$tag)" > $O/traffic$suf.json
  cp $O/traffic$suf.json profiles/r${ROUND}_traffic$suf.json      # (in this run's copy of the tree: bench.py reads it from profiles/)
  if [ "$sq" = 1 ]; then
    timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU -d $O/pq_$tag -o pmc -- $BP > /dev/null 2>> $O/pmc_$tag.err
    python tools/rocpd_summary.py pmc "$(db pq_$tag)" > $O/pmc_sq_$tag.txt
    timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum -d $O/pl_$tag -o pmc -- $BP > /dev/null 2>> $O/pmc_$tag.err
    python tools/rocpd_summary.py pmc "$(db pl_$tag)" > $O/pmc_l2_$tag.txt
  fi
  # ---- the bench lines: as shipped (steps in flight), and with --depth 1 + the per-op table
  python bench.py "$@" $CPU > $O/bench_$tag.json 2> $O/bench_$tag.err
  python bench.py "$@" --depth 1 --no-cpu-baseline --no-other-configs --per-op > $O/bench_${tag}_depth1.json 2> $O/perop_$tag.txt
  # ---- rocprofv3 kernel trace of the --depth 1 command (per-symbol averages == the live hipEvent figures) and of the shipped depth
  local B1="python bench.py $* --depth 1 --no-cpu-baseline --no-latency --no-other-configs"
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/ps_$tag -o stats -- $B1 > $O/bench_${tag}_under_rocprof.json 2> $O/rocprof_$tag.err
  python tools/rocpd_summary.py stats "$(db ps_$tag)" > $O/kernel_stats_$tag.txt
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/ps3_$tag -o stats -- python bench.py "$@" --no-cpu-baseline --no-latency --no-other-configs > /dev/null 2>> $O/rocprof_$tag.err
  python tools/rocpd_summary.py stats "$(db ps3_$tag)" > $O/kernel_stats_${tag}_in_flight.txt
  rm -rf $O/ps_$tag $O/ps3_$tag $O/pf_$tag $O/pw_$tag $O/pq_$tag $O/pl_$tag
  unset YOLORET_TUNE_CACHE
}

for w in $WHAT; do
  case $w in
    c2) CPU="" one c2 "" 1 ;;
    c3) CPU="--no-cpu-baseline" one c3 _efficientnetb0lite_416_b128_bf16 1 --model efficientnetb0-lite --batch 128 --dtype bf16 ;;
    c4) CPU="--no-cpu-baseline" one c4 _mobilenetv2x14_512_b64_f32 0 --model mobilenetv2x14 --size 512 --batch 64 ;;
    c5) CPU="--no-cpu-baseline" one c5 _efficientnetb3lite_640_b32_f16 1 --model efficientnetb3-lite --size 640 --batch 32 --dtype f16 ;;
    # the reference's own EfficientNets (squeeze-excite + swish, efficientnet.py:406-536), same batch / size / type as c3 / c5
    c3se) CPU="--no-cpu-baseline" one c3se _efficientnetb0_416_b128_bf16 1 --model efficientnetb0 --batch 128 --dtype bf16 ;;
    c5se) CPU="--no-cpu-baseline" one c5se _efficientnetb3_640_b32_f16 1 --model efficientnetb3 --size 640 --batch 32 --dtype f16 ;;
    dist)   # the N > 1 path with one rank: RCCL initialised, the all-gather of the records issued for real on its own stream
      python bench.py --force-dist --no-cpu-baseline --no-latency --no-other-configs > $O/bench_c2_force_dist.json 2> /dev/null
      python bench.py --force-dist --depth 1 --no-cpu-baseline --no-latency > $O/bench_c2_force_dist_depth1.json 2> /dev/null
      python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_c2_bf16.json 2> /dev/null ;;
  esac
done
for f in $O/bench*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d['roofline']
    print('%-40s %9.1f img/s %8.4f ms/step in flight %s  p50(B=1) %s  dominant %s frac %.3f traffic %s' % (sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d.get('steps_in_flight'), d.get('p50_ms_b1'), r['kernel'], r['frac'], r.get('traffic')))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done | tee $O/summary.txt
