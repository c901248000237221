#!/bin/bash
# c2 / c4 with the float32 pointwise convs in the split form (default) and on the float32 MFMA (YOLORET_PW_SPLIT=0).
mkdir -p gpurun_out/ab
run() {
  local tag=$1; shift
  for sp in 1 0; do
    export YOLORET_PW_SPLIT=$sp
    export YOLORET_TUNE_CACHE=$PWD/gpurun_out/ab/tuned_${tag}_pw$sp.json
    python bench.py "$@" --no-cpu-baseline --no-latency --no-other-configs > gpurun_out/ab/${tag}_pw$sp.json 2> gpurun_out/ab/${tag}_pw$sp.err
    python - gpurun_out/ab/${tag}_pw$sp.json $tag $sp <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print('%-4s pw split=%s  in flight %9.1f img/s   serial %9.1f img/s   dominant %s frac %.3f' % (sys.argv[2], sys.argv[3], d['value'], d.get('serial_steps', {}).get('img_s', 0), d['roofline']['kernel'], d['roofline']['frac']))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'FAILED', e)
PY
  done
}
for w in ${*:-c2 c4}; do
  case $w in
    c2) run c2 ;;
    c4) run c4 --model mobilenetv2x14 --size 512 --batch 64 ;;
  esac
done
