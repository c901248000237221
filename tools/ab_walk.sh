#!/bin/bash
# c3 / c5 and the SE EfficientNets with the walking depthwise form (default) and with the tile walk (YOLORET_DW_WALK=0): img/s in flight and serial.
mkdir -p gpurun_out/ab
run() {  # run <tag> <bench args...>
  local tag=$1; shift
  for walk in 1 0; do
    export YOLORET_DW_WALK=$walk
    export YOLORET_TUNE_CACHE=$PWD/gpurun_out/ab/tuned_${tag}_$walk.json
    python bench.py "$@" --no-cpu-baseline --no-latency --no-other-configs > gpurun_out/ab/${tag}_walk$walk.json 2> gpurun_out/ab/${tag}_walk$walk.err
    python - gpurun_out/ab/${tag}_walk$walk.json $tag $walk <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print('%-6s walk=%s  in flight %9.1f img/s   serial %9.1f img/s   dominant %s' % (sys.argv[2], sys.argv[3], d['value'], d.get('serial_steps', {}).get('img_s', 0), d['roofline']['kernel']))
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'FAILED', e)
PY
  done
}
for w in ${*:-c3se c5se c3 c5}; do
  case $w in
    c3) run c3 --model efficientnetb0-lite --batch 128 --dtype bf16 ;;
    c5) run c5 --model efficientnetb3-lite --size 640 --batch 32 --dtype f16 ;;
    c3se) run c3se --model efficientnetb0 --batch 128 --dtype bf16 ;;
    c5se) run c5se --model efficientnetb3 --size 640 --batch 32 --dtype f16 ;;
  esac
done
