#!/bin/bash
# Same-box A/B behind the `fallback_ops` of the squeeze-excite configurations: every op on that list runs an unfused
# form BY CHOICE - the fused form exists and is switched off by a measured default.  This prints the measurement:
#   default | the 5x5 stride-1 blocks' expand + depthwise fused (YR_OP_MBX) | every walkable head block fused (YR_OP_HEAD, 16-bit walking form)
run() {  # env, model, dtype, size, batch
  env $1 python bench.py --model $2 --dtype $3 --size $4 --batch $5 --no-cpu-baseline --no-latency --no-fp32-forms --no-other-configs --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-46s %s: in flight %8.1f  serial %8.1f  launches %d  fallback_ops %d' % ('$1', '$2', d['value'], d['serial_steps']['img_s'], d['roofline_step']['launches_per_step'], len(d['roofline_step']['fallback_ops'])))"
}
for i in 1 2; do
  for E in "YR_AB=default" "YOLORET_MBX_K5_MAX_CEXP=100000" "YOLORET_HEAD_WALK16_MAX_NK=8" "YOLORET_MBX_K5_MAX_CEXP=100000 YOLORET_HEAD_WALK16_MAX_NK=8"; do
    run "$E" efficientnetb0 bf16 416 128
    run "$E" efficientnetb3 f16 640 32
  done
done
