#!/bin/bash
# which kernel family breaks with three steps in flight?  (debug aid for tools/inflight_probe.py)
for cfg in "" "YOLORET_FUSE_HEAD=0" "YOLORET_HEAD_WALK=0" "YOLORET_HEAD_DMA=0" "YOLORET_HEAD_WALK=0 YOLORET_HEAD_DMA=0"; do
  echo "== $cfg"
  env $cfg timeout 200 python tools/inflight_probe.py 2>&1 | grep -E "in flight|round 0" | head -8
done
