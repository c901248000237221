#!/bin/bash
# p50 at batch 1/2/4/8 of plan variants (tools/lat_sweep.py)
for cfg in "" "YOLORET_FUSE_HEAD=0" "YOLORET_SE_TAIL=1" "YOLORET_FUSE_HEAD=0 YOLORET_SE_TAIL=1" "YOLORET_HEAD_DMA=0 YOLORET_SE_TAIL=1" "YOLORET_HEAD_WALK=0 YOLORET_SE_TAIL=1"; do
  echo "== $cfg: $(env $cfg timeout 300 python tools/lat_sweep.py 2>&1 | tail -1)"
done
