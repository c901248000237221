#!/bin/bash
# The three 16-bit stride-1 depthwise forms side by side (digests must agree); run through gpurun from the repo root.
mkdir -p gpurun_out
for se in "" se; do
  echo "== walking form (depthwise_walk.hip) $se"; timeout 300 python tools/dwq_probe.py $se
  echo "== tile walk (depthwise_lds.hip) $se"; YOLORET_DW_WALK=0 timeout 300 python tools/dwq_probe.py $se
  echo "== dw_kernel $se"; YOLORET_DW_LDS=0 timeout 300 python tools/dwq_probe.py $se
done
