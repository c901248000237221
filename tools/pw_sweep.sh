#!/bin/bash
# per-layer pointwise timings for every forced tile shape (tuning aid)
for cfg in 256x16 128x32 128x48 128x64 128x80 128x96 128x128 64x16 64x32 64x48 64x64 64x80 64x96 64x128; do
  YR_PW_CFG=$cfg python bench.py --per-op --no-cpu-baseline --no-latency --steps 3 --warmup 1 --profile-iters 3 2>&1 >/dev/null | grep "pw_kernel" | grep -v SYMBOL | awk -v c=$cfg '{print c, $1, $3}'
done
