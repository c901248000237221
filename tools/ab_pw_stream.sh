#!/bin/bash
# Same-box A/B of the pixel-stationary 1x1 conv form (pointwise_stream.hip): the bench line with YOLORET_PW_STREAM=0 (the tiled split
# kernel everywhere: the first half of round 6), = 1 without the two-output pairs, = 1 with them (shipped), alternating; then the per-op
# table of the pointwise convs for both.   bash tools/ab_pw_stream.sh > gpurun_out/ab_pw_stream.txt
line() { python bench.py --no-cpu-baseline --no-other-configs --no-latency --no-fp32-forms --steps 40 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in flight %.1f img/s  serial %.1f img/s  launches %d' % (d['value'], d['serial_steps']['img_s'], d['roofline_step']['launches_per_step']))"; }
for rep in 1 2; do
  echo "YOLORET_PW_STREAM=0:                          $(YOLORET_PW_STREAM=0 line)"
  echo "YOLORET_PW_STREAM=1 YOLORET_PW_STREAM_PAIRS=0: $(YOLORET_PW_STREAM_PAIRS=0 line)"
  echo "YOLORET_PW_STREAM=1 (shipped):                $(line)"
done
for v in 0 1; do
  echo "--- per-op, YOLORET_PW_STREAM=$v (serial, ms at 64 images)"
  YOLORET_PW_STREAM=$v python bench.py --depth 1 --no-cpu-baseline --no-other-configs --no-latency --no-fp32-forms --per-op 2>&1 >/dev/null | grep -E "pwt_kernel|pws_kernel" | grep -v SYMBOL | awk '{printf "%-20s %-26s %s\n", $1, $2, $3}'
done
