# same-box A/B: the library of commit 4cbe96b (before the split forms' VALU diet) against HEAD's, alternating
O=gpurun_out/g6; mkdir -p $O
cp yoloret_amd/libyoloret_hip.so /tmp/head.so
B="python bench.py --no-cpu-baseline --no-other-configs --no-latency --no-fp32-forms"
for r in 1 2; do
  cp tools/_ab/prev.so yoloret_amd/libyoloret_hip.so
  $B > $O/prev_inflight_$r.json 2> /dev/null
  $B --depth 1 --per-op > $O/prev_serial_$r.json 2> $O/perop_prev_$r.txt
  cp /tmp/head.so yoloret_amd/libyoloret_hip.so
  $B > $O/head_inflight_$r.json 2> /dev/null
  $B --depth 1 --per-op > $O/head_serial_$r.json 2> $O/perop_head_$r.txt
done
B4="python bench.py --model mobilenetv2x14 --size 512 --batch 64 --no-cpu-baseline --no-other-configs --no-latency --no-fp32-forms"
cp tools/_ab/prev.so yoloret_amd/libyoloret_hip.so; $B4 > $O/prev_c4.json 2> /dev/null
cp /tmp/head.so yoloret_amd/libyoloret_hip.so; $B4 > $O/head_c4.json 2> /dev/null
for f in $O/*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('%-44s %9.1f img/s  %.4f ms/step  in flight %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], d.get('steps_in_flight')))"; done | tee $O/ab_summary.txt
