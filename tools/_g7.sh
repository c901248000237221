O=gpurun_out/g7; mkdir -p $O
python -m pytest tests/test_gpu_postprocess.py tests/test_gpu_yolo.py -q 2>&1 | tail -6 > $O/t1.txt
cat $O/t1.txt
B="python bench.py --no-cpu-baseline --no-other-configs --no-latency --no-fp32-forms"
$B --depth 1 --per-op > $O/b_c2_d1.json 2> $O/perop_c2.txt
$B > $O/b_c2.json 2> /dev/null
B3="python bench.py --model efficientnetb0 --batch 128 --dtype bf16 --no-cpu-baseline --no-other-configs --no-latency"
$B3 --depth 1 --per-op > $O/b_c3se_d1.json 2> $O/perop_c3se.txt
$B3 > $O/b_c3se.json 2> /dev/null
B5="python bench.py --model efficientnetb3 --size 640 --batch 32 --dtype f16 --no-cpu-baseline --no-other-configs --no-latency"
$B5 --depth 1 --per-op > $O/b_c5se_d1.json 2> $O/perop_c5se.txt
$B5 > $O/b_c5se.json 2> /dev/null
cp yoloret_amd/libyoloret_hip.so /tmp/head.so
cp tools/_ab/prev.so yoloret_amd/libyoloret_hip.so
YOLORET_HEAD_WALK16_MAX_NK=0 $B3 > $O/prev_c3se.json 2> /dev/null
YOLORET_HEAD_WALK16_MAX_NK=0 $B5 > $O/prev_c5se.json 2> /dev/null
cp /tmp/head.so yoloret_amd/libyoloret_hip.so
for f in $O/*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('%-30s %9.1f img/s  %.4f ms/step  in flight %s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], d.get('steps_in_flight')))"; done
grep -E "^nms|^decode" $O/perop_c2.txt $O/perop_c3se.txt $O/perop_c5se.txt | cut -c1-140
