mkdir -p gpurun_out/g1; O=gpurun_out/g1
python -m pytest tests/test_gpu_head.py -q -x -k "16bit" 2>&1 | tail -15 > $O/t1.txt
python -m pytest tests/test_gpu_narrow.py tests/test_gpu_fullbatch.py -q 2>&1 | tail -25 > $O/t2.txt
B="python bench.py --model efficientnetb0 --batch 128 --dtype bf16 --no-cpu-baseline --no-other-configs --no-latency"
$B --depth 1 --per-op > $O/b_c3se_d1.json 2> $O/perop_c3se.txt
$B > $O/b_c3se.json 2> $O/b_c3se.err
YOLORET_FUSE_HEAD=0 $B > $O/b_c3se_nohead.json 2> /dev/null
YOLORET_FUSE_HEAD=0 $B --depth 1 > $O/b_c3se_nohead_d1.json 2> /dev/null
B5="python bench.py --model efficientnetb3 --size 640 --batch 32 --dtype f16 --no-cpu-baseline --no-other-configs --no-latency"
$B5 --depth 1 --per-op > $O/b_c5se_d1.json 2> $O/perop_c5se.txt
$B5 > $O/b_c5se.json 2> /dev/null
YOLORET_FUSE_HEAD=0 $B5 > $O/b_c5se_nohead.json 2> /dev/null
for f in $O/b_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d.get('steps_in_flight'))"; done
cat $O/t1.txt $O/t2.txt
