O=gpurun_out/g5; mkdir -p $O
python -m pytest tests/test_gpu_head.py tests/test_gpu_postprocess.py -q 2>&1 | tail -6 > $O/t1.txt
B="python bench.py --no-cpu-baseline --no-other-configs --no-latency --no-fp32-forms"
$B > $O/b_c2.json 2> /dev/null
$B --depth 1 --per-op > $O/b_c2_d1.json 2> $O/perop_c2.txt
YOLORET_HEAD_WALK_MAX_NK=7 $B > $O/b_c2_nk7.json 2> /dev/null
YOLORET_HEAD_WALK_MAX_NK=7 $B --depth 1 --per-op > $O/b_c2_nk7_d1.json 2> $O/perop_c2_nk7.txt
python tools/_relink.py headwalk.hip -DHW_NOHOIST_MIN_NKE=4 > /dev/null 2>&1
$B --depth 1 --per-op > $O/b_c2_nohoist4_d1.json 2> $O/perop_c2_nohoist4.txt
$B > $O/b_c2_b.json 2> /dev/null
for f in $O/b_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d.get('steps_in_flight'))"; done
cat $O/t1.txt
grep -E "_head|decode" $O/perop_c2.txt | cut -c1-90; echo; grep -E "_head" $O/perop_c2_nk7.txt | cut -c1-90; echo; grep -E "_head" $O/perop_c2_nohoist4.txt | cut -c1-90
