mkdir -p gpurun_out/g3; O=gpurun_out/g3
python -m pytest tests/test_gpu_mbr.py tests/test_gpu_head.py tests/test_gpu_split_range.py tests/test_gpu_ops.py tests/test_gpu_graph.py tests/test_gpu_hoist.py -q 2>&1 | tail -25 > $O/t1.txt
python bench.py --no-cpu-baseline --no-other-configs --no-latency --no-fp32-forms > $O/b_c2.json 2> $O/b_c2.err
python bench.py --no-cpu-baseline --no-other-configs --no-latency --no-fp32-forms --depth 1 --per-op > $O/b_c2_d1.json 2> $O/perop_c2.txt
python bench.py --model mobilenetv2x14 --size 512 --batch 64 --no-cpu-baseline --no-other-configs --no-latency --no-fp32-forms > $O/b_c4.json 2> /dev/null
for f in $O/b_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d.get('steps_in_flight'))"; done
cat $O/t1.txt
