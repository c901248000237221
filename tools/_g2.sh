mkdir -p gpurun_out/g2; O=gpurun_out/g2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math tools/cut_probe.hip -o /tmp/cut_probe 2>/dev/null && /tmp/cut_probe > $O/cut_probe.txt 2>&1
python -m pytest tests/test_gpu_mbr.py tests/test_gpu_head.py tests/test_gpu_split_range.py tests/test_gpu_ops.py tests/test_gpu_graph.py tests/test_gpu_narrow.py -q 2>&1 | tail -15 > $O/t1.txt
python bench.py --no-cpu-baseline --no-other-configs --no-latency --no-fp32-forms > $O/b_c2.json 2> $O/b_c2.err
python bench.py --no-cpu-baseline --no-other-configs --no-latency --no-fp32-forms --depth 1 --per-op > $O/b_c2_d1.json 2> $O/perop_c2.txt
B="python bench.py --model efficientnetb0 --batch 128 --dtype bf16 --no-cpu-baseline --no-other-configs --no-latency"
$B > $O/b_c3se.json 2> /dev/null
B5="python bench.py --model efficientnetb3 --size 640 --batch 32 --dtype f16 --no-cpu-baseline --no-other-configs --no-latency"
$B5 > $O/b_c5se.json 2> /dev/null
for f in $O/b_*.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d.get('steps_in_flight'))"; done
cat $O/cut_probe.txt $O/t1.txt
