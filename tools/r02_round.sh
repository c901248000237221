#!/bin/bash
# Round-2 GPU pass: the whole -m gpu suite, then the headline bench and the 16-bit configurations with per-op tables.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R" || exit 1
O=gpurun_out/r02
mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.txt
python bench.py --per-op > $O/bench_c2_f32.json 2> $O/perop_c2_f32.txt
python bench.py --dtype bf16 --per-op --no-cpu-baseline > $O/bench_c2_bf16.json 2> $O/perop_c2_bf16.txt
python bench.py --model efficientnetb0 --batch 128 --dtype bf16 --per-op --no-cpu-baseline > $O/bench_c3_b0_bf16.json 2> $O/perop_c3_b0_bf16.txt
python bench.py --model efficientnetb0 --batch 128 --per-op --no-cpu-baseline > $O/bench_c3_b0_f32.json 2> $O/perop_c3_b0_f32.txt
python bench.py --model efficientnetb0-lite --batch 128 --dtype bf16 --per-op --no-cpu-baseline > $O/bench_c3_b0lite_bf16.json 2> $O/perop_c3_b0lite_bf16.txt
python bench.py --model efficientnetb3 --size 640 --batch 32 --dtype f16 --per-op --no-cpu-baseline > $O/bench_c5_b3_f16.json 2> $O/perop_c5_b3_f16.txt
python bench.py --model efficientnetb3 --size 640 --batch 32 --per-op --no-cpu-baseline > $O/bench_c5_b3_f32.json 2> $O/perop_c5_b3_f32.txt
tail -3 $O/pytest_gpu.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], d['value'], 'img/s', d['ms_per_step'], 'ms', 'p50', d.get('p50_ms_b1'), d['roofline']['kernel'], d['roofline']['frac'], 'h2d', (d.get('incl_h2d') or {}).get('img_s'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
