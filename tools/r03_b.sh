#!/bin/bash
# c3 / c5 / c2-bf16 benches with per-op tables (after a kernel change): gpurun_out/r03b
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R" || exit 1
O=gpurun_out/${1:-r03b}
mkdir -p $O
python bench.py --model efficientnetb0-lite --batch 128 --dtype bf16 --per-op --no-cpu-baseline > $O/bench_c3.json 2> $O/perop_c3.txt
python bench.py --model efficientnetb3-lite --size 640 --batch 32 --dtype f16 --per-op --no-cpu-baseline > $O/bench_c5.json 2> $O/perop_c5.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], d['value'], 'img/s', d['ms_per_step'], 'ms', 'p50', d.get('p50_ms_b1'), d['roofline']['kernel'], d['roofline']['frac'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
grep "^SYMBOL" $O/perop_c3.txt | head -12
grep "^SYMBOL" $O/perop_c5.txt | head -12
