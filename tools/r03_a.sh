#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R" || exit 1
O=gpurun_out/r03a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_narrow.py -x -q -k "pointwise16" 2>&1 | tail -15 > $O/pytest.txt
timeout 300 python tools/pwh_probe.py bf16 > $O/probe_bf16.txt 2>&1
timeout 300 python tools/pwh_probe.py f16 > $O/probe_f16.txt 2>&1
tail -5 $O/pytest.txt; cat $O/probe_bf16.txt $O/probe_f16.txt
