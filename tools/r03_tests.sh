#!/bin/bash
# Full-resolution / full-batch parity tests with their printed error and agreement lines kept: gpurun_out/r03t/fullres_tests.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R" || exit 1
O=gpurun_out/r03t
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullbatch.py tests/test_gpu_narrow.py tests/test_gpu_graph.py -q -s -k "full_batch or baseline_configs_16bit or full_resolution or c2_batch64" 2>&1 | grep -v "amdgpu.ids" > $O/fullres_tests.txt
tail -30 $O/fullres_tests.txt | cut -c1-250
