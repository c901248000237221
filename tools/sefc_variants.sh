#!/bin/bash
# debug aid: se_fc variants (objects prebuilt as csrc/_obj/elementwise.o.<tag>) beside the disturbing op
for tag in "$@"; do
  cp yoloret_amd/csrc/_obj/elementwise.o.$tag yoloret_amd/csrc/_obj/elementwise.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o yoloret_amd/libyoloret_hip.so $(python -c "from yoloret_amd import build as B; import os; print(' '.join(os.path.join(B.CSRC,'_obj',f.replace('.hip','.o')) for f in B.SOURCES))")
  echo "== variant $tag: $(YR_ONLY_OPS=${OPS:-11-11} REP=30 ITERS=6 timeout 200 python tools/sefc_probe2.py 2>&1 | tail -1)"
done
