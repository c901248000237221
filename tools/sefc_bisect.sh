#!/bin/bash
LIST=1 ITERS=1 timeout 200 python tools/sefc_probe2.py 2>&1 | grep -E "^[0-9]+ " | tr '\n' ';' | cut -c1-3000; echo
for r in "$@"; do
  # (needs a library built with the debug hooks: python tools/relink.py runtime.hip -DYR_DEBUG_HOOKS)
  echo "== ops $r: $(YR_ONLY_OPS=$r REP=${REP:-4} ITERS=6 timeout 200 python tools/sefc_probe2.py 2>&1 | tail -1)"
done
