#!/bin/bash
# Same-box A/B of the c2 bench under two environments:  tools/ab_bench.sh "ENV_A=.." "ENV_B=.." [rounds] [extra bench flags]
A="$1"; B="$2"; R=${3:-2}; shift 3
for i in $(seq $R); do for E in "$A" "$B"; do
  env $E python bench.py --no-cpu-baseline --no-latency --no-fp32-forms --no-other-configs --steps 40 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$E: in flight', d['value'], 'serial', d['serial_steps']['img_s'])"
done; done
