#!/bin/bash
# Register / scratch use of every kernel in one .hip file (compile-time report, no GPU needed):
#   tools/kernel_regs.sh yoloret_amd/csrc/mbh.hip [mangled-name filter]
f=$(realpath "$1")
cd /tmp || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I/root/repo/include \
    --cuda-device-only -c "$f" -o /tmp/_regs.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  grep -E "Function Name|  VGPRs:|AGPRs|ScratchSize|Occupancy" | sed -e 's/.*remark: [^ ]* *//' -e 's/ \[-Rpass.*//' |
  paste - - - - - | sed -e 's/Function Name: //' | grep -E "${2:-.}"
