#!/bin/bash
# developer aid: register/LDS/scratch use of every kernel of engine.hip (device-only compile, ~12 s)
cd "$(dirname "$0")/../rac-2d_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -mllvm -disable-machine-licm --cuda-device-only "$@" -c engine.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|Occupancy|SGPRs Spill|ScratchSize|error" | sed -e 's/.*remark: *//' | paste -sd' ' | sed -e 's/Function Name: /\n/g' | grep -E "${FILTER:-.}"
