#!/bin/bash
# Per-kernel register / LDS / scratch usage as hipcc reports it (-Rpass-analysis=kernel-resource-usage) for one device source.
# usage: scripts/kernel_resources.sh langevin-mcmc_amd/csrc/device/step_small_plain.hip
hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-result -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kr_$$.o 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|LDS Size|SGPRs:" | sed 's/.*remark: //' | paste - - - - - - - | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g'
rm -f /tmp/kr_$$.o
