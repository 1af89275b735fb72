#!/bin/bash
# Register pressure of ONE trace-kernel instantiation in a few seconds (no GPU needed):
#   tools/diag/probe_kernel.sh 'trace_pool_kernel<false, 1024, false>' [extra hipcc flags]
# prints VGPRs / spills / scratch from -Rpass-analysis=kernel-resource-usage and leaves the ISA in /tmp/ff_probe.s
K=${1:-trace_pool_kernel<false, 1024, false>}; shift
cd "$(dirname "$0")/../../gpupathtracer_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize \
  --cuda-device-only -S -o /tmp/ff_probe.s ff_kernels.hip "-DFF_PROBE=$K" -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  grep -E "VGPRs:|ScratchSize|SGPRs Spill|VGPRs Spill|TotalSGPRs|error" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | head -5 | tr "\n" ";"; echo
