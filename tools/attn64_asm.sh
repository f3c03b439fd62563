#!/bin/bash
# compile the attn64 kernels alone to ISA (/tmp/attn64.s) and print their resource usage
cd /root/repo/pyramid-flow_amd/csrc
cat > _inst_tmp.hip <<'EOT'
#include "attention.hip"
int pf_set_err(const char*) { return -1; }
namespace {
template __global__ void attn64_kernel<2, 0>(const AArgs);
template __global__ void attn64_kernel<2, 1>(const AArgs);
template __global__ void attn64_kernel<2, 4>(const AArgs);
template __global__ void attn64_kernel<2, 33>(const AArgs);
}
EOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -Wno-unused-value -S --cuda-device-only -o /tmp/attn64.s _inst_tmp.hip 2>&1 | grep -v "argument unused" | head -20
rm -f _inst_tmp.hip
grep -A25 "\.name:.*attn64_kernel" /tmp/attn64.s | grep "\.name\|vgpr_spill_count\|\.vgpr_count\|private_segment_fixed_size"
