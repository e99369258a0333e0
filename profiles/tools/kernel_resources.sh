#!/bin/bash
# compile fh_kernels.hip for gfx950 and print the resource usage of kernels matching $1
cd "$(dirname "$0")/../../free-hunch_amd/csrc"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -munsafe-fp-atomics -c fh_kernels.hip -o /tmp/fhk.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A10 "Function Name: .*$1" | grep -E "Function Name|VGPRs|Scratch|SGPRs|error" 
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -munsafe-fp-atomics -c fh_kernels.hip -o /tmp/fhk.o 2>&1 | grep -E "error" | head
