#!/bin/bash
# usage: spillmap.sh <mangled-prefix>
cd "$(dirname "$0")/../../free-hunch_amd/csrc" && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -munsafe-fp-atomics -S --cuda-device-only fh_kernels.hip -o /tmp/fhk.s 2>/dev/null
awk "/^$1/,/\.end_amdhsa_kernel/" /tmp/fhk.s > /tmp/pipe.s
grep -n "scratch_\|s_barrier\|sc1\|s_endpgm\|global_load_dwordx4.*nt\|global_store\|global_atomic\|v_writelane\|v_readlane" /tmp/pipe.s | awk '{print $1" "$2}' | sed 's/global_load_dwordx4/GL4/;s/scratch_load_dwordx2/SL2/;s/scratch_load_dwordx4/SL4/;s/scratch_load_dword/SL/;s/scratch_store_dwordx2/SS2/;s/scratch_store_dwordx4/SS4/;s/scratch_store_dword/SS/;s/v_writelane_b32/WL/;s/v_readlane_b32/RL/;s/s_barrier/BAR/' | paste -sd' ' | fold -w 230
