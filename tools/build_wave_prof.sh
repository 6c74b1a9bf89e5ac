#!/bin/bash
# product library + the -DTMX_WAVE_PROF variant of the one-wave kernels (trajopt_amd/_build_prof/, for tools/wave_prof.py / wave_iter_cost.py)
set -e
cd "$(dirname "$0")/../trajopt_amd/csrc"
make 2>&1 | grep -i "error" -A5 || true
mkdir -p ../_build_prof
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -Wall -Wno-unused-function -fno-strict-aliasing -mllvm -amdgpu-remove-redundant-endcf=0"
/opt/rocm/bin/hipcc $FLAGS -DTMX_WAVE_PROF -x hip -c -o ../_build_prof/tmx_wave.o tmx_wave.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../_build_prof/libtrajopt_mi355x.so ../_build/tmx_api.o ../_build_prof/tmx_wave.o -L/opt/rocm/lib -lrccl
