#!/bin/bash
# builds timing-experiment variants of the library side by side: tools/build_variants.sh name1:"-DFLAGS" name2:"..."
#   -> trajopt_amd/_build/v_<name>/lib.so   (tmx_api.cpp with the variant's flags + the product's tmx_wave.o)
# Base flags = the Makefile's (CODEGEN included; set BASE="..." to vary scheduler / allocator options).  Time them with
# tools/bench_libs.py, profile them with tools/prof_phases.py.
cd "$(dirname "$0")/../trajopt_amd/csrc"
BASE=${BASE:--O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-function -fno-strict-aliasing -mllvm -amdgpu-remove-redundant-endcf=0}
[ -f ../_build/tmx_wave.o ] || make ../_build/tmx_wave.o
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  mkdir -p ../_build/v_$name
  ( /opt/rocm/bin/hipcc $BASE $flags -x hip -c -o ../_build/v_$name/tmx_api.o tmx_api.cpp &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../_build/v_$name/lib.so ../_build/v_$name/tmx_api.o ../_build/tmx_wave.o -L/opt/rocm/lib -lrccl &&
    rm -f ../_build/v_$name/tmx_api.o ) &
done
wait
ls ../_build/
