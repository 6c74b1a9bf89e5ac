#!/bin/bash
# builds timing-experiment variants of the library: tools/build_variants.sh name1:"-DFLAGS" name2:"..."
# (base flags = the Makefile's without CODEGEN, so that scheduler / allocator options can be varied; compare against a
#  variant built with the Makefile's CODEGEN flags, and time them with tools/bench_libs.py)
cd "$(dirname "$0")/../trajopt_amd/csrc"
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  mkdir -p ../_build/v_$name
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -Wno-unused-function $flags -x hip -shared -o ../_build/v_$name/lib.so tmx_api.cpp -L/opt/rocm/lib -lrccl &
done
wait
ls ../_build/
