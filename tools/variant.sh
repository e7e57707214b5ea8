#!/bin/bash
# usage: tools/variant.sh <name> <file.hip> <extra hipcc flags...>
# builds pygpa_amd/variants/libgpa_<name>.so with one translation unit recompiled with the extra flags
# (performance experiments; select with GPA_HIP_LIB=<path> at run time)
set -e
name=$1; tu=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p pygpa_amd/variants pygpa_amd/csrc/_build
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result -Wno-unused-value -ffp-contract=fast -fno-slp-vectorize"
hipcc $F "$@" -c pygpa_amd/csrc/$tu.hip -o pygpa_amd/csrc/_build/${tu}_$name.o
objs=""
for t in gpa_sweep gpa_passb_shared gpa_sweep_ext gpa_reconstruct gpa_unwrap gpa_dft2 gpa_warp gpa_peaks gpa_api; do
  if [ $t = $tu ]; then objs="$objs pygpa_amd/csrc/_build/${tu}_$name.o"; else objs="$objs pygpa_amd/csrc/_build/$t.o"; fi
done
hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc -o pygpa_amd/variants/libgpa_$name.so $objs
echo pygpa_amd/variants/libgpa_$name.so
