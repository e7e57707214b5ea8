#!/bin/bash
# usage: tools/variant.sh <name> <tu> <extra hipcc flags...>      (tu without .hip, e.g. gpa_unwrap_cols)
# builds pygpa_amd/variants/libgpa_<name>.so with one translation unit recompiled with the extra flags
# (performance experiments; select with GPA_HIP_LIB=<path> at run time).  The other objects are the shipped build's
# (python -m pygpa_amd.build first); the variant's object lives under _build/variants/, which build() leaves alone.
set -e
name=$1; tu=$2; shift 2
cd "$(dirname "$0")/.."
mkdir -p pygpa_amd/variants pygpa_amd/csrc/_build/variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result -Wno-unused-value -ffp-contract=fast -fno-slp-vectorize"
hipcc $F "$@" -c pygpa_amd/csrc/$tu.hip -o pygpa_amd/csrc/_build/variants/${tu}_$name.o
objs=""
for t in $(python3 -c "from pygpa_amd.build import SOURCES; print(' '.join(s[:-4] for s in SOURCES))"); do
  if [ $t = $tu ]; then objs="$objs pygpa_amd/csrc/_build/variants/${tu}_$name.o"; else objs="$objs pygpa_amd/csrc/_build/$t.o"; fi
done
hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc -o pygpa_amd/variants/libgpa_$name.so $objs
echo pygpa_amd/variants/libgpa_$name.so
