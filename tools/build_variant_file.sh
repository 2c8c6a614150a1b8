#!/bin/bash
# build 3pre_amd/lib/libpre3_<tag>.so with extra -D flags on ONE source file (same-box A/B of kernel variants):
#   tools/build_variant_file.sh pre3_match nomfma -DI8Q_EXP=1 ; PRE3_LIB=3pre_amd/lib/libpre3_nomfma.so python tools/match_ab.py child
set -e
src=$1; tag=$2; shift; shift
cd 3pre_amd/csrc
make >/dev/null
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function -ffp-contract=on -mllvm -pragma-unroll-threshold=200000 -c $src.hip -o /tmp/${src}_$tag.o
objs=""; for f in pre3_api pre3_geom pre3_update pre3_cholp pre3_match pre3_map pre3_vo pre3_comm; do if [ $f = $src ]; then objs="$objs /tmp/${src}_$tag.o"; else objs="$objs $f.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib -o ../lib/libpre3_$tag.so $objs -ldl
