#!/bin/bash
# build 3pre_amd/lib/libpre3_<tag>.so from the working tree with extra -D flags on pre3_update.hip (same-box A/B of kernel variants):
#   tools/build_variant.sh nb4 -DB3_NBUF=4 ;  PRE3_LIB=3pre_amd/lib/libpre3_nb4.so python tools/k9ab.py
set -e
tag=$1; shift
cd 3pre_amd/csrc
make >/dev/null
/opt/rocm/bin/hipcc "$@" -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -Wall -Wno-unused-function -ffp-contract=on -mllvm -pragma-unroll-threshold=200000 -c pre3_update.hip -o /tmp/pre3_update_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-rpath,/opt/rocm/lib -o ../lib/libpre3_$tag.so pre3_api.o pre3_geom.o pre3_match.o pre3_map.o pre3_vo.o /tmp/pre3_update_$tag.o
