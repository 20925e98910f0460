#!/bin/bash
# tools/ab_build.sh NAME "-DFLAG=.. ..." : build an instrumented variant of the library into tools/ab/libT360_NAME.so
# (git-ignored) for same-call A/B runs on the GPU box:  T360_LIB=$PWD/tools/ab/libT360_NAME.so tools/sweep.sh ...
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
S=$R/transform360_amd/csrc
O=$R/tools/ab/$NAME
mkdir -p "$O"
# NOINSTR=1: a variant of the SHIPPED configuration (no tuning switches) -- bench.py then reports its numbers like the library's
FL=(-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -I$R/include -I$S $([ "${NOINSTR:-0}" = 1 ] || echo -DT360_INSTRUMENT) "$@")
for f in t360_mapgen.hip t360_remap.hip t360_remap_tiled.hip t360_lowpass.hip t360_resize.hip; do
  /opt/rocm/bin/hipcc "${FL[@]}" -c $S/$f -o $O/${f%.hip}.o &
done
for f in t360_filtercfg.cpp t360_plan.cpp t360_hoststage.cpp t360_transform.cpp t360_capi.cpp; do
  /opt/rocm/bin/hipcc "${FL[@]}" -x hip -c $S/$f -o $O/${f%.cpp}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -Wl,-rpath,/opt/rocm/lib -o $R/tools/ab/libT360_$NAME.so $O/*.o
echo built $R/tools/ab/libT360_$NAME.so
