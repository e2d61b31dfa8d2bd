#!/bin/bash
# Developer tool: a variant of the library that differs from the built one in ONE translation unit compiled with extra flags
# (20 s instead of a whole devlib build):   tools/quick_variant.sh <tag> <unit, e.g. tu_stft_f64_p0> [flags...]
#   -> tools/_build/libssrhip_<tag>.so  (SSR_DEV_LIB for tools/exp_*.py)
TAG=$1; U=$2; shift 2
R="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p $R/tools/_build/obj_q_$TAG
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $R/ssr_eval_amd/csrc/$U.hip -o $R/tools/_build/obj_q_$TAG/$U.o || exit 1
OBJS=""
for o in $R/ssr_eval_amd/csrc/_obj/tu_*.o; do
  b=$(basename $o)
  if [ "$b" = "$U.o" ]; then OBJS="$OBJS $R/tools/_build/obj_q_$TAG/$U.o"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_build/libssrhip_$TAG.so $OBJS -ldl && echo $R/tools/_build/libssrhip_$TAG.so
