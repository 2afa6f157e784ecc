#!/bin/bash
# usage: tools/build_variant.sh NAME "EXTRA HIPCC FLAGS" [file.hip ...]   -> waterlily.jl_amd/libwlhip_NAME.so
# Rebuilds the named sources (default: wl_convt.hip) with the extra flags and links them with the default objects: an A/B
# library for `WLHIP_LIB=waterlily.jl_amd/libwlhip_NAME.so python bench.py ...` in the same gpurun call as the default one.
set -e
NAME=$1; shift
EXTRA=$1; shift
FILES=${@:-wl_convt.hip}
cd "$(dirname "$0")/../waterlily.jl_amd/csrc"
make -j8 >/dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-result"
OBJS=""
for f in wl_flow wl_poisson wl_capi wl_sim wl_comm wl_fused wl_fused2 wl_convz wl_convm wl_convt wl_convf wl_resjac; do
  if [[ " $FILES " == *" $f.hip "* ]]; then
    /opt/rocm/bin/hipcc $FLAGS $EXTRA -c $f.hip -o /tmp/${f}_$NAME.o
    OBJS="$OBJS /tmp/${f}_$NAME.o"
  else
    OBJS="$OBJS $f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libwlhip_$NAME.so $OBJS -ldl
echo built ../libwlhip_$NAME.so
