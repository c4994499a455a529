#!/bin/bash
# Experimental library build with extra compiler flags, for same-box A/B runs (TTTS_LIB=...):
# usage: tools/build_variant.sh <out.so> [-DFLAG ...]     (objects of untouched files are reused from the main build)
out=$1; shift
d=transformertts_amd/build_variant; mkdir -p $d
objs=""
for src in transformertts_amd/csrc/*.hip; do
  o=$d/$(basename $src).o
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $src -o $o &
  objs="$objs $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs
