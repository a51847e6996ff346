#!/bin/bash
# Developer helper: an alternative build of libfgc.so with extra compiler flags, into gpurun_variants/libfgc_<tag>.so (in-tree so that
# it travels to the GPU box; git-ignored).  usage: tools/build_variant.sh <tag> "<flags, e.g. -DFGC_PT_RT=1>" [file.hip ...]
# Only the listed sources get the flags (default: all); the rest link from the in-tree objects.
set -e
tag=$1; flags=$2; shift 2
cd "$(dirname "$0")/../facet_graph_convolution_amd/csrc"
mkdir -p ../../gpurun_variants/obj_$tag
srcs=${@:-$(ls fgc_*.hip | tr "\n" " ")}
objs=""
for f in $(ls fgc_*.hip); do
  o=${f%.hip}.o
  if echo " $(echo $srcs) " | grep -q " $f "; then
    # (the Makefile's per-file flags)
    pf=""; case " fgc_mlp_bf16.hip fgc_mlp.hip fgc_conv_bwd.hip fgc_conv_narrow.hip " in *" $f "*) pf="-fno-slp-vectorize";; esac
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-unused-result $pf $flags -c $f -o ../../gpurun_variants/obj_$tag/$o &
    objs="$objs ../../gpurun_variants/obj_$tag/$o"
  else
    objs="$objs $o"
  fi
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_variants/libfgc_$tag.so $objs
echo "gpurun_variants/libfgc_$tag.so"
