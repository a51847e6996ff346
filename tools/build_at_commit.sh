#!/bin/bash
# Developer helper: a complete build of libfgc.so from the kernel sources of another commit, into gpurun_variants/libfgc_<tag>.so
# (travels to the GPU box; git-ignored), for same-box A/B runs through FGC_LIB (tools/bench_variants.sh).  Every object is
# compiled from that commit's csrc/ and include/: objects of two commits must not be mixed (the option enum, struct layouts).
# usage: tools/build_at_commit.sh <commit> <tag>      (FGC_DEV_PARTIAL=1 when the old library lacks newer entry points)
set -e
commit=$1; tag=$2
root="$(cd "$(dirname "$0")/.." && pwd)"
tmp=$(mktemp -d)
git -C "$root" archive "$commit" facet_graph_convolution_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/facet_graph_convolution_amd/csrc" -j8 > "$tmp/build.log" 2>&1 || { tail -5 "$tmp/build.log"; exit 1; }
mkdir -p "$root/gpurun_variants"
cp "$tmp/facet_graph_convolution_amd/csrc/libfgc.so" "$root/gpurun_variants/libfgc_$tag.so"
rm -rf "$tmp"
echo "gpurun_variants/libfgc_$tag.so"
