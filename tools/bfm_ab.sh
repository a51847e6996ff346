#!/bin/bash
# Developer helper (GPU box): bf16 conv kernels, matrix-pipe aggregation (in-tree library, variants) against the vector form
# (FGC_NO_BFM=1), per-kernel times of the bf16 100k step.  usage: tools/bfm_ab.sh [variant.so ...]
P="fwd:dconv1/conv_w8 bwd:dconv1/conv_w8_kernel<data> fwd:dconv2/conv_w8 bwd:dconv2/conv_w8_kernel<data> bwd:conv2/conv_w8_kernel<data> fwd:dconv3/conv_w8 bwd:dconv3/conv_w8_kernel<data>"
bash tools/ko_bench.sh "--dtype bf16" "$P" - "$@" -
echo "vector form (FGC_NO_BFM=1):"
FGC_NO_BFM=1 bash tools/ko_bench.sh "--dtype bf16" "$P" - -
