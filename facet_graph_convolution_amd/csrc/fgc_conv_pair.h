// Pair form of a graph convolution over a 4x-upsampled coarse tensor (fgc_conv_desc.pair_rowptr; fgc_conv_pair.hip).
#pragma once
#include "fgc_common.h"

namespace fgc {

// does this descriptor run in the pair form (a function of the descriptor alone: workspace plans depend on it)
bool pairs_ok(const fgc_conv_desc* d);
bool pairs_graph_allowed(int64_t rows, int64_t n_pairs, int max_in_deg, int cout);   // fgc_conv_pairs_allowed
// blocks (of four fine nodes) per workgroup of pair_fwd_kernel / pair_bwd_logits_kernel: one db / dc partial each
int pair_blocks_per_wg(int cout);
static inline int pair_num_wgs(const fgc_conv_desc* d) {
    const int bpg = pair_blocks_per_wg(d->cout);
    return ((d->n >> 2) + bpg - 1) / bpg;
}
// h = W0 x and the logit table for source rows [0, rows), then y for all blocks.  FGC_CONV_BF16: the forward workspace
// holds a bf16 copy of W0 (M * cout * cin elements; written here unless FGC_CONV_PACKED, else by fgc_conv_pack)
int launch_pair_fwd(const fgc_conv_desc* d, float* ag, float* y, void* workspace, hipStream_t st);
// s, db partials, per-pair dt and dl, da, dc partials
int launch_pair_bwd_logits(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, float* db_part, float* dc_part,
                           hipStream_t st);

}  // namespace fgc
