// Narrow-input graph convolution (cin <= 8): the network's first layer.  See fgc_conv_narrow.hip.
#pragma once
#include "fgc_conv_core.h"

namespace fgc {

bool narrow_supported(const fgc_conv_desc* d);
// forward (after the logit table `ag` has been computed); honours d->tile_list
// zsave (may be NULL): [n, narrow_zld(cin)] receives the aggregates for the backward pass (FGC_CONV_SAVE_Z)
// out_bf16: y and y_pool are bf16 tensors (FGC_CONV_BF16); the input x0 and everything else stay fp32
int launch_narrow_fwd(const fgc_conv_desc* d, const float* ag, float* y, float* y_pool, float* zsave, hipStream_t st,
                      bool out_bf16 = false);

// backward of a narrow FIRST layer (io->dx0 == NULL).  `scratch` holds narrow_bwd_floats(d) floats and must survive
// from the stage-2 call to the stage-8 call.
int narrow_zld(int cin);
int narrow_splits(const fgc_conv_desc* d);
size_t narrow_bwd_floats(const fgc_conv_desc* d);
int narrow_bwd_logits(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, float* scratch, float* db_part, hipStream_t st);
// does the stage-2 kernel compute s = dy * lrelu'(y) / deg (+ the pooled gradient) and the db partials itself (stage 1 is then
// empty), and how many db partials does the layer leave (nb_db = the count of stage 1's own launch)
bool narrow_fuses_ds(const fgc_conv_desc* d, const fgc_conv_bwd_io* io);
int narrow_db_partials(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, int nb_db);
// parts: 1 = weight-gradient GEMM, 2 = the fixed-order sums (they write dW0 / du / dv / dc / db); jobs_out (may be NULL)
// receives the NARROW_RED_JOBS reduction jobs so that a caller can run them together with other layers'
struct RedJob;
constexpr int NARROW_RED_JOBS = 5;
int narrow_bwd_params(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, float* scratch, const float* db_part,
                      int nb_db, int parts, RedJob* jobs_out, hipStream_t st);

void narrow_tn_operands(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, float* scratch, const float** A, int* zld_out,
                        float** slab_out, int* rps_out);

// streaming TN GEMM (fgc_conv_bwd.hip): slab[split][P][c0] = A[rows of the split, P]^T x0[rows of the split, c0]
int tn_balanced_splits(int desired, int maxs, int rows);   // slab count of a weight-gradient GEMM, XCD-balanced
int launch_gemm_tn_stream(const char* tag, const float* A, int lda, int P, const float* x0, int c0, int rows,
                          int rows_per_split, int nsplits, float* slab, hipStream_t st);

}  // namespace fgc
