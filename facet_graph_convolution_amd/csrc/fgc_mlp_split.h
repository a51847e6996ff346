// The fp32 MLP forward on the bf16 matrix pipe through three-term operand splits (fgc_mlp_bf16.hip); called from fgc_mlp.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/fgc.h"

namespace fgc {
bool mlp_split_enabled();
// which operand layouts the options in force select for the shape (include/fgc.h: fgc_mlp_layout_id), 1 ... 255; and whether an
// FGC_MLP_PACKED call's flags name another one (FGC_MLP_LAYOUT)
int mlp_layout_id(int cin, int hidden, int cout, bool bf16);
size_t mlp_split_pack_bytes(int cin, int hidden);
bool mlp_fwd_split_ok(const float* x, int cin, int hidden, int cout);   // x == NULL: the shape alone
int launch_mlp_fwd_split(const float* x, int n, int cin, int hidden, int cout, const float* W1, const float* b1, const float* W2,
                         const float* b2, float alpha, float* y, float* abs_partial, void* workspace, bool packed, hipStream_t st);
// backward (round 5): fp32 x / dx, every 1024-wide product on split operands
struct PackJob;
bool mlp_bwd_split_enabled();
bool mlp_bwd_split_ok(const float* x, const float* dx, int cin, int hidden, int cout);   // x == dx == NULL: the shape alone
size_t mlp_bwd_split_workspace_bytes(int n, int cin, int hidden);
int mlp_bwd_split_pack_jobs(const fgc_pack_extra* e, PackJob* jobs, size_t* totals);     // 3 jobs, or -1
int launch_mlp_bwd_split(const float* x, const float* dy, int n, int cin, int hidden, int cout, const float* W1, const float* b1,
                         const float* W2, float alpha, float* dx, float* dW1, float* db1, float* dW2, float* db2, void* workspace,
                         bool packed, hipStream_t st);
}  // namespace fgc
