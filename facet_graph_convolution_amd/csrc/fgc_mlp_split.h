// The fp32 MLP forward on the bf16 matrix pipe through three-term operand splits (fgc_mlp_bf16.hip); called from fgc_mlp.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace fgc {
bool mlp_split_enabled();
size_t mlp_split_pack_bytes(int cin, int hidden);
bool mlp_fwd_split_ok(const float* x, int cin, int hidden, int cout);   // x == NULL: the shape alone
int launch_mlp_fwd_split(const float* x, int n, int cin, int hidden, int cout, const float* W1, const float* b1, const float* W2,
                         const float* b2, float alpha, float* y, float* abs_partial, void* workspace, bool packed, hipStream_t st);
}  // namespace fgc
