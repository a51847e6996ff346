// Epilogue descriptors shared by the tiled conv kernels, and the eight-wave kernels' entry points (fgc_conv_w8.hip).
#pragma once
#include "fgc_conv_core.h"

namespace fgc {

struct FwdEpilogue {
    const float* bias;
    int bias_mask;
    int act;
    float alpha;
    float* y;
    float* y_pool;
};

struct DataEpilogue {
    const float* dl;      // [nnz, 12]
    float* dag;           // reads 0..8 (da), writes 12..20 (dg)
    float* r;             // [n, rld]: 9*cout aggregate columns, then da (0..8) | dg (12..20) of the node at column 9*cout
    int rld;              // 9*cout + 24, or that rounded up to whole 128-byte lines (FGC_CONV_R_PAD: conv_r_ld)
    const float* u;       // [9, cin]
    const float* v;       // [9, cin]
    int cin, c0f, c1f;    // forward input split
    int shiftf;           // forward input shift (0 / 2)
    float* dx0;
    float* dx1;
    int acc0, acc1;
};

// row stride of r in elements (fp32 floats / bf16 halves): 9*cout + 24, padded to a whole number of 128-byte lines with
// FGC_CONV_R_PAD in fgc_conv_bwd_io.flags (rows of 1 248 bytes start at 0 / 96 / 64 / 32 bytes into a line, so three of four
// 128-byte pieces a node's 16 lanes store straddle two lines)
static inline int conv_r_ld(int cout, int io_flags, bool bf16) {
    const int pl = FGC_M * cout + 24, q = bf16 ? 64 : 32;
    return (io_flags & FGC_CONV_R_PAD) ? (pl + q - 1) / q * q : pl;
}
// ... of THIS io: the stride the caller stated with the buffer (fgc_conv_bwd_io.r_ld, checked by io_r_ld_ok), else the one the
// call's flags imply.  The writer (stage 4) and the readers (stage 8, fgc_conv_bwd_reduce) all come through here.
static inline int io_r_ld(const fgc_conv_bwd_io* io, int cout, bool bf16) {
    return io->r_ld > 0 ? io->r_ld : conv_r_ld(cout, io->flags, bf16);
}
static inline bool io_r_ld_ok(const fgc_conv_bwd_io* io, int cout, bool bf16) {
    const int pl = FGC_M * cout + 24;
    return io->r_ld == 0 || (io->r_ld >= pl && (io->r_ld - pl) % (bf16 ? 8 : 4) == 0);
}

// which packed-operand layouts the option values in force select for the descriptor (include/fgc.h: fgc_conv_layout_id);
// defined beside the pack code in fgc_conv_bwd.hip
uint64_t conv_layout_id(const fgc_conv_desc* d);

bool w8_supported(const CoreParams& p, int max_deg);
// max_deg: the largest degree of the gathered graph (<= KMAX); <= 16 selects the 16-slot form of the fast kernel
// bf16 = FGC_CONV_BF16 storage (needs w8_bf16_supported)
bool w8_bf16_supported(const CoreParams& p, int max_deg);
int launch_fwd_w8(const CoreParams& p, const FwdEpilogue& ep, size_t smem, int max_deg, hipStream_t st, bool bf16 = false);
int launch_data_w8(const CoreParams& p, const DataEpilogue& ep, size_t smem, int max_deg, hipStream_t st, bool bf16 = false);
// pair form: rows of the gathered operand by edge id (p.eid)
bool w8_erow_supported(const CoreParams& p, int max_deg);
int launch_data_w8_erow(const CoreParams& p, const DataEpilogue& ep, size_t smem, int max_deg, hipStream_t st, bool bf16 = false);

// bf16 storage, degrees <= 16: the aggregation on the bf16 matrix pipe (fgc_conv_bfm.hip).  half = 16-node workgroups.
bool bfm_supported(const CoreParams& p, int max_deg, bool data, int cin_fwd);
int launch_fwd_bfm(const CoreParams& p, const FwdEpilogue& ep, bool half, hipStream_t st);
int launch_data_bfm(const CoreParams& p, const DataEpilogue& ep, bool half, bool erow, hipStream_t st);

}  // namespace fgc
