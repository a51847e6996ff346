// Shared helpers for libfgc (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>

#include "../../include/fgc.h"

namespace fgc {

void set_error(const char* fmt, ...);

#define FGC_CHECK_ARG(cond, ...)          \
    do {                                  \
        if (!(cond)) {                    \
            fgc::set_error(__VA_ARGS__);  \
            return FGC_EINVAL;            \
        }                                 \
    } while (0)

#define FGC_CHECK_LAUNCH(what)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            fgc::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));    \
            return FGC_EHIP;                                                          \
        }                                                                             \
    } while (0)

// ---- optional per-kernel timing (fgc_profile_*): hipEvents around every launch, off by default ----
bool prof_enabled();
void prof_begin(const char* name, hipStream_t st);
void prof_end(hipStream_t st);
struct ProfScope {
    hipStream_t st;
    bool on;
    ProfScope(const char* name, hipStream_t s) : st(s), on(prof_enabled()) {
        if (on) prof_begin(name, st);
    }
    ~ProfScope() {
        if (on) prof_end(st);
    }
};
// launch with timing scope: FGC_LAUNCH("name", stream, kernel, grid, block, smem, args...)
#define FGC_LAUNCH(name, st, kernel, grid, block, smem, ...)                      \
    do {                                                                          \
        fgc::ProfScope prof__(name, st);                                          \
        hipLaunchKernelGGL(kernel, grid, block, smem, st, __VA_ARGS__);           \
    } while (0)

// ---- process-level options (fgc_set_option / fgc_get_option, include/fgc.h): which kernel form a launch takes where the
// ---- library has more than one.  No launch path reads the environment: the table is filled once from FGC_<NAME> variables
// ---- the first time an option is read (fgc_host.hip), after that only fgc_set_option changes it.
#define FGC_OPTION_LIST(X)                                                                                                     \
    X(NO_W8, 0)               /* 1: never the eight-wave conv kernels (fgc_conv_w8.hip) */                                       \
    X(NO_W8FAST, 0)           /* 1: their generic form (per-lane degree tests) also for the fast shapes */                       \
    X(W8_NT16, 1)             /* half tiles: 0 never, 1 forward + big-level data gradient, 2 data gradient always */              \
    X(W8_DATA16_MIN_N, 81920) /* nodes from which the data-gradient kernel takes half tiles */                                   \
    X(W8_DATA_SMEM_PAD, 0)    /* developer knob: extra LDS bytes per data-gradient workgroup (fewer resident workgroups) */       \
    X(NO_PAIRS, 0)            /* 1: up-convolutions in the fine form (fgc_conv_pair.hip off) */                                  \
    X(NO_NARROW, 0)           /* 1: the first layer through the tiled kernels */                                                 \
    X(NO_NARROW_MMA, 0)       /* 1: the first layer's per-node products on the vector ALU */                                      \
    X(NO_NARROW_FUSED_DS, 0)  /* 1: the first layer's s = dy lrelu'(y) / deg as a launch of its own */                            \
    X(NO_FUSED_DS, 0)         /* 1: ds_db_kernel launches instead of the d-logits prologue (fp32) */                             \
    X(NO_FUSED_DS_BF16, 0)    /* ... (bf16) */                                                                                   \
    X(NO_FUSED_DS128, 0)      /* ... for the 128-wide layers only */                                                             \
    X(NO_DS_VEC, 0)           /* 1: ds_db_kernel one element per thread */                                                       \
    X(NO_K1M, 0)              /* 1: d-logits per-edge products on the vector ALU */                                              \
    X(NO_K1DEEP, 0)           /* 1: d-logits kernel without the deep gather */                                                   \
    X(K1_NT16, 1)             /* 0: d-logits kernel on 32-node tiles */                                                          \
    X(NO_TNBF16, 0)           /* 1: bf16 weight gradients on the fp32 MFMA */                                                    \
    X(NO_TNSTREAM, 0)         /* 1: the LDS-staged weight-gradient GEMM */                                                       \
    X(TN_SLOTS, 64)           /* workgroup slots per XCD of the streaming weight-gradient GEMM */                                \
    X(TNB_WGS, 256)           /* workgroups of the bf16 weight-gradient GEMM */                                                  \
    X(NO_PROJ_STREAM, 0)      /* 1: the assignment-logit tables by the first (blocking) form of the kernel */                    \
    X(NO_MLP_SPLIT, 0)        /* 1: the MLP's 1024-wide products on the fp32 MFMA (forward and backward) */                      \
    X(NO_MLP_BWD_SPLIT, 0)    /* 1: ... the backward only */                                                                     \
    X(NO_K1_SPLIT, 0)         /* 1: the dz GEMM of the fp32 d-logits kernel (half tiles, 32 outputs) on the fp32 MFMA */                \
    X(NO_BFM, 0)              /* 1: bf16 conv kernels aggregate on the vector ALU (conv_w8_kernel<BF>), not the matrix pipe */    \
    X(K1_QS14, 0)             /* 1: fp32 half-tile d-logits kernel with a 14-slot table where degrees allow (5 workgroups / CU) */  \
    X(W8_HALF2, 0)            /* 1: fp32 half-tile forward conv with the aggregate tile in two halves (6 workgroups / CU) */
enum Opt {
#define FGC_OPT_ENUM(name, def) OPT_##name,
    FGC_OPTION_LIST(FGC_OPT_ENUM)
#undef FGC_OPT_ENUM
    OPT_COUNT
};
int64_t opt(Opt o);
// per-descriptor overrides (fgc_conv_desc.options): every entry point that takes a descriptor opens a scope for the time it
// works on that descriptor; opt() consults the calling thread's innermost scope first.  Nothing global is written.
struct OptScope {
    const void* prev_list;
    int prev_n;
    OptScope(const void* overrides /* fgc_option_override[] */, int n);
    ~OptScope();
    OptScope(const OptScope&) = delete;
    OptScope& operator=(const OptScope&) = delete;
};
#define FGC_OPT_SCOPE(d) fgc::OptScope opt_scope__((d) ? (const void*)(d)->options : nullptr, (d) ? (d)->n_options : 0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2c __attribute__((ext_vector_type(2)));

#ifdef __HIPCC__
// ---- bf16 storage (FGC_CONV_BF16: activations live in HBM as bf16, all sums stay fp32) ----------------------
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// two bf16 in one dword (low half = the element at the lower address) -> two floats
__device__ __forceinline__ f32x2c bf2_to_f2(unsigned w) {
    return f32x2c{__uint_as_float(w << 16), __uint_as_float(w & 0xFFFF0000u)};
}
// round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned f2_to_bf2(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2c{a, b}, bf16x2));
}
__device__ __forceinline__ f32x4 bf4_to_f4(u32x2 w) {
    return f32x4{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xFFFF0000u), __uint_as_float(w[1] << 16),
                 __uint_as_float(w[1] & 0xFFFF0000u)};
}
__device__ __forceinline__ u32x2 f4_to_bf4(f32x4 v) { return u32x2{f2_to_bf2(v[0], v[1]), f2_to_bf2(v[2], v[3])}; }
__device__ __forceinline__ float bf_to_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f_to_bf(float v) {
    return __builtin_bit_cast(unsigned short, (__bf16)v);
}
// element idx of an activation tensor that is fp32 or bf16 behind the same pointer
__device__ __forceinline__ float ld_act(const float* base, size_t idx, int bf16) {
    return bf16 ? bf_to_f(reinterpret_cast<const unsigned short*>(base)[idx]) : base[idx];
}
__device__ __forceinline__ void st_act(float* base, size_t idx, float v, int bf16) {
    if (bf16) reinterpret_cast<unsigned short*>(base)[idx] = f_to_bf(v);
    else base[idx] = v;
}

// Sum over each row of 16 lanes, result in every lane, on the VALU's DPP crossbar (4 adds) instead of 4 LDS-pipe
// ds_bpermute round trips: quad xor 1, quad xor 2, then mirror within 8 and within 16 (the partial sums are already
// uniform inside each quad, so the mirrors act as xor 4 / xor 8).  Lane 0 adds in the same order as an xor butterfly.
#define FGC_ROW16_SUM(v)                      \
    do {                                      \
        (v) += fgc_dpp_c<0xB1>(v);            \
        (v) += fgc_dpp_c<0x4E>(v);            \
        (v) += fgc_dpp_c<0x141>(v);           \
        (v) += fgc_dpp_c<0x140>(v);           \
    } while (0)
template <int CTRL>
__device__ __forceinline__ float fgc_dpp_c(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
#endif

// ---- geometry of the fused "edge-aggregate + MFMA" kernels ------------------------------
// A node's input row is processed KC = 4*LPN channels per pass by LPN lanes (float4 each).
// The tiled conv kernels gather 8 lanes x 4 channels per node and pass; the numbers that follow from it are compile-time
// constants in device code (LDS offsets fold into the instructions) and are mirrored by conv_geom for the host.
constexpr int KC = 32;        // channels per pass
constexpr int KPASS = 288;    // FGC_M * KC, a multiple of 16
constexpr int ZSTRIDE = 296;  // LDS row stride of the aggregate tile: == 8 mod 16, >= KPASS
// bf16 form of the aggregate tile: 288 bf16 = 576 B per row, padded to 608 B (== 32 mod 64: the same ds_read_b128
// A-fragment pattern - lane l reads row l&15 at byte offset 16*(l>>4) - stays conflict free); in bf16 elements
constexpr int ZSTRIDE_BF = 304;

struct ConvGeom {
    int cin, cout;
    int lpn;        // lanes per node: 2,4,8
    int kc;         // channels per pass = 4*lpn
    int passes;     // ceil(cin / kc)
    int kpass;      // M*kc rounded up to a multiple of 16 (MFMA k-group)
    int zstride;    // LDS row stride (floats), == 8 mod 16, >= kpass
    int npad;       // output columns padded to a multiple of 16
    int T;          // nodes per workgroup tile
};

static inline int lds_stride_for(int kpass) {
    // smallest s >= kpass with s % 16 == 8: the ds_read_b128 A-fragment pattern
    // (lane l -> row l&15, float offset 4*(l>>4)) is then bank-conflict free (brute-forced
    // against the gfx950 b128 lane groups for every such stride < 2000).
    int s = (kpass / 16) * 16 + 8;
    while (s < kpass) s += 16;
    return s;
}

static inline ConvGeom conv_geom(int cin, int cout) {
    ConvGeom g;
    g.cin = cin;
    g.cout = cout;
    // 8 lanes x 4 channels per node and pass for every width the tiled kernels see (narrow inputs, cin <= 8, take
    // fgc_conv_narrow.hip): keeps all 256 threads, the 32-channel z tile and the matrix-core kernels in play
    const int lpn = 8;
    g.lpn = lpn;
    g.kc = 4 * lpn;
    g.passes = (cin + g.kc - 1) / g.kc;
    g.kpass = (FGC_M * g.kc + 15) / 16 * 16;
    g.zstride = lds_stride_for(g.kpass);
    g.npad = (cout + 15) / 16 * 16;
    g.T = 32;
    static_assert(KC == 32 && KPASS == (FGC_M * KC + 15) / 16 * 16 && ZSTRIDE >= KPASS && ZSTRIDE % 16 == 8, "geometry");
    return g;
}

// leaky ReLU for 0 <= alpha <= 1 as max(v, alpha v) in two instructions.  fmaxf (and v_med3 with +inf, which the
// compiler folds back into it) puts a canonicalising v_max_f32(v, v) in front of the maximum because it cannot see that
// an MFMA result is never a signalling NaN: a third vector instruction per hidden activation next to the fp32 MFMAs.
// Same value as fmaxf(v, alpha * v) for every non-NaN input.
__device__ __forceinline__ float lrelu01(float v, float alpha) {
    const float av = alpha * v;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(av));
    return r;
}

// d lrelu / d pre = 1 (pre > 0), alpha (pre < 0), 0 (pre == 0: the relu gradient TensorFlow uses) in two instructions
// instead of two compares and two selects: pre * 2^126 saturates a median at 1 or at -alpha for every normal number and
// stays 0 for 0; the sign is dropped by the |.| operand modifier of the multiply that consumes it.  (A denormal pre -
// below 1.2e-38 in magnitude - gives a slope between the two; 0 <= alpha <= 1.)
__device__ __forceinline__ float lrelu01_slope(float pre, float alpha) {
    return __builtin_fabsf(__builtin_amdgcn_fmed3f(pre * 0x1p126f, -alpha, 1.0f));
}

}  // namespace fgc
