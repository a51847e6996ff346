// Developer probe (not part of the library): cost of one vector-memory instruction on gfx950 as a function of its width and
// of how many different 128-byte lines its 64 lanes touch, for loads that hit in L2 / L1 (a 1 MB working set per launch).
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_variants/vmem_rate_probe tools/probes/vmem_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// W = bytes per lane (4, 8, 16); LPR = lanes per row (a row = LPR * W contiguous bytes); rows are picked by a per-lane-group
// pseudo-random index that changes every iteration, inside a window of `rows` rows of `rowbytes` bytes.
template <int W, int LPR>
__global__ __launch_bounds__(256) void k_load(const char* base, int rows, int rowbytes, int iters, float* out) {
    const int lane = threadIdx.x & 63, grp = (threadIdx.x + blockIdx.x * 256) / LPR, within = lane % LPR;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, -1, 0x00020000);
    float acc = 0.f;
    unsigned h = (unsigned)grp * 2654435761u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u;
            const unsigned row = (h >> 8) % (unsigned)rows;
            const unsigned off = row * (unsigned)rowbytes + (unsigned)(within * W);
            if constexpr (W == 4) acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
            else if constexpr (W == 8) { const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0)); acc += v[0] + v[1]; }
            else { const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0)); acc += v[0] + v[3]; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int W>
__global__ __launch_bounds__(256) void k_store(char* base, int rows, int rowbytes, int iters, int lpr) {
    const int lane = threadIdx.x & 63, grp = (threadIdx.x + blockIdx.x * 256) / lpr, within = lane % lpr;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, -1, 0x00020000);
    unsigned h = (unsigned)grp * 2654435761u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u;
            const unsigned row = (h >> 8) % (unsigned)rows;
            const unsigned off = row * (unsigned)rowbytes + (unsigned)(within * W);
            if constexpr (W == 4) __builtin_amdgcn_raw_buffer_store_b32(h, rs, off, 0, 0);
            else if constexpr (W == 8) __builtin_amdgcn_raw_buffer_store_b64(u32x2{h, 2u}, rs, off, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(u32x4{h, 2u, 3u, 4u}, rs, off, 0, 0);
        }
    }
}

int main() {
    const size_t bytes = 64u << 20;
    char* buf;
    float* out;
    hipMalloc(&buf, bytes);
    hipMemset(buf, 0, bytes);
    hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 400, grid = 256 * 8;    // 8 workgroups of 4 waves per CU
    auto time = [&](auto launch) {
        launch(4);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch(iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // instructions per CU = 8 workgroups * 4 waves * iters * 8
        return (double)ms * 1e6 / (8.0 * 4 * iters * 8);
    };
    printf("ns per wave-level instruction and CU (2.4 GHz: 16 cycles = 6.7 ns; 64 B/clk = 1 KB per 6.7 ns)\n");
    printf("%-64s %10s %12s\n", "pattern", "ns / instr", "GB/s per CU");
    struct C { const char* name; int w, lpr, rows, rowbytes; };
    // working set: rows * rowbytes (1 MB: L2-resident, too large for the 32 KB L1) or 16 KB (L1-resident)
    std::vector<C> cases = {
        {"load b32, 64 lanes one 256-B run, 1 MB set", 4, 64, 4096, 256},
        {"load b64, 64 lanes one 512-B run, 1 MB set", 8, 64, 2048, 512},
        {"load b128, 64 lanes one 1-KB run, 1 MB set", 16, 64, 1024, 1024},
        {"load b128, 64 lanes one 1-KB run, 16 KB set (L1)", 16, 64, 16, 1024},
        {"load b32, 64 lanes one 256-B run, 16 KB set (L1)", 4, 64, 64, 256},
        {"load b64, 16 lanes per 128-B row (4 rows / instr), 1 MB set", 8, 16, 8192, 128},
        {"load b128, 8 lanes per 128-B row (8 rows / instr), 1 MB set", 16, 8, 8192, 128},
        {"load b32, 16 lanes per 64-B row (4 rows / instr), 1 MB set", 4, 16, 16384, 64},
        {"load b128, 1 lane per 64-B row (64 rows / instr), 1 MB set", 16, 1, 16384, 64},
        {"load b32, 1 lane per 64-B row (64 rows / instr), 1 MB set", 4, 1, 16384, 64},
        {"load b128, 4 lanes per 64-B row (16 rows / instr), 1 MB set", 16, 4, 16384, 64},
        {"load b64, 16 lanes per 128-B row, 16 KB set (L1)", 8, 16, 128, 128},
        {"load b128, 8 lanes per 128-B row, 16 KB set (L1)", 16, 8, 128, 128},
        {"load b128, 1 lane per 64-B row, 16 KB set (L1)", 16, 1, 256, 64},
    };
    for (auto& c : cases) {
        double ns = 0;
#define RUN(W_, L_) ns = time([&](int it) { k_load<W_, L_><<<grid, 256>>>(buf, c.rows, c.rowbytes, it, out); })
        if (c.w == 4 && c.lpr == 64) RUN(4, 64); else if (c.w == 8 && c.lpr == 64) RUN(8, 64); else if (c.w == 16 && c.lpr == 64) RUN(16, 64);
        else if (c.w == 8 && c.lpr == 16) RUN(8, 16); else if (c.w == 16 && c.lpr == 8) RUN(16, 8); else if (c.w == 4 && c.lpr == 16) RUN(4, 16);
        else if (c.w == 16 && c.lpr == 1) RUN(16, 1); else if (c.w == 4 && c.lpr == 1) RUN(4, 1); else if (c.w == 16 && c.lpr == 4) RUN(16, 4);
        printf("%-64s %10.2f %12.1f\n", c.name, ns, 64.0 * c.w / ns);
    }
    struct S { const char* name; int w, lpr, rows, rowbytes; };
    std::vector<S> st = {
        {"store b32, 64 lanes one 256-B run, 4 MB set", 4, 64, 16384, 256},
        {"store b64, 64 lanes one 512-B run, 4 MB set", 8, 64, 8192, 512},
        {"store b128, 64 lanes one 1-KB run, 4 MB set", 16, 64, 4096, 1024},
        {"store b64, 16 lanes per 128-B row, 4 MB set", 8, 16, 32768, 128},
        {"store b128, 8 lanes per 128-B row, 4 MB set", 16, 8, 32768, 128},
        {"store b128, 1 lane per 64-B row, 4 MB set", 16, 1, 65536, 64},
        {"store b32, 1 lane per 64-B row, 4 MB set", 4, 1, 65536, 64},
    };
    for (auto& c : st) {
        double ns = 0;
        if (c.w == 4) ns = time([&](int it) { k_store<4><<<grid, 256>>>(buf, c.rows, c.rowbytes, it, c.lpr); });
        else if (c.w == 8) ns = time([&](int it) { k_store<8><<<grid, 256>>>(buf, c.rows, c.rowbytes, it, c.lpr); });
        else ns = time([&](int it) { k_store<16><<<grid, 256>>>(buf, c.rows, c.rowbytes, it, c.lpr); });
        printf("%-64s %10.2f %12.1f\n", c.name, ns, 64.0 * c.w / ns);
    }
    return 0;
}
