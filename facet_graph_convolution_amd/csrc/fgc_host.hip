// Host-side pieces of libfgc: error text, K-list <-> CSR, transposed CSR.
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "fgc_common.h"

namespace fgc {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace fgc

// ---- per-kernel timing -------------------------------------------------------------------
namespace fgc {
struct ProfRec {
    std::string name;
    hipEvent_t a, b;
};
static char g_tag[48] = "";
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static hipEvent_t g_cur_start;
static const char* g_cur_name;
bool prof_enabled() { return g_prof_on; }
void prof_begin(const char* name, hipStream_t st) {
    hipEventCreate(&g_cur_start);
    hipEventRecord(g_cur_start, st);
    g_cur_name = name;
}
void prof_end(hipStream_t st) {
    ProfRec r;
    r.name = std::string(g_tag) + "/" + g_cur_name;
    r.a = g_cur_start;
    hipEventCreate(&r.b);
    hipEventRecord(r.b, st);
    g_prof.push_back(r);
}
}  // namespace fgc

// label prepended to the kernel names recorded from now on (e.g. the layer being run)
extern "C" int fgc_profile_tag(const char* tag) {
    snprintf(fgc::g_tag, sizeof(fgc::g_tag), "%s", tag ? tag : "");
    return FGC_OK;
}
extern "C" int fgc_profile_enable(int on) {
    fgc::g_prof_on = on != 0;
    return FGC_OK;
}
// Synchronises, then writes "name count total_ms\n" lines (aggregated per kernel name) into buf and clears the
// recorded events.  Returns the number of bytes written (excluding the terminator), or a negative error.
extern "C" int fgc_profile_collect(char* buf, int32_t buf_bytes) {
    FGC_CHECK_ARG(buf && buf_bytes > 0, "fgc_profile_collect: bad buffer");
    struct Agg {
        std::string name;
        int count;
        double ms;
    };
    std::vector<Agg> agg;
    for (auto& r : fgc::g_prof) {
        hipEventSynchronize(r.b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.a, r.b);
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
        bool found = false;
        for (auto& a : agg)
            if (a.name == r.name) {
                a.count++;
                a.ms += ms;
                found = true;
                break;
            }
        if (!found) agg.push_back({r.name, 1, (double)ms});
    }
    fgc::g_prof.clear();
    int off = 0;
    buf[0] = 0;
    for (auto& a : agg) {
        std::string nm = a.name;
        for (auto& ch : nm)
            if (ch == ' ') ch = '_';
        const int w = snprintf(buf + off, buf_bytes - off, "%s %d %.6f\n", nm.c_str(), a.count, a.ms);
        if (w < 0 || w >= buf_bytes - off) break;
        off += w;
    }
    return off;
}

// ---- process-level options ---------------------------------------------------------------------
namespace fgc {
struct OptEntry {
    const char* name;
    int64_t def;
};
static const OptEntry g_opt_table[OPT_COUNT] = {
#define FGC_OPT_ROW(name, def) {#name, def},
    FGC_OPTION_LIST(FGC_OPT_ROW)
#undef FGC_OPT_ROW
};
static std::atomic<int64_t> g_opt[OPT_COUNT];
static std::once_flag g_opt_once;
// the ONE place the library reads its environment: FGC_<NAME>, once per process, as the initial value of option NAME
static void opt_init() {
    for (int i = 0; i < OPT_COUNT; ++i) {
        char var[64];
        snprintf(var, sizeof(var), "FGC_%s", g_opt_table[i].name);
        const char* e = getenv(var);
        g_opt[i].store((e && *e) ? strtoll(e, nullptr, 10) : g_opt_table[i].def, std::memory_order_relaxed);
    }
}
static thread_local const fgc_option_override* tl_over = nullptr;
static thread_local int tl_over_n = 0;
OptScope::OptScope(const void* overrides, int n) : prev_list(tl_over), prev_n(tl_over_n) {
    if (overrides && n > 0) {      // (a descriptor without a list leaves an enclosing scope in force)
        tl_over = static_cast<const fgc_option_override*>(overrides);
        tl_over_n = n;
    }
}
OptScope::~OptScope() {
    tl_over = static_cast<const fgc_option_override*>(prev_list);
    tl_over_n = prev_n;
}
int64_t opt(Opt o) {
    for (int i = 0; i < tl_over_n; ++i)
        if (tl_over[i].index == (int)o) return tl_over[i].value;
    std::call_once(g_opt_once, opt_init);
    return g_opt[o].load(std::memory_order_relaxed);
}
static int opt_index(const char* name) {
    if (!name) return -1;
    if (!strncmp(name, "FGC_", 4)) name += 4;
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(name, g_opt_table[i].name)) return i;
    return -1;
}
}  // namespace fgc

extern "C" int fgc_set_option(const char* name, int64_t value) {
    const int i = fgc::opt_index(name);
    FGC_CHECK_ARG(i >= 0, "fgc_set_option: unknown option '%s'", name ? name : "(null)");
    std::call_once(fgc::g_opt_once, fgc::opt_init);
    fgc::g_opt[i].store(value, std::memory_order_relaxed);
    return FGC_OK;
}
extern "C" int fgc_get_option(const char* name, int64_t* value) {
    const int i = fgc::opt_index(name);
    FGC_CHECK_ARG(i >= 0 && value, "fgc_get_option: unknown option '%s'", name ? name : "(null)");
    *value = fgc::opt((fgc::Opt)i);
    return FGC_OK;
}
extern "C" int32_t fgc_option_count(void) { return fgc::OPT_COUNT; }
extern "C" const char* fgc_option_name(int32_t index) {
    return index >= 0 && index < fgc::OPT_COUNT ? fgc::g_opt_table[index].name : nullptr;
}

extern "C" const char* fgc_last_error(void) { return fgc::g_err; }
extern "C" int fgc_version(void) { return FGC_ABI_VERSION; }
extern "C" size_t fgc_struct_size(int32_t which) {
    return which == 0 ? sizeof(fgc_conv_desc) : (which == 1 ? sizeof(fgc_conv_bwd_io) : (which == 2 ? sizeof(fgc_pack_extra) : 0));
}

// K-list -> CSR, slot order and duplicates preserved (model.py:380-405 gathers every non-zero slot;
// model.py:436 counts them).
extern "C" int fgc_csr_from_klist(const int32_t* adj_h, int32_t n, int32_t K, int32_t* rowptr_h, int32_t* col_h,
                                  int64_t* nnz_out) {
    FGC_CHECK_ARG(adj_h && rowptr_h && n >= 0 && K > 0, "fgc_csr_from_klist: bad arguments");
    int64_t nnz = 0;
    rowptr_h[0] = 0;
    for (int32_t i = 0; i < n; ++i) {
        const int32_t* row = adj_h + (size_t)i * K;
        for (int32_t k = 0; k < K; ++k) {
            const int32_t a = row[k];
            if (a == 0) continue;
            FGC_CHECK_ARG(a >= 1 && a <= n, "fgc_csr_from_klist: adj[%d,%d]=%d outside [0,%d]", i, k, a, n);
            if (col_h) col_h[nnz] = a - 1;
            ++nnz;
        }
        FGC_CHECK_ARG(nnz <= INT32_MAX, "fgc_csr_from_klist: more than 2^31 edges");
        rowptr_h[i + 1] = (int32_t)nnz;
    }
    if (nnz_out) *nnz_out = nnz;
    return FGC_OK;
}

extern "C" int fgc_klist_from_csr(const int32_t* rowptr_h, const int32_t* col_h, int32_t n, int32_t K,
                                  int32_t* adj_h) {
    FGC_CHECK_ARG(rowptr_h && col_h && adj_h && n >= 0 && K > 0, "fgc_klist_from_csr: bad arguments");
    for (int32_t i = 0; i < n; ++i) {
        const int32_t d = rowptr_h[i + 1] - rowptr_h[i];
        FGC_CHECK_ARG(d >= 0 && d <= K, "fgc_klist_from_csr: row %d has %d entries, K=%d", i, d, K);
        int32_t* row = adj_h + (size_t)i * K;
        for (int32_t k = 0; k < K; ++k) row[k] = k < d ? col_h[rowptr_h[i] + k] + 1 : 0;
    }
    return FGC_OK;
}

extern "C" int fgc_csr_transpose(const int32_t* rowptr_h, const int32_t* col_h, int32_t n, int32_t* trowptr_h,
                                 int32_t* tcol_h, int32_t* tedge_h) {
    FGC_CHECK_ARG(rowptr_h && col_h && trowptr_h && tcol_h && tedge_h && n >= 0, "fgc_csr_transpose: bad arguments");
    const int32_t nnz = rowptr_h[n];
    memset(trowptr_h, 0, sizeof(int32_t) * ((size_t)n + 1));
    for (int32_t e = 0; e < nnz; ++e) {
        const int32_t j = col_h[e];
        FGC_CHECK_ARG(j >= 0 && j < n, "fgc_csr_transpose: col[%d]=%d outside [0,%d)", e, j, n);
        trowptr_h[j + 1]++;
    }
    for (int32_t j = 0; j < n; ++j) trowptr_h[j + 1] += trowptr_h[j];
    std::vector<int32_t> fill(trowptr_h, trowptr_h + n);
    for (int32_t i = 0; i < n; ++i) {
        for (int32_t e = rowptr_h[i]; e < rowptr_h[i + 1]; ++e) {
            const int32_t j = col_h[e];
            const int32_t pos = fill[j]++;
            tcol_h[pos] = i;
            tedge_h[pos] = e;
        }
    }
    return FGC_OK;
}

// Parent-compressed graph of a level whose conv reads a 4x-upsampled coarse tensor (custom_upsampling, model.py:817-825,
// feeding custom_conv2d at model.py:905,926).  Every neighbour j of a fine node contributes x_(j >> 2), and the soft
// assignment of edge (i, j) depends on (i >> 2, j >> 2) only, so the edges of the four siblings of block p = i >> 2 that
// point into the same coarse row P collapse into ONE pair (p, P) carrying the four multiplicities.  Pairs of a block are
// stored in ascending P.  Two calls: pcol_h == NULL counts.
extern "C" int fgc_pair_graph(const int32_t* rowptr_h, const int32_t* col_h, int32_t n, int32_t* prow_h, int32_t* pcol_h,
                              uint32_t* pmul_h, int64_t* n_pairs_out) {
    FGC_CHECK_ARG(rowptr_h && col_h && prow_h && n >= 0 && n % 4 == 0, "fgc_pair_graph: bad arguments (n=%d must be a multiple of 4)", n);
    FGC_CHECK_ARG((pcol_h == nullptr) == (pmul_h == nullptr), "fgc_pair_graph: pcol / pmul must come together");
    const int32_t nc = n / 4;
    int64_t np = 0;
    prow_h[0] = 0;
    std::vector<std::pair<int32_t, int32_t>> tmp;   // (P, child)
    for (int32_t b = 0; b < nc; ++b) {
        tmp.clear();
        for (int32_t ch = 0; ch < 4; ++ch) {
            const int32_t i = 4 * b + ch;
            for (int32_t e = rowptr_h[i]; e < rowptr_h[i + 1]; ++e) {
                const int32_t j = col_h[e];
                FGC_CHECK_ARG(j >= 0 && j < n, "fgc_pair_graph: col[%d]=%d outside [0,%d)", e, j, n);
                tmp.emplace_back(j >> 2, ch);
            }
        }
        std::sort(tmp.begin(), tmp.end());
        size_t t = 0;
        while (t < tmp.size()) {
            const int32_t P = tmp[t].first;
            uint32_t cnt[4] = {0, 0, 0, 0};
            for (; t < tmp.size() && tmp[t].first == P; ++t) cnt[tmp[t].second]++;
            FGC_CHECK_ARG(cnt[0] < 256 && cnt[1] < 256 && cnt[2] < 256 && cnt[3] < 256,
                          "fgc_pair_graph: more than 255 edges of one node into one coarse row (block %d)", b);
            if (pcol_h) {
                pcol_h[np] = P;
                pmul_h[np] = cnt[0] | (cnt[1] << 8) | (cnt[2] << 16) | (cnt[3] << 24);
            }
            ++np;
        }
        FGC_CHECK_ARG(np <= INT32_MAX, "fgc_pair_graph: more than 2^31 pairs");
        prow_h[b + 1] = (int32_t)np;
    }
    if (n_pairs_out) *n_pairs_out = np;
    return FGC_OK;
}

// ---- CRC-32C, slicing-by-8 --------------------------------------------------------------
namespace fgc {
struct Crc32cTables {
    uint32_t t[8][256];
    Crc32cTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? 0x82F63B78u : 0u);
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xffu];
    }
};
}  // namespace fgc

extern "C" uint32_t fgc_crc32c(uint32_t crc, const void* data, size_t n) {
    static const fgc::Crc32cTables T;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = ~crc;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7u)) {
        c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xffu];
        --n;
    }
    while (n >= 8) {
        uint64_t w;
        memcpy(&w, p, 8);
        w ^= c;
        c = T.t[7][w & 0xff] ^ T.t[6][(w >> 8) & 0xff] ^ T.t[5][(w >> 16) & 0xff] ^ T.t[4][(w >> 24) & 0xff] ^
            T.t[3][(w >> 32) & 0xff] ^ T.t[2][(w >> 40) & 0xff] ^ T.t[1][(w >> 48) & 0xff] ^ T.t[0][w >> 56];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ T.t[0][(c ^ *p++) & 0xffu];
    return ~c;
}
