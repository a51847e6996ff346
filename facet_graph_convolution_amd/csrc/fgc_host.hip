// Host-side pieces of libfgc: error text, K-list <-> CSR, transposed CSR.
#include <string.h>

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

extern "C" const char* fgc_last_error(void) { return fgc::g_err; }
extern "C" int fgc_version(void) { return 100; }

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
