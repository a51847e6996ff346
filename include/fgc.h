/*
 * fgc.h - C ABI of libfgc.so, the MI355X (gfx950) facet-graph-convolution kernels.
 *
 * The reference (Elensil/Facet_Graph_Convolution) has no FFI: its "operator API" is the
 * set of Python functions in Code/model.py that build TensorFlow graph nodes.  This
 * header is the boundary a maintainer would bind instead (ctypes stub: INTEGRATION.md);
 * each entry point names the reference symbol (file:line) it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _h (host);
 *   - the caller owns every buffer; the library allocates nothing and never synchronises;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*); 0 = default stream;
 *   - return 0 on success, a negative FGC_E* code otherwise; fgc_last_error() gives the
 *     text for the calling thread;
 *   - float tensors are fp32 row-major with channels innermost ([n, C]); index tensors int32;
 *   - M (number of soft-assignment weight matrices) is fixed to FGC_M = 9 (model.py:855).
 *
 * Adjacency: the reference K-list (int32 [n, K], one-indexed, 0 = empty slot, slot 0 =
 * self; utils.py:243-295, utils.py:1799-1827) is converted once to CSR that PRESERVES
 * slot order and duplicates: col[rowptr[i] .. rowptr[i+1]) = adj[i, k] - 1 over the
 * non-zero slots k in increasing k.  deg(i) = rowptr[i+1] - rowptr[i] = count_nonzero
 * (model.py:436).
 */
#ifndef FGC_H
#define FGC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FGC_M 9            /* soft-assignment matrices per conv (model.py:855,868,880) */
#define FGC_AG_LD 24       /* row stride of the assignment-logit table: a[0..8] pad, g[12..20] pad */
#define FGC_DL_LD 12       /* row stride of the per-edge dlogit buffer */

#define FGC_OK 0
#define FGC_EINVAL (-22)
#define FGC_ENOMEM (-12)
#define FGC_EHIP (-5)

/* ABI version of this header.  fgc_version() returns the value the LIBRARY was built with: a binding compares the two
 * (and the struct sizes below) before its first call.  102: fgc_set_option / fgc_get_option; fgc_conv_pack(extra),
 * flags of fgc_mlp_fwd / fgc_mlp_bwd and their _bf16 forms, larger fgc_conv_desc / fgc_conv_bwd_io (all since 101).
 * 103: fgc_conv_pairs_allowed; options NO_BFM, K1_QS14; fgc_conv_desc.options / n_options (per-descriptor option overrides).
 * 104: fgc_conv_bwd_io.r_ld (the stride of r stated with the buffer); fgc_conv_desc.packed_layout + fgc_conv_layout_id,
 * fgc_mlp_layout_id / FGC_MLP_LAYOUT (packed operands carry the identity of their layout). */
#define FGC_ABI_VERSION 104

const char* fgc_last_error(void);
int fgc_version(void);
/* Process-level options: which kernel form a launch takes where the library has more than one (A/B switches and
 * developer knobs; the defaults are the measured best).  Names are those of csrc/fgc_common.h's FGC_OPTION_LIST, with or
 * without an "FGC_" prefix ("NO_PAIRS", "FGC_W8_DATA16_MIN_N", ...).  The table is filled ONCE per process, the first time
 * any option is read, from environment variables FGC_<NAME>; after that only fgc_set_option changes it and no launch path
 * reads the environment.  Values apply to the calls that follow; do not change an option between a call that packs or
 * plans (fgc_conv_pack, the *_workspace_bytes queries, fgc_conv_uses_pairs) and the calls that consume its result.
 * fgc_option_name(i), 0 <= i < fgc_option_count(), enumerates the names.  Unknown name: FGC_EINVAL. */
int fgc_set_option(const char* name, int64_t value);
int fgc_get_option(const char* name, int64_t* value);
int32_t fgc_option_count(void);
const char* fgc_option_name(int32_t index);
/* sizeof the descriptor structs as this library was compiled (which = 0: fgc_conv_desc, 1: fgc_conv_bwd_io,
 * 2: fgc_pack_extra; else 0):
 * lets a foreign-language binding check its mirror of the layouts before the first call */
size_t fgc_struct_size(int32_t which);

/* Optional per-kernel timing (hipEvents around every launch of the library; off by default, not
 * capturable into a hipGraph while on).  fgc_profile_collect synchronises and writes one line
 * "tag/kernel count total_ms" per distinct kernel into buf; returns the bytes written. */
int fgc_profile_enable(int on);
int fgc_profile_tag(const char* tag);
int fgc_profile_collect(char* buf, int32_t buf_bytes);

/* ------------------------------------------------------------------------------------
 * Host-side graph conversion (CPU; pointers are HOST pointers)
 * ---------------------------------------------------------------------------------- */

/* K-list -> CSR.  rowptr_h [n+1]; col_h may be NULL to query nnz only.
 * replaces: the K-padded adjacency fed to get_slices/get_patches (model.py:380-405). */
int fgc_csr_from_klist(const int32_t* adj_h, int32_t n, int32_t K, int32_t* rowptr_h, int32_t* col_h,
                       int64_t* nnz_out);
/* CSR -> K-list (zero padded).  Bit-identical round trip for rows whose zeros are trailing. */
int fgc_klist_from_csr(const int32_t* rowptr_h, const int32_t* col_h, int32_t n, int32_t K, int32_t* adj_h);
/* Transposed CSR for the backward pass: for every node j the list of (i, e) such that forward
 * edge e = (i -> j); in-edges ordered by e.  trowptr_h [n+1], tcol_h [nnz] (= i), tedge_h [nnz] (= e). */
int fgc_csr_transpose(const int32_t* rowptr_h, const int32_t* col_h, int32_t n, int32_t* trowptr_h,
                      int32_t* tcol_h, int32_t* tedge_h);
/* Parent-compressed ("pair") graph of a level whose convolution reads a 4x-upsampled coarse tensor
 * (custom_upsampling model.py:817-825 feeding custom_conv2d at model.py:905,926; n % 4 == 0).  Neighbour j of a fine
 * node contributes the coarse row j >> 2, and the soft assignment of edge (i, j) (model.py:74-95) depends on
 * (i >> 2, j >> 2) only: all edges of the four siblings of block p = i >> 2 into one coarse row P are ONE pair (p, P)
 * with four multiplicities.  prow_h [n/4 + 1]; pcol_h [n_pairs] = P, ascending inside a block; pmul_h [n_pairs] =
 * multiplicity of P in the list of child 0 | child 1 << 8 | child 2 << 16 | child 3 << 24 (each < 256).
 * pcol_h = pmul_h = NULL: count only.  The transposed pair graph is fgc_csr_transpose(prow, pcol, n/4). */
int fgc_pair_graph(const int32_t* rowptr_h, const int32_t* col_h, int32_t n, int32_t* prow_h, int32_t* pcol_h,
                   uint32_t* pmul_h, int64_t* n_pairs_out);

/* ------------------------------------------------------------------------------------
 * Host-side mesh preprocessing (CPU, native C++; HOST pointers).  These replace the Python loops that
 * build the tensors the network is fed (dataClasses.py:34-233).
 * ---------------------------------------------------------------------------------- */

/* unit face normals (computeFacesNormals utils.py:63-68 + normalize utils.py:26-35, fp32) and face
 * barycentres divided by the bounding-box diagonal (getTrianglesBarycenter utils.py:1264-1294, fp32
 * arithmetic stored as double like the reference's float64 array).  V [nv,3] f32, F [nf,3] u32. */
int fgc_face_features(const float* V_h, int32_t nv, const uint32_t* F_h, int32_t nf, float* normals_h,
                      double* centres_h);
/* vertex-sharing facet adjacency K-list (getFacesLargeAdj utils.py:243-295), bit-exact: row i =
 * [i+1, neighbours+1 in construction order (edge neighbours twice), 0...].  *unregistered counts the
 * connections dropped because a row was full. */
int fgc_faces_large_adj(const uint32_t* F_h, int32_t nf, int32_t nv, int32_t K, int32_t* adj_h,
                        int64_t* unregistered);
/* one greedy pairing pass (metis_one_level lib/coarsening.py:135-192), bit-exact given its arguments */
int fgc_metis_one_level(const int32_t* rr_h, const int32_t* cc_h, const float* vv_h, int64_t nnz,
                        const int64_t* rid_h, const float* weights_h, int32_t N, int32_t* cluster_id_h,
                        double* total_assoc);
/* Graph hierarchy = listToSparseWNormals (utils.py:1753-1796) + coarsen (lib/coarsening.py:5-31).
 * The handle owns host memory (the only allocation the library makes); free it with fgc_hierarchy_free.
 * parents_h (optional): `levels` recorded cluster-assignment arrays to replay (then `seed` is unused and the
 * result is bit-identical to the reference run that produced them); parents_len_h their lengths. */
typedef struct fgc_hierarchy fgc_hierarchy;
int fgc_hierarchy_build(const int32_t* adj_h, int32_t n, int32_t K, const double* pos_h, const float* normals_h,
                        int32_t levels, uint64_t seed, const int32_t* const* parents_h,
                        const int32_t* parents_len_h, fgc_hierarchy** out);
void fgc_hierarchy_free(fgc_hierarchy* h);
int32_t fgc_hierarchy_size(const fgc_hierarchy* h, int32_t level);       /* padded node count */
int32_t fgc_hierarchy_real_size(const fgc_hierarchy* h, int32_t level);  /* before fake nodes */
int fgc_hierarchy_new_to_old(const fgc_hierarchy* h, int32_t level, int32_t* out_h);
int fgc_hierarchy_parents(const fgc_hierarchy* h, int32_t level, int32_t* out_h);
/* K-list of graph `level` in binary-tree order (perm_adjacency coarsening.py:269-296 + sparseToList
 * utils.py:1799-1827); *saturated = 1 if a row had more than K-1 neighbours. */
int fgc_hierarchy_klist(const fgc_hierarchy* h, int32_t level, int32_t K, int32_t* adj_h, int32_t* saturated);

/* Breadth-first patch of a facet graph = getGraphPatch_wMask (utils.py:1508-1696), host: what the reference cuts
 * meshes above MAX_PATCH_SIZE into (dataClasses.py:76-171).  adj_h [n, K] one-indexed K-list, mask_h [n] (1 = node
 * already covered by an earlier patch: added as context, expanded only while the patch is below min_patch_size).
 * Outputs: patch_adj_h [nodes_num + K, K] one-indexed K-list of the patch in discovery order (rows of expanded nodes
 * keep their slots, rows of nodes still queued when growth stopped are compacted), old_index_h [nodes_num + K],
 * *patch_n the number of nodes written, *next_seed an uncovered node seen just outside the patch or -1.
 * Same queue discipline as the reference, so all three are bit-identical to its output. */
int fgc_graph_patch(const int32_t* adj_h, int32_t n, int32_t K, int32_t nodes_num, int32_t seed, const int8_t* mask_h,
                    int32_t min_patch_size, int32_t* patch_adj_h, int32_t* old_index_h, int32_t* patch_n,
                    int32_t* next_seed);

/* Breadth-first patch of a MESH = getMeshPatch (utils.py:1298-1410), host: what the multi-scale pipeline cuts meshes
 * above maxSize into (dataClasses.py:270-372).  V_h [nv,3], F_h [nf,3] int32, adj_h [nf,K] one-indexed K-list of the
 * faces.  Grows from face `seed` until face_num faces are in (plus the at most K-1 discovered by the last expanded
 * face).  Outputs, all in discovery order: v_out_h [v_cap,3] vertices of the patch (v_cap = int(0.6 * face_num) + K in
 * the reference; -EINVAL where the reference raises IndexError), f_out_h [face_num+K,3] faces in patch vertex ids,
 * adj_out_h [face_num+K,K] their one-indexed K-list (rows of faces still queued when growth stopped are compacted),
 * v_old_h / f_old_h the mesh index of every patch vertex / face, *n_v / *n_f the counts.  Same queue discipline as the
 * reference: bit-identical outputs. */
int fgc_mesh_patch(const float* V_h, int32_t nv, const int32_t* F_h, int32_t nf, const int32_t* adj_h, int32_t K,
                   int32_t face_num, int32_t seed, float* v_out_h, int32_t v_cap, int32_t* f_out_h, int32_t* adj_out_h,
                   int32_t* v_old_h, int32_t* f_old_h, int32_t* n_v, int32_t* n_f);

/* Faces incident to every vertex = getVerticesFaces (utils.py:370-395), host.  faces_h [nf,3] int32 with -1 rows for
 * fake faces (skipped); v_faces_h [nv, k_v] receives the face (row) indices in face order, -1 padded.  -EINVAL if a
 * vertex is in more than k_v faces (the reference raises IndexError). */
int fgc_vertices_faces(const int32_t* faces_h, int32_t nf, int32_t nv, int32_t k_v, int32_t* v_faces_h);

/* Edge map = getEdgeMap (utils.py:91-183), host.  e_map_h [3*nf, 4] receives [v1, v2, f1, f2] per edge (f2 = -1
 * on a boundary), *n_edges the number of edges written; v_e_map_h [nv, max_edges] the edge ids incident to every
 * vertex in creation order, -1 padded.  Same visiting order as the reference, so both tables are bit-identical to
 * its output.  -EINVAL if a vertex has more than max_edges edges (the reference raises IndexError there). */
int fgc_edge_map(const uint32_t* faces_h, int32_t nf, int32_t nv, int32_t max_edges, int32_t* e_map_h,
                 int32_t* n_edges, int32_t* v_e_map_h);

/* ------------------------------------------------------------------------------------
 * Graph convolution  (replaces custom_conv2d, model.py:427-504, invariance-off branch,
 * with get_weight_assigments model.py:74-95 and get_patches model.py:380-405 fused in)
 *
 *   a_i = u x_i + c,  g_j = v x_j,  q_ik = softmax_m(a_i + g_j(i,k))
 *   y_i = (1/deg_i) sum_k sum_m q_ikm W0[m] x_j(i,k) + b [deg_i > 0]
 *
 * The input of node j is the channel concatenation [x0 | x1] of row (j >> shift) of each
 * source (shift = 2 reads a 4x-upsampled coarse tensor, model.py:817-825, without
 * materialising it; x1 = NULL for a single source; concat replaces tf.concat model.py:909,929).
 * act: 0 = none, 1 = leaky ReLU with `alpha` (model.py:828-830) applied to y.
 * y_pool (optional, may be NULL): max over each 4 consecutive rows of y (model.py:779-788).
 * ---------------------------------------------------------------------------------- */
/* One per-call override of a process-level option (fgc_set_option): `index` is the option's position in fgc_option_name's
 * enumeration.  A descriptor that carries a list (fgc_conv_desc.options) runs EVERY entry point it is passed to - workspace
 * queries, fgc_conv_uses_pairs, fgc_conv_pack, forward, backward, fgc_conv_bwd_reduce - with these values instead of the
 * process-level ones, for that descriptor only and on the calling thread only: two networks in one process (or two layers of
 * one network) can take different kernel forms without touching global state.  The same caveat as for fgc_set_option holds per
 * descriptor: keep a descriptor's list the same from its plan / pack calls to the calls that consume their results. */
typedef struct fgc_option_override {
    int32_t index;
    int32_t reserved;       /* 0 */
    int64_t value;
} fgc_option_override;

typedef struct fgc_conv_desc {
    int32_t n;              /* nodes of this level */
    int32_t nnz;            /* edges */
    const int32_t* rowptr;  /* [n+1] */
    const int32_t* col;     /* [nnz]; at least one readable entry (a level of nothing but edgeless nodes reads col[0]) */
    const float* x0;        /* [(n >> shift), c0] */
    const float* x1;        /* [(n >> shift), c1] or NULL */
    int32_t c0, c1;         /* cin = c0 + c1 */
    int32_t shift;          /* 0 or 2 */
    int32_t cout;
    const float* W0;        /* [M, cout, cin]  (model.py:430) */
    const float* b;         /* [cout]          (model.py:431) */
    const float* u;         /* [M, cin]        (model.py:432) */
    const float* c;         /* [M]             (model.py:433) */
    const float* v;         /* [M, cin]        (model.py:447) */
    int32_t bias_mask;      /* model.py:496-500 */
    int32_t act;
    float alpha;
    int32_t src_rows;       /* rows of x0/x1 to compute assignment logits for; 0 = n >> shift.  Larger when the
                               source tensors carry halo rows behind the owned ones (facet sharding) */
    int32_t max_deg;        /* max_i deg(i) if the caller knows it, else 0.  <= 24 (every reference K-list) enables
                               the producer/consumer kernels; 0 or larger falls back to the edge-chunking path */
    /* Partial forward calls (all zero = the whole layer in one call).  A facet-sharded caller computes the tiles
     * that gather owned rows only while the halo rows are still travelling, then the rest:
     *   call 1: tile_list = interior tiles, proj_rows = owned source rows
     *   call 2: tile_list = boundary tiles, proj_row0 = owned source rows, proj_rows = halo rows, FGC_CONV_PACKED
     * fgc_conv_bwd ignores these four fields (its split is in fgc_conv_bwd_io). */
    const int32_t* tile_list; /* device, [n_tiles]: indices of the 32-row output tiles to compute; NULL = all */
    int32_t n_tiles;          /* may be 0 with a non-NULL list: no conv launch, logits only */
    int32_t proj_row0;        /* assignment logits are computed for source rows [proj_row0, proj_row0 + proj_rows) */
    int32_t proj_rows;        /* 0 = all src_rows (proj_row0 must be 0), < 0 = none */
    int32_t flags;            /* FGC_CONV_PACKED */
    /* Optional, shift == 2 only: the pair graph of this level (fgc_pair_graph).  With it (and hc) the layer runs on its
     * COARSE source rows: h_P = W0 x_P once per coarse row (one [n/4, cin] x [cin, M*cout] product instead of a tile
     * product per fine node), y_i = (1/deg_i) sum_P mult_iP sum_m q_pPm h_Pm - the same sums as the fine form in another
     * order; fgc_conv_uses_pairs tells whether a descriptor qualifies.  All NULL / 0: the fine form. */
    const int32_t* pair_rowptr;  /* [n/4 + 1] */
    const int32_t* pair_col;     /* [n_pairs]: source row of the parent (< src_rows); at least one readable entry */
    const uint32_t* pair_mul;    /* [n_pairs]: four 8-bit multiplicities (fgc_pair_graph); at least one readable entry */
    int32_t n_pairs;
    int32_t max_pair_deg;        /* largest number of pairs of a block */
    int32_t max_pair_in_deg;     /* largest number of in-pairs of a coarse row (transposed pair graph); 0 = not known:
                                    forward only.  fgc_conv_bwd needs 1 .. 24 */
    float* hc;                   /* [src_rows (or n/4), M*cout], 16-byte aligned: the transformed coarse rows, written by
                                    fgc_conv_fwd and read again by fgc_conv_bwd (bf16 with FGC_CONV_BF16).  A partial forward
                                    call transforms the source rows [proj_row0, proj_row0 + proj_rows) only; with a tile_list
                                    of no tiles it stops there, without a tile_list it then computes every block (a
                                    facet-sharded caller: owned rows while the halo parents travel, then the rest) */
    const fgc_option_override* options;  /* HOST pointer, [n_options]: per-descriptor option values (see above); NULL = none */
    int32_t n_options;
    int32_t reserved1;
    uint64_t packed_layout;      /* 0, or fgc_conv_layout_id(this descriptor) as it read when the operands in the workspaces
                                    were packed: a call told FGC_CONV_PACKED then returns FGC_EINVAL if the option values of
                                    the moment select another operand layout, instead of multiplying by the wrong one */
} fgc_conv_desc;

/* the workspace still holds the packed operands of the previous call with this descriptor: skip the packing */
#define FGC_CONV_PACKED 1
#define FGC_CONV_DEFER_REDUCE 2   /* fgc_conv_bwd_io.flags: stage 8 leaves its partial sums for fgc_conv_bwd_reduce */
#define FGC_CONV_DEFER_DW 16      /* fgc_conv_bwd_io.flags, with FGC_CONV_DEFER_REDUCE: stage 8 does not launch the layer's
                                   * weight-gradient GEMM either; fgc_conv_bwd_reduce runs the GEMMs of all such layers in one
                                   * launch per kernel form (same slabs, bit-identical gradients).  The layer's r (a first layer
                                   * over a narrow input: ds and its saved aggregates), its inputs x0 / x1 and its workspace must
                                   * then stay untouched until that call: a caller that shares r between layers cannot use it */
#define FGC_CONV_R_PAD 32         /* fgc_conv_bwd_io.flags: the rows of r are padded to whole 128-byte lines - row stride
                                     fgc_conv_r_ld(cout, 1, bf16) elements instead of M*cout + 24 (columns unchanged: M*cout
                                     aggregate columns, then da | dg; the pad columns are never read).  Read only when
                                     fgc_conv_bwd_io.r_ld == 0; a caller that sets r_ld states the stride once, with the
                                     buffer, and need not repeat the flag in every staged call */
#define FGC_CONV_SAVE_Z 4         /* fgc_conv_desc.flags, first layer over a narrow input (cin <= 8): the forward pass
                                  * leaves the aggregates z [n, roundup4(9*cin)] in its workspace (sized for it by
                                  * fgc_conv_workspace_bytes when the flag is set) so that the backward pass, given
                                  * fgc_conv_bwd_io.z_saved, does not recompute them for the weight gradient */

#define FGC_CONV_BF16 8           /* fgc_conv_desc.flags: bf16 STORAGE / fp32 accumulate (a build extension: the reference is
                                  * fp32 only, train.py:409-427).  x0, x1, y, y_pool and, in fgc_conv_bwd, dy, ds, r, dx0 and
                                  * dx1 point to bf16 (uint16_t) tensors of the same shapes; the gathers move half the
                                  * bytes and the dense contractions run on v_mfma_f32_16x16x32_bf16 with fp32
                                  * accumulation.  Logit tables (ag, dag), per-edge d-logits (dl), parameters and
                                  * parameter gradients stay fp32; the weight-gradient reduction over nodes multiplies the
                                  * bf16-stored operands on the fp32 MFMA (exact products, fp32 sums).  Supported for the
                                  * shapes of the network (widths that are multiples of 32, degrees <= 24); a narrow first
                                  * layer (cin <= 8) keeps its fp32 input x0 and ds, and stores y / y_pool as bf16.
                                  * Same flag in fgc_conv_bwd_io.flags is not needed: the descriptor's flag rules. */

/* bytes of scratch the conv entry points need for this descriptor (packed weights) */
size_t fgc_conv_workspace_bytes(const fgc_conv_desc* d);

/* forward.  ag [(n >> shift), FGC_AG_LD] receives the assignment logits (kept for backward).
 * y [n, cout]; y_pool [n/4, cout] or NULL. */
int fgc_conv_fwd(const fgc_conv_desc* d, float* ag, float* y, float* y_pool, void* workspace,
                 size_t workspace_bytes, void* stream);

/* backward.  Transposed CSR required.  dy is the gradient w.r.t. the POST-activation output y
 * (y itself is passed to recover the leaky-ReLU mask).  Scratch the caller provides:
 *   ds   [n, cout]           dy * lrelu'(y) / deg
 *   dl   [nnz, FGC_DL_LD]    per-edge d(logit)
 *   dag  [n, FGC_AG_LD]      d a (0..8) | d g (12..20) per node of this level
 *   r    [n, M*cout + 24]    backward-side aggregate, then da | dg of the node (one GEMM gives dW0, du and dv)
 * Outputs: dW0,db,du,dc,dv (overwritten); dx0/dx1 [(n >> shift), c0/c1] either overwritten
 * (accumulate = 0) or added to (accumulate = 1); dx pointers may be NULL (conv1: no input grad). */
typedef struct fgc_conv_bwd_io {
    const int32_t* trowptr; /* [n+1] */
    const int32_t* tcol;    /* [nnz] */
    const int32_t* tedge;   /* [nnz] */
    int32_t max_in_deg;     /* max in-degree of the transposed graph if known, else 0 (see max_deg) */
    int32_t stages;         /* 0 = whole backward.  Otherwise a bit mask, so that a facet-sharded caller can exchange
                               halo rows between the pieces: 1 = ds/db, 2 = logits (dl, da, dc), 4 = data kernel
                               (r, dg, dx) over data_tile_list, 8 = weight gradients (dW0, du, dv; needs every r and
                               dg row).  ds then has rows for halo sources behind the n owned ones, and dl rows for
                               incoming cross-shard edges behind the nnz owned ones. */
    const float* ag;        /* saved by forward */
    const float* y;         /* forward output (post activation) */
    const float* dy;        /* [n, cout] */
    float* ds;
    float* dl;
    float* dag;
    float* r;
    float* dx0;
    float* dx1;
    int32_t accumulate0, accumulate1;
    float* dW0;
    float* db;
    float* du;
    float* dc;
    float* dv;
    const int32_t* data_tile_list; /* device, [n_data_tiles]: 32-row tiles stage 4 computes; NULL = all.  Tiles whose
                                      in-edges all come from owned rows need neither halo rows of ds nor remote dl.  Pair
                                      form: tiles of the n / 4 COARSE rows the data kernel runs over (those whose in-pairs
                                      all have owned parents need no incoming dt / dl row) */
    int32_t n_data_tiles;
    int32_t flags;                 /* FGC_CONV_PACKED: the operands are already packed (an earlier stage call, or
                                    * fgc_conv_pack); FGC_CONV_DEFER_REDUCE: see fgc_conv_bwd_reduce */
    const float* z_saved;          /* optional: the workspace of the forward call made with FGC_CONV_SAVE_Z */
    /* optional: the layer's output also went through the 4:1 max pooling (custom_binary_tree_pooling, model.py:779-788)
     * and the pooled tensor's gradient is still to be folded in: stage 1 then uses
     *   dy_i + [y_i == pool_y_(i/4)] * pool_dy_(i/4) / #{rows of the group equal to the maximum}
     * instead of dy_i (tf.reduce_max's gradient, as fgc_pool4_bwd computes it) - no separate pass over dy.
     * pool_y, pool_dy: [n / 4, cout], the pooled output and its gradient (bf16 with FGC_CONV_BF16). */
    const float* pool_y;
    const float* pool_dy;
    /* pair form (fgc_conv_desc.pair_rowptr): the transposed pair graph and scratch for the per-pair output gradients
     * dt_pP = sum_children mult_iP dy_i / deg_i.  dl then holds one row per PAIR, r one row per COARSE row
     * ([n/4, M*cout + 24]); ds is not used. */
    const int32_t* tpair_rowptr; /* [n/4 + 1] */
    const int32_t* tpair_col;    /* [n_pairs]: block p of the in-pair */
    const int32_t* tpair_edge;   /* [n_pairs]: its pair id */
    float* dt;                   /* [n_pairs (+ incoming cross-shard pairs), cout], 16-byte aligned (bf16 with
                                    FGC_CONV_BF16): stage 1|2 writes the rows of the owned pairs, stage 4 gathers the rows the
                                    transposed pair graph names (tpair_edge), like dl */
    int32_t r_ld;                /* row stride of r in elements (fp32 floats / bf16 halves), a property of the BUFFER: stage 4
                                    writes with it, stage 8 and fgc_conv_bwd_reduce read with it, whatever `flags` holds in
                                    each of those calls.  0 = derive it from flags (FGC_CONV_R_PAD set or not) in every call,
                                    as ABI 103 did.  Otherwise >= M*cout + 24 and congruent to it modulo 4 (bf16: modulo 8) -
                                    fgc_conv_r_ld() returns the two strides the kernels are tuned for. */
    int32_t reserved0;
} fgc_conv_bwd_io;

size_t fgc_conv_bwd_workspace_bytes(const fgc_conv_desc* d);
/* row stride of fgc_conv_bwd_io.r in elements (fp32: floats, FGC_CONV_BF16: 2-byte halves): M*cout + 24, or with
 * padded != 0 (FGC_CONV_R_PAD) that rounded up to a multiple of 128 bytes */
int32_t fgc_conv_r_ld(int32_t cout, int32_t padded, int32_t bf16);
/* 1 if a staged (facet-sharded) backward of this layer needs the halo rows of ds and the d-logits of incoming
 * cross-shard edges between its stages; 0 for a first layer over a narrow input (dx0 == NULL, cin <= 8), whose
 * parameter gradients are sums over the owned nodes only (stages 1, 2, 8; stage 4 is empty). */
int fgc_conv_bwd_needs_exchange(const fgc_conv_desc* d, const fgc_conv_bwd_io* io);
/* 1 if fgc_conv_fwd / fgc_conv_bwd run this descriptor in the pair form.  All of:
 *   - pair_rowptr, pair_col, pair_mul and hc given; shift == 2; ONE source (x1 == NULL, c1 == 0); n % 4 == 0;
 *   - cin (= c0) 32, 64 or 128, cout 32 or 64; fp32 or bf16 storage (FGC_CONV_BF16: hc and dt are then bf16);
 *   - n_pairs > 0, max_pair_deg > 0, 0 <= max_pair_in_deg <= 24 (the data-gradient kernel's edge-slot limit);
 *   - rows of hc (src_rows, or n / 4) < 2^24 and rows * 9 * cout * 4 < 2^32; n_pairs < 2^24 - 2^20 and
 *     n_pairs * cout * 4 < 2^32 (row ids go through 24-bit multiplies and 32-bit buffer offsets; the margin leaves room for
 *     a facet-sharded rank's incoming cross-shard pairs behind its own in dt);
 *   - x0 and hc 16-byte aligned;
 *   - the whole layer in one call, OR one of the two partial forward calls a facet-sharded caller makes: "transform +
 *     logits of the source rows [proj_row0, proj_row0 + proj_rows) only" (tile_list != NULL with n_tiles == 0) and "the rest
 *     of the source rows, then every block" (tile_list == NULL, proj_row0 / proj_rows naming the rest).  A tile list WITH
 *     tiles (an interior / boundary split of the blocks) selects the fine form;
 *   - options NO_PAIRS, NO_W8 and NO_W8FAST all 0 (fgc_set_option; the last two take the pair form's data-gradient kernel
 *     away).
 * Anything else: 0, and the layer runs in the fine form without an error.  fgc_conv_bwd of a pair-form layer REQUIRES the
 * transposed pair graph and dt in its io.  Ranks of a facet-sharded job must agree on the answer per layer (they exchange
 * different tensors in the two forms): decide it on the GLOBAL pair graph (largest in-degree, pair counts), not per rank. */
int fgc_conv_uses_pairs(const fgc_conv_desc* d);
/* The GRAPH-dependent limits of the list above alone (in-pairs per coarse row, row and pair counts against the 24-bit row ids
 * and 32-bit buffer offsets of the pair kernels), for counts that need not be a descriptor's: a facet-sharded job evaluates
 * them on the GLOBAL pair graph so that every rank decides alike (fgc_conv_uses_pairs applies the same function to the
 * descriptor's own counts).  rows: coarse source rows; 1 = allowed. */
int fgc_conv_pairs_allowed(int64_t rows, int64_t n_pairs, int32_t max_pair_in_deg, int32_t cout);
/* Which packed-operand layouts the option values of THIS moment (process options, then the descriptor's own overrides)
 * select for the descriptor - first-layer path or not, pair form or not, d-logits operand as fp32 or as bf16 split planes,
 * storage type.  Never 0.  Operands packed ahead of their use (fgc_conv_pack, or an earlier call) are only valid while this
 * value stays what it was: store it in fgc_conv_desc.packed_layout and the FGC_CONV_PACKED calls check it (packed_layout is
 * not part of the value). */
uint64_t fgc_conv_layout_id(const fgc_conv_desc* d);
int fgc_conv_bwd(const fgc_conv_desc* d, const fgc_conv_bwd_io* io, void* workspace, size_t workspace_bytes,
                 void* stream);

/* Extra jobs of fgc_conv_pack's launch (each part optional: rot_x == NULL / mlp_W1 == NULL skips it). */
typedef struct fgc_pack_extra {
    /* rot_y[r, 3v:3v+3] = R rot_x[r, 3v:3v+3] for rot_rows rows of rot_vecs 3-vectors (fgc_rotate_rows) */
    const float* rot_x;
    float* rot_y;
    const float* rot_R;         /* [9] on the device */
    int64_t rot_rows;
    int32_t rot_vecs;
    /* optional (rot_vecs <= 2): the assignment-logit table of the layer that reads rot_y (the network's first layer:
     * rot_ag [rot_rows, 24] = what its fgc_conv_fwd would compute from u [9, 3 rot_vecs], c [9], v [9, 3 rot_vecs]) in the
     * same pass; that layer's forward call then sets proj_rows = -1.  Same arithmetic as the layer's own table launch. */
    float* rot_ag;
    const float* rot_u;
    const float* rot_c;
    const float* rot_v;
    /* the MLP whose operands are packed: W1 [cin, hidden], W2 [hidden, cout] (bf16 backward only), n rows.  mlp_fwd_ws /
     * mlp_bwd_ws are the workspaces later handed to fgc_mlp_fwd / fgc_mlp_bwd (mlp_bf16 != 0: to the _bf16 forms) with
     * FGC_MLP_PACKED; they must stay untouched in between, so the two calls need workspaces of their own.  Either may
     * be NULL. */
    int32_t mlp_bf16;
    const float* mlp_W1;
    const float* mlp_W2;
    int32_t mlp_n, mlp_cin, mlp_hidden, mlp_cout;
    void* mlp_fwd_ws;
    void* mlp_bwd_ws;
} fgc_pack_extra;

/* Whole-network helpers for a caller that runs the same `count` layers every step (train.py:558-575 runs the graph of
 * model.py:853-941 once per iteration) and gives every layer a workspace of its own that stays untouched from the
 * first call of a step to the last: the per-layer housekeeping launches (each costs about 5 us on an idle MI355X
 * whatever its size) collapse into one.
 *
 * fgc_conv_pack: the packed weight operands of all layers in ONE launch - forward operands into fwd_ws[i], the two
 * backward operands into bwd_ws[i] (either array, or single entries, may be NULL to skip; ios may be NULL, it only
 * tells which layer takes the narrow first-layer path and has nothing to pack).  Call it after the weights change;
 * then pass FGC_CONV_PACKED in fgc_conv_desc.flags / fgc_conv_bwd_io.flags.
 * `extra` (may be NULL): the step's other housekeeping in the same launch - the rotation of the input rows
 * (train.py:563-565, what fgc_rotate_rows does) and the operands of the network's per-facet MLP, which the fgc_mlp_*
 * entry points then take with FGC_MLP_PACKED.  count may be 0 with descs NULL.
 *
 * fgc_conv_bwd_reduce: (layers with FGC_CONV_DEFER_DW: first their weight-gradient GEMMs, grouped.)  With
 * FGC_CONV_DEFER_REDUCE in fgc_conv_bwd_io.flags stage 8 leaves the partial sums of the
 * parameter gradients in the layer's workspace; this call sums them for all layers in two launches, in the same
 * fixed order as the per-layer path (bit-identical gradients). */
int fgc_conv_pack(const fgc_conv_desc* const* descs, const fgc_conv_bwd_io* const* ios, void* const* fwd_ws,
                  void* const* bwd_ws, int32_t count, const fgc_pack_extra* extra, void* stream);
int fgc_conv_bwd_reduce(const fgc_conv_desc* const* descs, const fgc_conv_bwd_io* const* ios, void* const* bwd_ws,
                        int32_t count, void* stream);

/* ------------------------------------------------------------------------------------
 * Per-facet MLP  cin -> hidden -> cout with leaky ReLU in between
 * (replaces lrelu(custom_lin(x,1024)) -> custom_lin(.,3), model.py:763-769,937-941; the
 * [n, hidden] tensor never leaves the CU).  W1 [cin, hidden], b1 [hidden], W2 [hidden, cout], b2 [cout].
 * abs_partial (optional): per-workgroup partial sums of |y| for normalizeTensor's global mean;
 * needs fgc_mlp_num_partials(n) floats.
 * Everything is fp32 in and out.  For 32- and 64-wide inputs fgc_mlp_fwd multiplies x W1 on the bf16 matrix pipe with
 * three-term operand splits (v = bf16(v) + bf16(v - v0) + bf16(v - v0 - v1); six exact partial products per product,
 * fp32 accumulation): as close to a float64 reference as the fp32 MFMA kernel it replaces (1.7e-7 vs 2.0e-7 on outputs
 * of size 1).  FGC_NO_MLP_SPLIT=1 in the environment keeps the fp32 MFMA.  alpha must be in [0, 1].
 * flags: FGC_MLP_PACKED = the workspace already holds this call's weight operands (fgc_conv_pack with an
 * fgc_pack_extra naming it, same W1 / W2 / shape, nothing written to it since): the call skips its own pack launch.
 * ---------------------------------------------------------------------------------- */
#define FGC_MLP_PACKED 1
/* The MLP's counterpart of fgc_conv_layout_id: which operand layouts the option values of this moment select for the shape
 * (1 ... 255; bf16 != 0: the _bf16 entry points).  OR FGC_MLP_LAYOUT(id as it read when the workspace was packed) into the
 * flags of an FGC_MLP_PACKED call and the call returns FGC_EINVAL if the options moved in between; without it nothing is
 * checked. */
#define FGC_MLP_LAYOUT(id) ((int32_t)(id) << 8)
int32_t fgc_mlp_layout_id(int32_t cin, int32_t hidden, int32_t cout, int32_t bf16);
int32_t fgc_mlp_num_partials(int32_t n);
size_t fgc_mlp_workspace_bytes(int32_t cin, int32_t hidden, int32_t cout);
size_t fgc_mlp_bwd_workspace_bytes(int32_t n, int32_t cin, int32_t hidden, int32_t cout);
int fgc_mlp_fwd(const float* x, int32_t n, int32_t cin, int32_t hidden, int32_t cout, const float* W1,
                const float* b1, const float* W2, const float* b2, float alpha, float* y, float* abs_partial,
                int32_t flags, void* workspace, size_t workspace_bytes, void* stream);
/* dW1,db1,dW2,db2 overwritten; dx [n, cin] overwritten. */
int fgc_mlp_bwd(const float* x, const float* dy, int32_t n, int32_t cin, int32_t hidden, int32_t cout,
                const float* W1, const float* b1, const float* W2, float alpha, float* dx, float* dW1,
                float* db1, float* dW2, float* db2, int32_t flags, void* workspace, size_t workspace_bytes, void* stream);

/* The same MLP with bf16-STORED activations (companion of FGC_CONV_BF16; a build extension, the reference is fp32 only:
 * train.py:409-427).  x [n, cin] and dx [n, cin] are bf16 (uint16_t) tensors, cin in {32, 64, 128} (backward: 32, 64),
 * hidden a multiple of 256; parameters, their gradients, y and dy stay fp32.  The products with the hidden layer run on
 * v_mfma_f32_16x16x32_bf16 with fp32 accumulation, everything else in fp32. */
size_t fgc_mlp_bf16_workspace_bytes(int32_t cin, int32_t hidden, int32_t cout);
size_t fgc_mlp_bwd_bf16_workspace_bytes(int32_t n, int32_t cin, int32_t hidden, int32_t cout);
int fgc_mlp_fwd_bf16(const void* x, int32_t n, int32_t cin, int32_t hidden, int32_t cout, const float* W1,
                     const float* b1, const float* W2, const float* b2, float alpha, float* y, float* abs_partial,
                     int32_t flags, void* workspace, size_t workspace_bytes, void* stream);
int fgc_mlp_bwd_bf16(const void* x, const float* dy, int32_t n, int32_t cin, int32_t hidden, int32_t cout,
                     const float* W1, const float* b1, const float* W2, float alpha, void* dx, float* dW1, float* db1,
                     float* dW2, float* db2, int32_t flags, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Element-wise / reduction ops
 * ---------------------------------------------------------------------------------- */
/* leaky ReLU (model.py:828-830), forward and backward from the OUTPUT y */
int fgc_lrelu_fwd(const float* x, float* y, int64_t count, float alpha, void* stream);
int fgc_lrelu_bwd(const float* y, const float* dy, float* dx, int64_t count, float alpha, void* stream);
/* 4:1 max pooling over consecutive rows (model.py:779-788, steps = 2) */
int fgc_pool4_fwd(const float* x, float* y, int32_t n_out, int32_t c, void* stream);
/* gradient of max pooling, split evenly over ties (tf.reduce_max semantics).  accumulate: dx += */
int fgc_pool4_bwd(const float* x, const float* y, const float* dy, float* dx, int32_t n_out, int32_t c,
                  int32_t accumulate, void* stream);
/* the same on bf16 tensors (all four; FGC_CONV_BF16 storage) */
int fgc_pool4_bwd_bf16(const void* x, const void* y, const void* dy, void* dx, int32_t n_out, int32_t c,
                       int32_t accumulate, void* stream);
/* 1:4 upsampling by repetition (model.py:817-825) and its gradient (sum of 4 rows) */
int fgc_upsample4_fwd(const float* x, float* y, int32_t n_in, int32_t c, void* stream);
int fgc_upsample4_bwd(const float* dy, float* dx, int32_t n_in, int32_t c, int32_t accumulate, void* stream);

/* custom_binary_tree_pooling 'max' (model.py:779-788) and custom_upsampling (model.py:817-825) for any steps:
 * group = 2^steps consecutive rows per pooled row (the 4:1 forms above are group = 4).  Pooling gradient = tf.reduce_max's:
 * split evenly over the entries equal to the maximum. */
int fgc_pool_fwd(const float* x, float* y, int32_t n_out, int32_t c, int32_t group, void* stream);
int fgc_pool_bwd(const float* x, const float* y, const float* dy, float* dx, int32_t n_out, int32_t c, int32_t group,
                 int32_t accumulate, void* stream);
int fgc_upsample_fwd(const float* x, float* y, int32_t n_in, int32_t c, int32_t group, void* stream);
int fgc_upsample_bwd(const float* dy, float* dx, int32_t n_in, int32_t c, int32_t group, int32_t accumulate, void* stream);

/* custom_lin (model.py:763-769) on its own: y [n, cout] = x [n, cin] W [cin, cout] + b, and its gradients
 * (dx may be NULL; dW [cin, cout], db [cout]; the sum over the n rows is split over workgroups and added in a fixed order).
 * Exact fp32 products on v_mfma_f32_16x16x4_f32.  The network's own two linear layers are fgc_mlp_fwd / fgc_mlp_bwd (hidden
 * layer kept on chip); this is for callers that compose custom_lin -> lrelu -> custom_lin themselves. */
size_t fgc_lin_bwd_workspace_bytes(int32_t n, int32_t cin, int32_t cout);
int fgc_lin_fwd(const float* x, int32_t n, int32_t cin, int32_t cout, const float* W, const float* b, float* y, void* stream);
int fgc_lin_bwd(const float* x, const float* dy, int32_t n, int32_t cin, int32_t cout, const float* W, float* dx, float* dW,
                float* db, void* workspace, size_t workspace_bytes, void* stream);

/* normalizeTensor (utils.py:1700-1715).  scratch: 2 + fgc_norm_num_partials(n) floats.
 * If abs_partial/num_partials come from fgc_mlp_fwd they are used for the global mean,
 * otherwise pass NULL/0 and the op reduces |x| itself. */
int32_t fgc_norm_num_partials(int32_t n);
int fgc_normalize_fwd(const float* x, int32_t n, const float* abs_partial, int32_t num_partials, float* y,
                      float* scratch, void* stream);
int fgc_normalize_bwd(const float* x, const float* dy, int32_t n, float* dx, float* scratch, void* stream);
/* The same op in pieces, for a tensor whose rows are sharded over ranks (the global mean and its gradient need one
 * scalar all-reduce each, done by the caller between the pieces):
 *   apply:        y = normalise rows of x with scratch[0] = mean|x| + 1e-5 supplied by the caller
 *   bwd_partial:  dx <- d(x/s) per row, partial[0..fgc_norm_num_partials(n)) <- block sums of <d(x/s), x>
 *   bwd_apply:    dx <- dx / s + sign(x) * scratch[1] / total_count, scratch[1] = -(global sum) / s^2 from the caller */
int fgc_normalize_apply(const float* x, int32_t n, const float* scratch, float* y, void* stream);
int fgc_normalize_bwd_partial(const float* x, const float* dy, int32_t n, const float* scratch, float* dx,
                              float* partial, void* stream);
int fgc_normalize_bwd_apply(const float* x, int32_t n, float total_count, const float* scratch, float* dx,
                            void* stream);

/* angular loss on sampled rows (train.py:509-517 gather + faceNormalsLoss train.py:1272-1294).
 * fn, gt [n,3]; sample_ind int32 [ns]; loss_out [2] = {loss in degrees, number of real rows}.
 * bwd: dfn [n,3] is zero-filled then receives d loss / d fn (scaled by dloss). */
int fgc_angular_loss_fwd(const float* fn, const float* gt, const int32_t* sample_ind, int32_t ns, float* loss_out,
                         void* stream);
int fgc_angular_loss_bwd(const float* fn, const float* gt, const int32_t* sample_ind, int32_t ns, int32_t n,
                         const float* loss_out, float dloss, float* dfn, void* stream);

/* The loss end of one training step in TWO launches: normalizeTensor on the network output (utils.py:1700-1715), the
 * rotation of the ground truth (train.py:439-451, applied to the sampled rows only), the sampled angular loss
 * (train.py:509-517, faceNormalsLoss train.py:1272-1294), its gradient, and normalizeTensor's gradient - what
 * fgc_normalize_fwd + fgc_rotate_rows + fgc_angular_loss_fwd / _bwd + fgc_normalize_bwd do in seven.
 *   y [n,3]: the network output; abs_partial / num_partials: the partial sums of |y| that fgc_mlp_fwd leaves;
 *   gt [n,3]: UNROTATED ground truth; R: device pointer to 9 floats or NULL (identity); sample_ind int32 [ns];
 *   gacc [n,3]: scratch that must be ZERO on entry and is zero again when the call has run (only sampled rows are ever
 *   touched: the caller zero-fills it once, when it allocates it);
 *   n_conv [n,3] (may be NULL): the normalised rows; dy [n,3]: d loss / d y; loss_out [2] = {loss in degrees, real rows};
 *   scratch: fgc_loss_step_scratch_floats(ns) floats ([0] = mean|y| + eps, [1] = d loss / d of it, then one partial
 *   triple per 256 samples).
 * Same arithmetic per row as the separate entry points; sums over the samples instead of over all rows, and the division
 * by the number of real samples is applied once, at the end. */
int32_t fgc_loss_step_scratch_floats(int32_t ns);
int fgc_loss_step(const float* y, int32_t n, const float* abs_partial, int32_t num_partials, const float* gt,
                  const float* R, const int32_t* sample_ind, int32_t ns, float* gacc, float* n_conv, float* dy,
                  float* loss_out, float* scratch, void* stream);

/* The same for a FACET-SHARDED step (the rows of y are split over ranks): three calls with the step's
 * two scalar all-reduces between them.  sums: fgc_loss_shard_floats(ns_total) floats, ns_total = the number of samples of
 * the WHOLE step (it sizes the partial table identically on every rank); total_count = 3 x the rows of the whole tensor.
 *   1. fgc_loss_shard_abs_sum: sums[0] = this rank's sum of |y| (from fgc_mlp_fwd's partials)   -> ALL-REDUCE sums[0:1]
 *   2. fgc_loss_shard_samples: this rank's samples (LOCAL row ids, ns_local may be 0) -> gacc rows and the partial table
 *      sums[4 + 3 b] = {sum of angles, real samples, sum(d xs . y)} per 256 samples             -> ALL-REDUCE sums[4:]
 *   3. fgc_loss_shard_rows: dy, n_conv for the local rows; loss_out = the loss of the whole step; gacc zero again */
int32_t fgc_loss_shard_floats(int32_t ns_total);
int fgc_loss_shard_abs_sum(const float* abs_partial, int32_t num_partials, float* sums, void* stream);
int fgc_loss_shard_samples(const float* y, float total_count, const float* gt, const float* R, const int32_t* sample_local,
                           int32_t ns_local, int32_t ns_total, float* gacc, float* sums, void* stream);
int fgc_loss_shard_rows(const float* y, int32_t n_local, float total_count, int32_t ns_total, float* sums, float* gacc,
                        float* n_conv, float* dy, float* loss_out, void* stream);

/* random-rotation augmentation (train.py:439-451): every 3-vector v of every row becomes R v.
 * R: DEVICE pointer to 9 floats (row major; kept on the device so that a captured hipGraph can be
 * replayed with a new rotation).  vecs = channels / 3. */
int fgc_rotate_rows(const float* x, float* y, int32_t n, int32_t vecs, const float* R, void* stream);

/* TensorFlow-1 Adam over one flat parameter buffer (train.py:520, tf.train.AdamOptimizer defaults:
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)).  t is 1-based. */
int fgc_adam_step(float* p, const float* g, float* m, float* v, int64_t count, int32_t t, float lr, float b1,
                  float b2, float eps, void* stream);

/* inference epilogue (train.py:115-121,136; utils.py:26-35): out[f] = normalize^2(n_conv[perm[f]]), f < num_faces */
int fgc_infer_epilogue(const float* n_conv, const int32_t* perm, int32_t num_faces, float* out, void* stream);

/* halo pack / unpack for facet sharding: dst[i] = src[idx[i]] rows of width c */
int fgc_gather_rows(const float* src, const int32_t* idx, int32_t count, int32_t c, float* dst, void* stream);
int fgc_scatter_add_rows(const float* src, const int32_t* idx, int32_t count, int32_t c, float* dst, void* stream);

/* Several row copies in ONE launch: the pack and the unpack side of a grouped halo exchange.  A sharded
 * step sends the halo rows of up to three tensors to every peer at once; packing them per (tensor, peer) into the send
 * buffer of one all-to-all, and copying what arrives into the halo tails of the tensors, is one call each instead of a
 * launch per piece.  Job j:  dst_j[i, 0..width) = src_j[(idx_j ? idx_j[i] : i), 0..width)  for i < rows.
 * Rows are `width` dwords (a bf16 row of C channels is C / 2 dwords).  At most FGC_ROW_JOBS_MAX jobs; `jobs` is a host
 * array read before the call returns.  Jobs must not overlap each other's destinations. */
#define FGC_ROW_JOBS_MAX 32
typedef struct fgc_row_job {
    const float* src;
    const int32_t* idx;       /* device, or NULL: rows are consecutive in src */
    float* dst;
    int32_t rows, width;
} fgc_row_job;
int fgc_copy_rows_jobs(const fgc_row_job* jobs, int32_t njobs, void* stream);

/* Vertex update from denoised normals = update_position2 (train.py:1467-1557; called with 60 iterations and
 * lambda = 1/18 by inferNetOld, train.py:129-139).  Jacobi iterations
 *   x_i <- x_i + lambda * sum_{edges e = (i,j,f1,f2) of i} sum_{f in {f1,f2}} n_f (n_f . (x_j - x_i))
 * x [nv,3] input positions, normals [nf,3], e_map [ne,4], v_e_map [nv,max_edges] (device copies of fgc_edge_map's
 * output).  x_out [nv,3] receives the result, tmp [nv,3] is scratch; x may alias neither.  iters >= 0. */
int fgc_vertex_update(const float* x, float* x_out, float* tmp, int32_t nv, const float* normals, int32_t nf,
                      const int32_t* e_map, int32_t ne, const int32_t* v_e_map, int32_t max_edges, int32_t iters,
                      float lambda, void* stream);

/* 4:1 "average ignoring zero rows" pooling = custom_binary_tree_pooling(x, steps=2, 'avg_ignore_zeros')
 * (model.py:792-814): two rounds of pairwise means in which a row that is zero in every channel is replaced by its
 * partner first.  x [n, c] with n % 4 == 0 -> y [n/4, c]. */
int fgc_pool4_avg_iz(const float* x, int32_t n, int32_t c, float* y, void* stream);

/* Node centres of the finest level = first step of updateFacesCenter (train.py:1779-1787): barycentre of every face
 * of faces [n0,3] (int32, -1 corners read a zero vertex) -> fpos [n0,3]. */
int fgc_face_centers(const float* x, int32_t nv, const int32_t* faces, int32_t n0, float* fpos, void* stream);

/* Multi-scale vertex update = update_position_MS with updateFacesCenter (train.py:1668-1798), coarsening_steps = 2.
 * x [nv,3] positions; faces [n0,3] int32 in node order, -1 rows = fake nodes; v_faces [nv,k_v] finest-level node ids
 * of every vertex, -1 padded; normals[s] [n0 / 4^s, 3] for s = 0,1,2.  Coarse to fine, iters[0] iterations with the
 * level-2 nodes, iters[1] with level 1, iters[2] with level 0; in every iteration the node centres are recomputed
 * from the current positions (face barycentres, then two avg_ignore_zeros poolings) and
 *   x_v += (1 / #faces(v)) * sum_k n (n . (c - x_v)),  n, c of node floor(v_faces[v,k] / 4^s).
 * x_out [nv,3] receives the result; dx_out (may be NULL) [3][nv,3] the displacement of each stage in execution
 * order; scratch: at least 3*(nv*2 + n0 + n0/4 + n0/16) floats.  x may alias neither output. */
int fgc_vertex_update_ms(const float* x, float* x_out, int32_t nv, const int32_t* faces, int32_t n0,
                         const int32_t* v_faces, int32_t k_v, const float* normals0, const float* normals1,
                         const float* normals2, const int32_t* iters, float* dx_out, float* scratch,
                         size_t scratch_floats, void* stream);

/* ------------------------------------------------------------------------------------
 * Checkpoint files (CPU; HOST pointers)
 * ---------------------------------------------------------------------------------- */

/* CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) of data[0..n) continued from crc (0 to start): the
 * checksum of the TensorFlow tensor-bundle files tf.train.Saver writes and restores (train.py:79-87,522-534,
 * 551-552): every block of `<prefix>.index` and every tensor of `<prefix>.data-*` carries it.  Unmasked value. */
uint32_t fgc_crc32c(uint32_t crc, const void* data, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* FGC_H */
