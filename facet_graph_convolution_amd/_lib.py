"""ctypes binding of libfgc.so (the C ABI declared in include/fgc.h).

There is NO fallback: if the shared library is missing or an entry point fails, a
RuntimeError is raised.  torch is imported first so that libfgc resolves the HIP runtime
torch already loaded (same SONAME libamdhip64.so.7), i.e. both share streams and pointers.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the dlopen below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FGC_LIB", os.path.join(_HERE, "csrc", "libfgc.so"))  # FGC_LIB: developer A/B builds

ABI_VERSION = 104   # FGC_ABI_VERSION of the include/fgc.h this binding was written against
FGC_M = 9
AG_LD = 24
DL_LD = 12

c_f32p = C.c_void_p
c_i32p = C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("nnz", C.c_int32),
        ("rowptr", C.c_void_p), ("col", C.c_void_p),
        ("x0", C.c_void_p), ("x1", C.c_void_p),
        ("c0", C.c_int32), ("c1", C.c_int32), ("shift", C.c_int32), ("cout", C.c_int32),
        ("W0", C.c_void_p), ("b", C.c_void_p), ("u", C.c_void_p), ("c", C.c_void_p), ("v", C.c_void_p),
        ("bias_mask", C.c_int32), ("act", C.c_int32), ("alpha", C.c_float), ("src_rows", C.c_int32), ("max_deg", C.c_int32),
        ("tile_list", C.c_void_p), ("n_tiles", C.c_int32), ("proj_row0", C.c_int32), ("proj_rows", C.c_int32),
        ("flags", C.c_int32),
        # pair form of a convolution over a 4x-upsampled input (graph.FacetGraph.pairs, include/fgc.h)
        ("pair_rowptr", C.c_void_p), ("pair_col", C.c_void_p), ("pair_mul", C.c_void_p),
        ("n_pairs", C.c_int32), ("max_pair_deg", C.c_int32), ("max_pair_in_deg", C.c_int32),
        ("hc", C.c_void_p),
        # per-descriptor option overrides (include/fgc.h: fgc_option_override); HOST pointer
        ("options", C.c_void_p), ("n_options", C.c_int32), ("reserved1", C.c_int32),
        # fgc_conv_layout_id as it read when the workspaces were packed (0 = the FGC_CONV_PACKED calls do not check)
        ("packed_layout", C.c_uint64),
    ]


class OptionOverride(C.Structure):
    """struct fgc_option_override (include/fgc.h): one option value that holds for ONE descriptor."""
    _fields_ = [("index", C.c_int32), ("reserved", C.c_int32), ("value", C.c_int64)]


def option_overrides(**values):
    """A ctypes array of fgc_option_override for ConvDesc.options (keep the array alive as long as the descriptor):
    option_overrides(NO_PAIRS=1, W8_NT16=0).  Names as in fgc_option_name's enumeration."""
    names = option_names()
    arr = (OptionOverride * len(values))()
    for k, (name, value) in enumerate(values.items()):
        arr[k].index, arr[k].reserved, arr[k].value = names.index(name[4:] if name.startswith("FGC_") else name), 0, int(value)
    return arr


class RowJob(C.Structure):
    """struct fgc_row_job (include/fgc.h): one row copy of fgc_copy_rows_jobs."""
    _fields_ = [("src", C.c_void_p), ("idx", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int32), ("width", C.c_int32)]


class ConvBwdIO(C.Structure):
    _fields_ = [
        ("trowptr", C.c_void_p), ("tcol", C.c_void_p), ("tedge", C.c_void_p), ("max_in_deg", C.c_int32), ("stages", C.c_int32),
        ("ag", C.c_void_p), ("y", C.c_void_p), ("dy", C.c_void_p),
        ("ds", C.c_void_p), ("dl", C.c_void_p), ("dag", C.c_void_p), ("r", C.c_void_p),
        ("dx0", C.c_void_p), ("dx1", C.c_void_p),
        ("accumulate0", C.c_int32), ("accumulate1", C.c_int32),
        ("dW0", C.c_void_p), ("db", C.c_void_p), ("du", C.c_void_p), ("dc", C.c_void_p), ("dv", C.c_void_p),
        ("data_tile_list", C.c_void_p), ("n_data_tiles", C.c_int32), ("flags", C.c_int32),
        ("z_saved", C.c_void_p), ("pool_y", C.c_void_p), ("pool_dy", C.c_void_p),
        ("tpair_rowptr", C.c_void_p), ("tpair_col", C.c_void_p), ("tpair_edge", C.c_void_p), ("dt", C.c_void_p),
        # row stride of r, stated with the buffer (0: each call derives it from its flags / CONV_R_PAD)
        ("r_ld", C.c_int32), ("reserved0", C.c_int32),
    ]


class PackExtra(C.Structure):
    """struct fgc_pack_extra (include/fgc.h): the rotation and the MLP operands that ride on fgc_conv_pack's launch."""
    _fields_ = [
        ("rot_x", C.c_void_p), ("rot_y", C.c_void_p), ("rot_R", C.c_void_p), ("rot_rows", C.c_int64), ("rot_vecs", C.c_int32),
        ("rot_ag", C.c_void_p), ("rot_u", C.c_void_p), ("rot_c", C.c_void_p), ("rot_v", C.c_void_p),
        ("mlp_bf16", C.c_int32), ("mlp_W1", C.c_void_p), ("mlp_W2", C.c_void_p),
        ("mlp_n", C.c_int32), ("mlp_cin", C.c_int32), ("mlp_hidden", C.c_int32), ("mlp_cout", C.c_int32),
        ("mlp_fwd_ws", C.c_void_p), ("mlp_bwd_ws", C.c_void_p),
    ]


CONV_PACKED = 1
MLP_PACKED = 1
CONV_DEFER_REDUCE = 2
CONV_DEFER_DW = 16
CONV_SAVE_Z = 4
CONV_BF16 = 8
CONV_R_PAD = 32


def mlp_layout(layout_id):
    """FGC_MLP_LAYOUT(id): the flag bits of an FGC_MLP_PACKED call that name the layout the workspace was packed in."""
    return int(layout_id) << 8


_SIGS = {
    "fgc_last_error": (C.c_char_p, []),
    "fgc_version": (C.c_int, []),
    "fgc_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "fgc_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    "fgc_option_count": (C.c_int32, []),
    "fgc_option_name": (C.c_char_p, [C.c_int32]),
    "fgc_struct_size": (C.c_size_t, [C.c_int32]),
    "fgc_crc32c": (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_size_t]),
    "fgc_profile_enable": (C.c_int, [C.c_int]),
    "fgc_profile_tag": (C.c_int, [C.c_char_p]),
    "fgc_profile_collect": (C.c_int, [C.c_char_p, C.c_int32]),
    "fgc_csr_from_klist": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgc_klist_from_csr": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "fgc_csr_transpose": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgc_pair_graph": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgc_face_features": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "fgc_faces_large_adj": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "fgc_graph_patch": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgc_mesh_patch": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                 C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "fgc_vertices_faces": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "fgc_face_centers": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "fgc_pool4_avg_iz": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "fgc_vertex_update_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    "fgc_edge_map": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgc_vertex_update": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                    C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "fgc_metis_one_level": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_int32, C.c_void_p, C.c_void_p]),
    "fgc_hierarchy_build": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                      C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgc_hierarchy_free": (None, [C.c_void_p]),
    "fgc_hierarchy_size": (C.c_int32, [C.c_void_p, C.c_int32]),
    "fgc_hierarchy_real_size": (C.c_int32, [C.c_void_p, C.c_int32]),
    "fgc_hierarchy_new_to_old": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "fgc_hierarchy_parents": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "fgc_hierarchy_klist": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "fgc_conv_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "fgc_conv_fwd": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                               C.c_void_p]),
    "fgc_conv_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "fgc_conv_bwd_needs_exchange": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvBwdIO)]),
    "fgc_conv_r_ld": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32]),
    "fgc_conv_uses_pairs": (C.c_int, [C.POINTER(ConvDesc)]),
    "fgc_conv_pairs_allowed": (C.c_int, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "fgc_conv_layout_id": (C.c_uint64, [C.POINTER(ConvDesc)]),
    "fgc_mlp_layout_id": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "fgc_conv_bwd": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvBwdIO), C.c_void_p, C.c_size_t, C.c_void_p]),
    "fgc_conv_pack": (C.c_int, [C.POINTER(C.POINTER(ConvDesc)), C.POINTER(C.POINTER(ConvBwdIO)), C.POINTER(C.c_void_p),
                                C.POINTER(C.c_void_p), C.c_int32, C.POINTER(PackExtra), C.c_void_p]),
    "fgc_conv_bwd_reduce": (C.c_int, [C.POINTER(C.POINTER(ConvDesc)), C.POINTER(C.POINTER(ConvBwdIO)),
                                      C.POINTER(C.c_void_p), C.c_int32, C.c_void_p]),
    "fgc_mlp_num_partials": (C.c_int32, [C.c_int32]),
    "fgc_mlp_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "fgc_mlp_bwd_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "fgc_mlp_fwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t,
                              C.c_void_p]),
    "fgc_mlp_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fgc_mlp_bf16_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "fgc_mlp_bwd_bf16_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "fgc_mlp_fwd_bf16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t,
                                   C.c_void_p]),
    "fgc_mlp_bwd_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fgc_pool4_bwd_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p]),
    "fgc_lrelu_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "fgc_lrelu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    "fgc_pool4_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "fgc_pool4_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                C.c_void_p]),
    "fgc_upsample4_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "fgc_upsample4_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "fgc_norm_num_partials": (C.c_int32, [C.c_int32]),
    "fgc_normalize_fwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "fgc_normalize_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgc_normalize_apply": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgc_normalize_bwd_partial": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "fgc_normalize_bwd_apply": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgc_angular_loss_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "fgc_angular_loss_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_float, C.c_void_p, C.c_void_p]),
    "fgc_pool_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "fgc_pool_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_void_p]),
    "fgc_upsample_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "fgc_upsample_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "fgc_lin_bwd_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "fgc_lin_fwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgc_lin_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "fgc_loss_shard_floats": (C.c_int32, [C.c_int32]),
    "fgc_loss_shard_abs_sum": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "fgc_loss_shard_samples": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgc_loss_shard_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "fgc_loss_step_scratch_floats": (C.c_int32, [C.c_int32]),
    "fgc_loss_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "fgc_rotate_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "fgc_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float,
                                C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "fgc_infer_epilogue": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "fgc_gather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "fgc_copy_rows_jobs": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "fgc_scatter_add_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
}

EXPORTS = tuple(_SIGS.keys())

_lib = None


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libfgc.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C facet_graph_convolution_amd/csrc`); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if os.environ.get("FGC_DEV_PARTIAL") and not hasattr(L, name):
                continue  # developer-only: library under construction
            fn = getattr(L, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        # a library built from another header would take the arguments below in another order: refuse it here
        if L.fgc_version() != ABI_VERSION:
            msg = ("libfgc.so has ABI version %d, this binding was written for %d (include/fgc.h: FGC_ABI_VERSION)"
                   % (L.fgc_version(), ABI_VERSION))
            # (developer builds of OTHER commits through FGC_LIB, tools/build_at_commit.sh: the struct sizes below still
            #  have to match; the Python tree should be checked out at the library's commit)
            if not os.environ.get("FGC_DEV_PARTIAL"):
                raise RuntimeError(msg)
            import warnings
            warnings.warn(msg + " - FGC_DEV_PARTIAL is set: going on")
        for which, mirror in ((0, ConvDesc), (1, ConvBwdIO), (2, PackExtra)):
            if L.fgc_struct_size(which) != C.sizeof(mirror):
                raise RuntimeError("libfgc.so: struct %d is %d bytes, the ctypes mirror %s %d" %
                                   (which, L.fgc_struct_size(which), mirror.__name__, C.sizeof(mirror)))
        _lib = L
    return _lib


def set_option(name, value):
    """fgc_set_option: a process-level switch of the library (include/fgc.h; names with or without 'FGC_').  The library reads
    FGC_<NAME> environment variables once, as initial values; afterwards this call is the only way to change one."""
    check(lib().fgc_set_option(name.encode(), int(value)), "fgc_set_option")


def get_option(name):
    v = C.c_int64(0)
    check(lib().fgc_get_option(name.encode(), C.byref(v)), "fgc_get_option")
    return v.value


def option_names():
    L = lib()
    return [L.fgc_option_name(i).decode() for i in range(L.fgc_option_count())]


class options:
    """with _lib.options(NO_PAIRS=1, W8_DATA16_MIN_N=0): ...  - set, then restore on exit."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *a):
        for k, v in self.old.items():
            set_option(k, v)


def check(rc, what="libfgc"):
    if rc != 0:
        msg = lib().fgc_last_error()
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device (or host) address of a tensor; None -> NULL."""
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr())
