"""ctypes binding of `libhermnet_hip.so` (C ABI in `include/hermnet_hip.h`).

The library is built in-tree by `__graft_entry__.build()` / `make -C hermnet_amd/csrc`.
There is NO fallback: if it is missing the product path raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HERMNET_LIB_PATH") or os.path.join(_HERE, "csrc", "libhermnet_hip.so")

HN_ENV = {"polynomial": 0, "exponential": 1}
_ERR = {1: "HN_ERR_BAD_ARG (unsupported shape or null pointer)",
        2: "HN_ERR_LDS (weight tile does not fit the 160 KiB LDS)",
        3: "HN_ERR_LAUNCH (kernel launch failed)"}

c_fp = ctypes.c_void_p  # device pointers travel as integers
ABI_VERSION = 13         # must equal hermnet_abi_version() of the loaded library (include/hermnet_hip.h)


class RbfDesc(ctypes.Structure):
    _fields_ = [("offset", c_fp), ("num_rbf", ctypes.c_int), ("inv_rc", ctypes.c_float),
                ("coeff", ctypes.c_float), ("env_kind", ctypes.c_int), ("env_p", ctypes.c_int)]


class Graph(ctypes.Structure):
    _fields_ = [("num_nodes", ctypes.c_int), ("num_edges", ctypes.c_int), ("num_rel", ctypes.c_int),
                ("type_rowptr", c_fp), ("csr_rowptr", c_fp), ("csr_src", c_fp),
                ("csc_rowptr", c_fp), ("csc_tgt", c_fp), ("csc_pos", c_fp),
                ("num_src", ctypes.c_int), ("res_row", c_fp)]


class PendingGrads(ctypes.Structure):
    """hn_pending_grads (include/hermnet_hip.h): the incoming gradients of an update backward, still in partial sums."""
    _fields_ = [(n, c_fp) for n in ("gn_parts", "gvec_parts", "x", "mean", "rstd", "gx1", "gvec1")] + \
               [("num_parts", ctypes.c_int), ("hidden_real", ctypes.c_int)] + \
               [(n, c_fp) for n in ("gxh", "hb", "w2t_frag16", "w1t_frag16")]


class RelationsOut(ctypes.Structure):
    _fields_ = [(n, c_fp) for n in ("node_order", "row_of_node", "z_rows", "row_real", "row_active", "csr_rowptr",
                                    "csr_src", "csr_perm", "src_id", "tgt_id", "shift_csr", "csc_rowptr", "csc_tgt",
                                    "csc_pos", "out_rowptr", "out_edges")]


# name -> (restype, argtypes); mirrors include/hermnet_hip.h one to one
SIGNATURES = {
    "hermnet_abi_version": (ctypes.c_int, []),
    "hermnet_build_info": (ctypes.c_char_p, []),
    "hermnet_edge_geometry_fwd": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp]),
    "hermnet_edge_geometry_bwd": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp]),
    "hermnet_edge_geometry_bwd_csc": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp]),
    "hermnet_message_scatter_fwd": (ctypes.c_int, [ctypes.POINTER(Graph), ctypes.POINTER(RbfDesc), ctypes.c_int,
                                                   c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_message_scatter_bwd": (ctypes.c_int, [ctypes.POINTER(Graph), ctypes.POINTER(RbfDesc), ctypes.c_int,
                                                   c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                                                   c_fp, c_fp, c_fp, c_fp, ctypes.c_int, c_fp, c_fp,
                                                   c_fp, c_fp, ctypes.c_int, c_fp]),
    "hermnet_edge_radial_table": (ctypes.c_int, [ctypes.POINTER(Graph), ctypes.POINTER(RbfDesc), c_fp, c_fp, c_fp]),
    "hermnet_neighbor_workspace": (ctypes.c_size_t, [ctypes.c_int]),
    "hermnet_neighbor_workspace_for": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "hermnet_neighbor_count": (ctypes.c_int, [c_fp, ctypes.c_int, c_fp, c_fp, c_fp, ctypes.c_double, c_fp,
                                              ctypes.c_size_t, c_fp, c_fp, c_fp]),
    "hermnet_neighbor_fill": (ctypes.c_int, [c_fp, ctypes.c_int, c_fp, c_fp, c_fp, ctypes.c_double, c_fp, ctypes.c_size_t,
                                             ctypes.c_long, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                             c_fp, c_fp, c_fp, c_fp, c_fp]),
    "hermnet_neighbor_fill_padded": (ctypes.c_int, [ctypes.c_int, c_fp, ctypes.c_size_t, ctypes.c_long, ctypes.c_float,
                                                    ctypes.c_int, c_fp, c_fp, c_fp, c_fp]),
    "hermnet_relation_counts": (ctypes.c_int, [c_fp, ctypes.c_int, c_fp, ctypes.c_int, c_fp, c_fp]),
    "hermnet_build_relations_workspace": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "hermnet_build_relations": (ctypes.c_int, [c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, ctypes.c_int,
                                               c_fp, ctypes.c_int, c_fp, ctypes.POINTER(RelationsOut), ctypes.c_int, c_fp,
                                               ctypes.c_size_t, c_fp]),
    "hermnet_build_triadic_workspace": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "hermnet_build_triadic": (ctypes.c_int, [c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, ctypes.c_int, ctypes.c_int,
                                             c_fp, c_fp, ctypes.POINTER(RelationsOut), c_fp, c_fp, ctypes.c_int, c_fp,
                                             ctypes.c_size_t, c_fp]),
    "hermnet_train_node_op": (ctypes.c_int, [ctypes.c_int, c_fp, ctypes.c_int, c_fp, ctypes.c_int, ctypes.c_long, ctypes.c_int,
                                             ctypes.c_float, ctypes.c_float, c_fp]),
    "hermnet_edge_message_fwd_rows": (ctypes.c_int, [c_fp] * 4 + [ctypes.c_long, ctypes.c_int] + [c_fp] * 5 + [ctypes.c_long] +
                                      [c_fp] * 3),
    "hermnet_edge_message_bwd_rows": (ctypes.c_int, [c_fp] * 6 + [ctypes.c_long, ctypes.c_int] + [c_fp] * 6 + [ctypes.c_long] +
                                      [c_fp] * 5),
    "hermnet_edge_message_bwd2_rows": (ctypes.c_int, [c_fp] * 10 + [ctypes.c_long, ctypes.c_int] + [c_fp] * 6 + [ctypes.c_long] +
                                       [c_fp] * 7),
    "hermnet_segment_sum": (ctypes.c_int, [c_fp, c_fp, c_fp, ctypes.c_long, ctypes.c_int, c_fp, c_fp]),
    "hermnet_edge_message_fwd": (ctypes.c_int, [c_fp] * 4 + [ctypes.c_long, ctypes.c_int] + [c_fp] * 6),
    "hermnet_edge_message_bwd": (ctypes.c_int, [c_fp] * 6 + [ctypes.c_long, ctypes.c_int] + [c_fp] * 9),
    "hermnet_edge_message_bwd2": (ctypes.c_int, [c_fp] * 10 + [ctypes.c_long, ctypes.c_int] + [c_fp] * 11),
    "hermnet_ssilu_fwd": (ctypes.c_int, [c_fp, c_fp, ctypes.c_int, c_fp, ctypes.c_long, ctypes.c_int, c_fp]),
    "hermnet_ssilu_bwd": (ctypes.c_int, [c_fp, c_fp, c_fp, ctypes.c_int, c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_long, ctypes.c_long, c_fp]),
    "hermnet_layernorm_fwd": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_float, c_fp]),
    "hermnet_layernorm_bwd": (ctypes.c_int, [c_fp] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_update_mid": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_update_out": (ctypes.c_int, [c_fp, c_fp, ctypes.c_int] + [c_fp] * 7 +
                           [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_update_out_bwd": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, ctypes.c_int] + [c_fp] * 8 +
                               [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_update_mid_bwd": (ctypes.c_int, [c_fp] * 6 + [ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_energy_head_fwd": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_energy_head_bwd": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_energy_head_fused_fwd": (ctypes.c_int, [c_fp] * 8 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_energy_head_fused_bwd": (ctypes.c_int, [c_fp] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_energy_head16_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "hermnet_energy_head16_fwd": (ctypes.c_int, [c_fp] * 8 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_energy_head16_bwd": (ctypes.c_int, [c_fp] * 6 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_node_chain_supported": (ctypes.c_int, [ctypes.c_int]),
    "hermnet_node_chain_tile_rows": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "hermnet_node_pre_fwd": (ctypes.c_int, [c_fp] * 10 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                        ctypes.c_float, c_fp, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_node_pre_bwd": (ctypes.c_int, [c_fp] * 11 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp,
                                                        ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_node_update_tile_rows": (ctypes.c_int, [c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "hermnet_node_update_fwd": (ctypes.c_int, [c_fp] * 16 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_node_update_bwd": (ctypes.c_int, [c_fp] * 14 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp,
                                                           c_fp]),
    "hermnet_node_fused_supported": (ctypes.c_int, [ctypes.c_int]),
    "hermnet_node_update_pre_fwd": (ctypes.c_int, [c_fp] * 16 + [ctypes.c_int, ctypes.c_int, ctypes.c_int] + [c_fp] * 8 +
                                    [ctypes.c_int, ctypes.c_int, ctypes.c_float, c_fp]),
    "hermnet_node_pre_fwd16": (ctypes.c_int, [c_fp] * 9 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                          ctypes.c_float, c_fp]),
    "hermnet_node_pre_bwd16": (ctypes.c_int, [c_fp] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_pair_mean": (ctypes.c_int, [ctypes.c_int, c_fp, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, c_fp, ctypes.c_int, c_fp]),
    "hermnet_halo_rows": (ctypes.c_int, [ctypes.c_int, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp]),
    "hermnet_halo_accumulate": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp]),
    "hermnet_halo_proj_rows": (ctypes.c_int, [ctypes.c_int, c_fp, ctypes.c_long, ctypes.c_int, c_fp, ctypes.c_long, ctypes.c_int,
                                              c_fp, ctypes.c_int, ctypes.c_int, c_fp, c_fp]),
    "hermnet_halo_proj_accumulate": (ctypes.c_int, [c_fp, ctypes.c_long, ctypes.c_int, c_fp, c_fp, c_fp, c_fp, ctypes.c_int,
                                                    ctypes.c_int, c_fp, c_fp]),
    "hermnet_stream_copy": (ctypes.c_int, [c_fp, c_fp, ctypes.c_size_t, ctypes.c_int, c_fp]),
    "hermnet_set_option": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "hermnet_get_option": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "hermnet_weight_fragments": (ctypes.c_int, [c_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_fp]),
    "hermnet_band_product_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "hermnet_band_product": (ctypes.c_int, [c_fp, c_fp, c_fp, ctypes.c_long, ctypes.c_int, ctypes.c_int, c_fp, c_fp]),
    "hermnet_band_product_grad_a": (ctypes.c_int, [c_fp, c_fp, c_fp, ctypes.c_long, ctypes.c_int, ctypes.c_int, c_fp, c_fp]),
    "hermnet_band_product_grad_b": (ctypes.c_int, [c_fp, c_fp, c_fp, ctypes.c_long, ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp]),
    "hermnet_col_sum": (ctypes.c_int, [c_fp, ctypes.c_long, ctypes.c_long, ctypes.c_int, ctypes.c_int, c_fp, c_fp]),
    "hermnet_edge_unit": (ctypes.c_int, [ctypes.c_int, c_fp, c_fp, c_fp, c_fp, ctypes.c_long, c_fp, c_fp, c_fp, c_fp]),
    "hermnet_basis_window": (ctypes.c_int, [ctypes.c_int, c_fp, c_fp, ctypes.c_long, c_fp, c_fp, ctypes.c_long, ctypes.c_int,
                                            ctypes.c_float, ctypes.c_int, c_fp, c_fp, c_fp, c_fp, c_fp]),
    "hermnet_band_product_grads": (ctypes.c_int, [c_fp, c_fp, c_fp, c_fp, ctypes.c_long, ctypes.c_int, ctypes.c_int, c_fp, c_fp, c_fp,
                                                  c_fp]),
    "hermnet_param_guard": (ctypes.c_int, [c_fp, c_fp, ctypes.c_int, c_fp, ctypes.c_int, c_fp, c_fp, c_fp]),
    "hermnet_shard_step_flags": (ctypes.c_int, [c_fp, ctypes.c_long, c_fp, ctypes.c_int, c_fp, ctypes.c_long, c_fp, c_fp, c_fp,
                                                ctypes.c_long, ctypes.c_float, c_fp]),
    "hermnet_host_rbf_row": (ctypes.c_int, [c_fp, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                            ctypes.c_int, c_fp, c_fp, ctypes.c_int, ctypes.c_float, c_fp, c_fp]),
}

_lib = None


def source_hash():
    """sha256 (first 16 hex digits) over the library's sources, as `csrc/Makefile` stamps it into
    `hermnet_build_info()` (`src=...`): *.hip, *.cpp, *.h of csrc/ in sorted order, then include/hermnet_hip.h."""
    import glob
    import hashlib
    csrc = os.path.join(_HERE, "csrc")
    files = sorted(os.path.basename(f) for pat in ("*.hip", "*.cpp", "*.h") for f in glob.glob(os.path.join(csrc, pat)))
    paths = [os.path.join(csrc, f) for f in files] + [os.path.join(_HERE, "..", "include", "hermnet_hip.h")]
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_info():
    """`hermnet_build_info()` of the loaded library (ABI, target, source hash, build date) as a str."""
    return load().hermnet_build_info().decode()


def load():
    """Load the native library (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "hermnet_amd: native library %s not found -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C hermnet_amd/csrc`).  There is no CPU/eager fallback for the hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.hermnet_abi_version() != ABI_VERSION:
        raise RuntimeError("hermnet_amd: %s has ABI version %d, the Python side expects %d -- rebuild it "
                           "(`make -C hermnet_amd/csrc`)" % (LIB_PATH, lib.hermnet_abi_version(), ABI_VERSION))
    # the library must have been built from the sources next to it: a stale prebuilt binary would otherwise be what
    # the tests and the benchmark measure (an explicitly chosen diagnostic build is not checked)
    if not os.environ.get("HERMNET_LIB_PATH") and os.environ.get("HERMNET_ALLOW_STALE_LIB", "0") == "0":
        info = lib.hermnet_build_info().decode()
        try:
            want = source_hash()
        except OSError:
            want = None                       # sources not shipped: nothing to compare with
        if want is not None and ("src=" + want) not in info:
            raise RuntimeError("hermnet_amd: %s was not built from the sources in hermnet_amd/csrc (library: %s; sources "
                               "hash to %s) -- rebuild it (`make -C hermnet_amd/csrc`)" % (LIB_PATH, info, want))
    _lib = lib
    return lib


# hermnet_set_option / hermnet_get_option (include/hermnet_hip.h: HN_OPT_*): tuning knobs and the alternative kernel forms
# that tests and A/Bs compare against -- process-wide, read by the launchers at every call
OPTIONS = {"fwd_variant": 0, "fwd_variant_l0": 1, "bwd_variant": 2, "bwd_variant_l0": 3, "fwd_rows": 4, "bwd_rows": 5,
           "bwd_cl_rows": 6, "bwd_lanes16": 7, "node_chain_wide": 8, "update_tile16": 9, "update_tile64_max": 10}


def get_option(name):
    v = ctypes.c_int(0)
    check(load().hermnet_get_option(OPTIONS[name], ctypes.byref(v)), "hermnet_get_option")
    return int(v.value)


def set_option(name, value):
    """Set a library option; returns the previous value."""
    old = get_option(name)
    check(load().hermnet_set_option(OPTIONS[name], int(value)), "hermnet_set_option")
    return old


class options(object):
    """`with _lib.options(bwd_lanes16=1): ...` -- options set for a block, restored behind it (tests, A/Bs)."""

    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %s" % (what, _ERR.get(rc, "error %d" % rc)))


def ptr(t):
    """Device/host pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
