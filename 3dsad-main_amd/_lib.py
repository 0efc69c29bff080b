"""ctypes binding of ``csrc/libsad_amd.so`` (C-ABI declared in ``include/sad_amd.h``).

No reference FFI exists to mirror (``/root/reference/README.md:1-2`` is the whole upstream
repository); the entry points are the ones BASELINE.json ``north_star`` names.  Loading fails loudly
when the library is missing — there is no CPU fallback anywhere in this package.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# SAD_AMD_LIB points at an alternative build of the same C-ABI (A/B measurements of kernel variants)
SO_PATH = os.environ.get("SAD_AMD_LIB") or os.path.join(CSRC, "libsad_amd.so")

MAX_LAYERS = 4
MAX_RADII = 4
ABI_VERSION = 4            # SAD_ABI_VERSION of include/sad_amd.h this binding was written against
# instrumentation ints of a row-packing table (include/sad_amd.h, SAD_WS_*)
WS_REFILLS, WS_INUSE, WS_CONFLICT = 5, 6, 7

c_f32p = ctypes.POINTER(ctypes.c_float)
c_i32p = ctypes.POINTER(ctypes.c_int32)
vp = ctypes.c_void_p


class MlpArgs(ctypes.Structure):
    """``struct sad_mlp_args`` (include/sad_amd.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_size_t),
        ("xyz", vp), ("new_xyz", vp), ("idx", vp), ("cnt", vp), ("workspace", vp), ("feat", vp),
        ("ld_feat", ctypes.c_int),
        ("B", ctypes.c_int), ("N", ctypes.c_int), ("M", ctypes.c_int), ("S", ctypes.c_int),
        ("C", ctypes.c_int),
        ("L", ctypes.c_int), ("dims", ctypes.c_int * (MAX_LAYERS + 1)),
        ("packed", vp), ("relu_mask", ctypes.c_int),
        ("out", vp), ("ld_out", ctypes.c_int), ("col_off", ctypes.c_int),
        ("geometry", ctypes.c_int),
        ("scratch", vp), ("scratch_bytes", ctypes.c_size_t),
        ("prescanned", ctypes.c_int),
        ("c_out", ctypes.c_int),           # ABI 3: != 0 = a chain packed zero-padded onto wider dims (in the ABI-2 struct's tail padding)
    ]


class MlpBf16Args(ctypes.Structure):
    """``struct sad_mlp_bf16_args`` (include/sad_amd.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_size_t),
        ("xyz", vp), ("new_xyz", vp), ("idx", vp), ("feat", vp),
        ("feat_bf16", ctypes.c_int), ("ld_feat", ctypes.c_int),
        ("B", ctypes.c_int), ("N", ctypes.c_int), ("M", ctypes.c_int), ("S", ctypes.c_int),
        ("C", ctypes.c_int),
        ("L", ctypes.c_int), ("dims", ctypes.c_int * (MAX_LAYERS + 1)),
        ("packed", vp), ("relu_mask", ctypes.c_int),
        ("out", vp), ("out_bf16", ctypes.c_int), ("ld_out", ctypes.c_int), ("col_off", ctypes.c_int),
        ("cnt", vp), ("workspace", vp), ("geometry", ctypes.c_int), ("prescanned", ctypes.c_int),
        # ABI 4, split pooling: continuation rows of a grouped chain / the pooled chains behind a plain layer's input rows
        ("cont", vp), ("n_pool", ctypes.c_int),
        ("pool_ws", vp * MAX_RADII), ("pool_cont", vp * MAX_RADII),
        ("pool_S", ctypes.c_int * MAX_RADII), ("pool_cols", ctypes.c_int * MAX_RADII),
    ]


# name -> (restype, argtypes); every symbol include/sad_amd.h declares
SIGNATURES = {
    "sad_version": (ctypes.c_int, []),
    "sad_last_error": (ctypes.c_char_p, []),
    "sad_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "sad_fps_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "sad_fps_f32": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    "sad_ffps_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "sad_pairdist_f32": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_float, vp, vp]),
    "sad_ffps_f32": (ctypes.c_int, [vp, vp] + [ctypes.c_int] * 5 + [ctypes.c_float, vp, vp, vp]),
    "sad_gather_xyz_f32": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "sad_gather_points": (ctypes.c_int, [vp, vp] + [ctypes.c_int] * 5 + [vp, vp]),
    "sad_group_points": (ctypes.c_int, [vp, vp] + [ctypes.c_int] * 6 + [vp, vp]),
    "sad_group_points_grad_f32": (ctypes.c_int, [vp, vp] + [ctypes.c_int] * 5 + [vp, vp]),
    "sad_group_points_grad_pm_f32": (ctypes.c_int, [vp, vp] + [ctypes.c_int] * 5 + [vp, vp]),
    "sad_max_pool_s_f32": (ctypes.c_int, [vp] + [ctypes.c_int] * 4 + [vp, vp, vp]),
    "sad_max_pool_s_grad_f32": (ctypes.c_int, [vp, vp] + [ctypes.c_int] * 4 + [vp, vp]),
    "sad_ball_query_f32": (ctypes.c_int, [vp, vp, ctypes.c_float, vp] + [ctypes.c_int] * 4 + [vp, vp]),
    "sad_ball_query_multi_f32": (ctypes.c_int, [vp, vp, ctypes.c_int, c_f32p, vp,
                                               ctypes.POINTER(ctypes.c_int), ctypes.POINTER(vp),
                                               ctypes.POINTER(vp),
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]),
    "sad_ball_query_grid_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "sad_ball_query_grid_f32": (ctypes.c_int, [vp, vp, ctypes.c_int, c_f32p, ctypes.POINTER(ctypes.c_int),
                                              ctypes.POINTER(vp), ctypes.POINTER(vp),
                                              ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]),
    "sad_subsample_pad_f32": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint, vp, vp]),
    "sad_copy_rows_u32": (ctypes.c_int, [vp, ctypes.c_longlong, vp, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, vp]),
    "sad_knn_f32": (ctypes.c_int, [vp, vp] + [ctypes.c_int] * 4 + [vp, vp]),
    "sad_mlp_packed_floats": (ctypes.c_size_t, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "sad_mlp_pack_f32": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                       ctypes.POINTER(vp), ctypes.POINTER(vp), vp, vp]),
    "sad_mlp_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "sad_mlp_rowscan": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int),
                                       ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp), vp]),
    "sad_mlp_rowscan_init": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int),
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                            ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), vp]),
    "sad_mlp_rowscan_split": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_int),
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                             ctypes.POINTER(ctypes.c_int), vp]),
    "sad_mlp_cont_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "sad_mlp_preferred_geometry": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "sad_mlp_padded_dims": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "sad_mlp_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.POINTER(ctypes.c_int)]),
    "sad_mlp_chain_f32": (ctypes.c_int, [ctypes.POINTER(MlpArgs), vp]),
    "sad_mlp_chain_multi_f32": (ctypes.c_int, [ctypes.POINTER(ctypes.POINTER(MlpArgs)), ctypes.c_int, vp]),
    "sad_mlp_packed_bytes_bf16": (ctypes.c_size_t, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int]),
    "sad_mlp_pack_bf16": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                        ctypes.POINTER(vp), ctypes.POINTER(vp), vp, vp]),
    "sad_mlp_preferred_geometry_bf16": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]),
    "sad_mlp_chain_bf16": (ctypes.c_int, [ctypes.POINTER(MlpBf16Args), vp]),
    "sad_mlp_chain_multi_bf16": (ctypes.c_int, [ctypes.POINTER(ctypes.POINTER(MlpBf16Args)), ctypes.c_int, vp]),
    "sad_candidates_f32": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_float, ctypes.c_float, ctypes.c_float, c_f32p,
                                         vp, vp, vp]),
    "sad_nms_bev_f32": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                      vp, vp, vp, vp]),
    "sad_nms_bev_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "sad_nms_bev_ws_f32": (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                         vp, vp, vp, vp, vp]),
    "sad_decode_boxes_f32": (ctypes.c_int, [vp, vp, ctypes.c_int, ctypes.c_int, c_f32p, vp, vp]),
}

_lib = None

# A step plan being recorded on this thread (plan.Recorder, else None): lib() then hands out a stand-in that passes every
# launch through and appends it to the plan (3dsad-main_amd/plan.py).  Thread-local: another thread's launches are its own.
import threading
_tls = threading.local()


def recorder():
    return getattr(_tls, "rec", None)


def set_recorder(rec) -> None:
    _tls.rec = rec


def is_launch(name: str) -> bool:
    """Does this entry point enqueue device work (last argument: the stream)?  Size queries, options and the version do not."""
    return (name.startswith("sad_") and not name.endswith(("_bytes", "_floats", "_bytes_bf16"))
            and "preferred_geometry" not in name and name not in ("sad_version", "sad_last_error", "sad_set_option", "sad_mlp_padded_dims"))


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 with hipcc (csrc/Makefile).  Works without a GPU."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return SO_PATH


def lib():
    """The loaded library.  Raises if it has not been built — the HIP path is the only path."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc, gfx950).  sad_amd has no CPU fallback.")
        # torch first: it brings its own HIP runtime (libamdhip64), and libsad_amd.so must bind to
        # THAT copy — loaded the other way round the process ends up with the system runtime under
        # torch's allocator and device-to-device copies of torch memory fail
        import torch  # noqa: F401
        handle = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if handle.sad_version() != ABI_VERSION:
            raise RuntimeError(f"{SO_PATH}: ABI version {handle.sad_version()}, this binding needs {ABI_VERSION} "
                               "(stale build: run __graft_entry__.build())")
        _lib = handle
    rec = getattr(_tls, "rec", None)
    return _lib if rec is None else rec.lib


def check(code: int, what: str) -> None:
    """Turn a negative SAD_E* return code into a RuntimeError carrying sad_last_error()."""
    if code != 0:
        msg = lib().sad_last_error()
        raise RuntimeError(f"{what} failed ({code}): {msg.decode() if msg else ''}")


def set_option(key: str, value: int) -> None:
    check(lib().sad_set_option(key.encode(), int(value)), "sad_set_option")
