"""ctypes binding of the C-ABI hot-path library (include/goslam_hip.h).

The library is built in-tree (`go_slam_amd/csrc/libgoslam_hip.so`, see csrc/Makefile and
`__graft_entry__.build`).  There is NO fallback: if the shared object is missing or a call
fails, a RuntimeError is raised -- a CPU/PyTorch substitute would silently void every parity
and performance claim made for this package.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# (GOSLAM_HIP_LIB: another build of the same ABI, e.g. the previous kernel for a same-box A/B in tools/)
LIB_PATH = os.environ.get("GOSLAM_HIP_LIB") or os.path.join(CSRC, "libgoslam_hip.so")

_lib = None

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/goslam_hip.h one-to-one
_P = c_void_p
SIGNATURES = {
    "gs_version": (ctypes.c_char_p, []),
    "gs_last_error": (ctypes.c_char_p, []),
    "gs_enc_conv_wpack_elems": (c_size_t, [c_int, c_int, c_int]),
    "gs_enc_conv_stat_chunks": (c_int, [c_int, c_int, c_int]),
    "gs_enc_conv": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "gs_norm_act_workspace_bytes_chunks": (c_size_t, [c_int, c_int, c_int]),
    "gs_timing_begin": (c_int, [_P]),
    "gs_timing_end": (c_int, []),
    "gs_timing_read": (c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int)]),
    "gs_timing_names": (c_int, [ctypes.c_char_p, c_int]),
    "gs_corr_index_forward": (c_int, [_P, _P, _P] + [c_int] * 7 + [_P]),
    "gs_corr_index_backward": (c_int, [_P, _P, _P] + [c_int] * 7 + [_P]),
    "gs_corr_lookup_pyramid": (c_int, [_P] * 6 + [c_int] * 9 + [_P]),
    "gs_corr_lookup_enc": (c_int, [_P] * 8 + [c_int] * 7 + [_P]),
    "gs_corr_lookup_enc_slots": (c_int, [_P] * 9 + [c_int] * 7 + [_P]),
    "gs_corr_lookup_pyramid_slots": (c_int, [_P] * 7 + [c_int] * 9 + [_P]),
    "gs_corr_volume_pyramid_slots": (c_int, [_P] * 7 + [c_int] * 5 + [_P, c_size_t, _P]),
    "gs_corr_volume_workspace_bytes": (c_size_t, [c_int] * 4),
    "gs_corr_volume_pyramid": (c_int, [_P] * 6 + [c_int] * 5 + [_P, c_size_t, _P]),
    "gs_corr_level_elems": (c_size_t, [c_int] * 4),
    "gs_altcorr_backward": (c_int, [_P] * 6 + [c_int] * 8 + [_P]),
    "gs_altcorr_forward": (c_int, [_P] * 4 + [c_int] * 9 + [_P]),
    "gs_altcorr_pyramid": (c_int, [_P] * 8 + [c_int] * 5 + [_P]),
    "gs_reproject": (c_int, [_P] * 7 + [c_int] * 3 + [_P]),
    "gs_projmap": (c_int, [_P] * 7 + [c_int] * 3 + [_P]),
    "gs_frame_distance": (c_int, [_P] * 6 + [c_int] * 3 + [c_float, _P]),
    "gs_iproj": (c_int, [_P] * 4 + [c_int] * 3 + [_P]),
    "gs_depth_filter": (c_int, [_P] * 6 + [c_int] * 4 + [_P]),
    "gs_cvx_upsample": (c_int, [_P] * 4 + [c_int] * 4 + [_P]),
    "gs_upmask_upsample": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "gs_bias_act": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "gs_motion_features": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "gs_ba_inputs": (c_int, [_P] * 6 + [c_int, c_int, c_int, _P]),
    "gs_lowmem_gather": (c_int, [_P] * 7 + [c_int, c_int, c_int, _P]),
    "gs_lowmem_scatter": (c_int, [_P] * 8 + [c_int, c_int, c_int, _P]),
    "gs_damping_rows": (c_int, [_P] * 5 + [c_int, c_int, c_float, c_float, _P]),
    "gs_conv1x1": (c_int, [_P, c_int, c_int, _P, _P, c_int, _P, c_int, c_int, ctypes.c_longlong, _P]),
    "gs_conv7x7_c4": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "gs_conv3x3_wpack_elems": (c_size_t, [c_int, c_int]),
    "gs_conv3x3_pp": (c_int, [_P, c_int, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "gs_conv3x3_bias_relu": (c_int, [_P, c_int, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "gs_conv3x3_gru_zr": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "gs_conv3x3_gru_zr2": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "gs_conv3x3_gru_q": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "gs_conv3x3_head": (c_int, [_P, c_int, _P, c_int, _P, _P, c_int, c_int, c_float, _P, c_int, c_int, c_int, _P]),
    "gs_segment_mean": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, _P]),
    "gs_norm_act_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gs_norm_act": (c_int, [_P, _P, _P, _P] + [c_int] * 6 + [c_float, _P, c_size_t, c_int, _P]),
    "gs_gru_glo_workspace_bytes": (c_size_t, [c_int]),
    "gs_gru_glo": (c_int, [_P] * 11 + [c_int, c_int, _P, c_size_t, _P]),
    "gs_gru_glo_fused_workspace_bytes": (c_size_t, [c_int, c_int]),
    "gs_gru_glo_fused": (c_int, [_P, c_int] + [_P] * 10 + [c_int, c_int, _P, c_size_t, _P]),
    "gs_gru_gate_zr": (c_int, [_P] * 6 + [c_int] * 3 + [_P]),
    "gs_gru_gate_q": (c_int, [_P] * 7 + [c_int] * 2 + [_P]),
    "gs_edge_prep": (c_int, [_P] * 4 + [c_int, _P, _P] + [c_int] * 6 + [c_float, c_int, c_int, _P]),
    "gs_edge_greedy": (c_int, [_P] * 5 + [c_int] * 5 + [c_float, c_int, c_int, _P]),
    "gs_ba_workspace_bytes": (c_size_t, [c_int] * 5),
    "gs_ba": (c_int, [_P] * 9 + [c_int] * 3 + [c_float, c_float] + [c_int] * 6 + [_P, _P, _P, _P, c_size_t, _P]),
    "gs_ba_ex": (c_int, [_P] * 9 + [c_int] * 3 + [c_float, c_float] + [c_int] * 6 + [_P, _P, _P, _P, c_size_t, c_int, _P]),
    # include/goslam_neus.h
    "gs_grid_meta_default": (c_int, [_P]),
    "gs_neus_level_major_min_points": (c_int, [c_int]),
    "gs_render_sample": (c_int, [_P] * 7 + [c_float, _P, _P, _P] + [c_int] * 3 + [_P]),
    "gs_grid_encode": (c_int, [_P] * 4 + [c_int, _P]),
    "gs_grid_backward": (c_int, [_P, _P, _P, c_int, c_float, _P, _P, c_int, c_float, _P, _P, c_int, _P]),
    "gs_mlp_workspace_bytes": (c_size_t, [c_int, c_int]),
    "gs_mlp_forward": (c_int, [_P] * 3 + [c_int] * 3 + [_P, c_size_t, _P]),
    "gs_neus_forward_workspace_bytes": (c_size_t, [c_int, c_int]),
    "gs_neus_forward": (c_int, [_P] * 9 + [c_float] + [_P] * 18 + [c_float, _P, c_float, c_int, c_int, _P, c_size_t, _P]),
    "gs_neus_backward_rays": (c_int, [_P] * 13 + [c_int, c_int, _P]),
    "gs_mapping_loss": (c_int, [_P] * 8 + [c_float] * 4 + [c_int] + [_P] * 4 + [c_int, c_int, _P]),
    "gs_map_grad_sqnorm": (c_int, [_P, c_size_t, c_float, _P, c_size_t, _P, _P]),
    "gs_map_adamw": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_float, _P, c_size_t] + [c_float] * 6 + [c_int, _P, c_float, _P]),
    "gs_map_gram_blocks": (c_int, [c_int]),
    "gs_map_gram": (c_int, [_P, c_int, _P, _P]),
    "gs_map_step_prep": (c_int, [_P, c_int, _P, c_float, c_float, c_int] + [_P] * 13),
    "gs_map_step_post": (c_int, [_P, c_int, c_float, _P, c_int, _P, _P, _P, c_float, _P, _P, c_int, c_float, c_int, _P, _P, _P]),
    "gs_map_adamw_seg": (c_int, [_P, _P, _P, _P, _P, c_size_t, c_float, _P, _P, _P, _P, _P, c_size_t] + [c_float] * 6
                         + [c_int, _P, _P, c_float, _P]),
    "gs_mlp_backward_blocks": (c_int, [c_int]),
    "gs_mlp_backward": (c_int, [_P, _P, _P, _P, c_float, _P, _P, c_int, _P]),
    "gs_neus_backward_points": (c_int, [_P] * 7 + [c_float] + [_P] * 9 + [c_int, c_float, _P, _P, c_int, c_float] + [_P] * 5 + [c_int, c_float, c_int, _P, c_int, c_int, _P, _P]),
    "gs_neus_backward_points_binned": (c_int, [_P] * 7 + [c_float] + [_P] * 9 + [c_int, c_float, _P, _P, c_float] + [_P] * 5
                                       + [c_int, c_float, c_int, _P, c_int, c_int, _P, c_size_t, _P, _P, _P]),
    "gs_neus_bin_workspace_bytes": (c_size_t, [c_int]),
    "gs_ray_draw": (c_int, [_P] * 7 + [c_int] * 4 + [c_float] * 4 + [_P] * 5),
}


class GridMeta(ctypes.Structure):
    """gs_grid_meta of include/goslam_neus.h."""
    _fields_ = [("scale", c_float * 16), ("resolution", ctypes.c_uint32 * 16), ("size", ctypes.c_uint32 * 16),
                ("offset", ctypes.c_uint32 * 16), ("hashed", ctypes.c_uint32 * 16), ("total", ctypes.c_uint32)]


def grid_meta():
    m = GridMeta()
    check(lib().gs_grid_meta_default(ctypes.byref(m)), "gs_grid_meta_default")
    return m


def build(verbose=False):
    """Compile libgoslam_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libgoslam_hip.so failed:\n" + res.stdout[-4000:])
    return LIB_PATH


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(go_slam_amd has no CPU fallback)")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError => header/library mismatch, fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().gs_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (status {rc}): {msg}")


class kernel_timer:
    """`with kernel_timer(device) as t: ...; t.read()` -> {launch name: (total ms, launches)} for every kernel the
    library launched on torch's current stream inside the block (gs_timing_*, include/goslam_hip.h)."""

    def __init__(self, device=None):
        self.device = device

    def __enter__(self):
        check(lib().gs_timing_begin(stream_ptr(self.device)), "gs_timing_begin")
        return self

    def __exit__(self, *exc):
        lib().gs_timing_end()
        return False

    def read(self):
        L = lib()
        buf = ctypes.create_string_buffer(1 << 16)
        check(L.gs_timing_names(buf, len(buf)), "gs_timing_names")
        out = {}
        for name in [n for n in buf.value.decode().split(",") if n]:
            ms, cnt = ctypes.c_double(0.0), c_int(0)
            check(L.gs_timing_read(name.encode(), ctypes.byref(ms), ctypes.byref(cnt)), "gs_timing_read")
            out[name] = (ms.value, cnt.value)
        return out


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream on `device` (kernels are enqueued there)."""
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_EMPTY_SENTINEL = {}


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  A zero-element tensor has no storage (data_ptr() == 0), which
    the C side would reject as a NULL argument before it reaches its `n == 0 -> nothing to do` early-out; such
    tensors are represented by a small per-device sentinel buffer that is never dereferenced."""
    if t is None:
        return ctypes.c_void_p(0)
    p = t.data_ptr()
    if p == 0 and t.numel() == 0 and t.is_cuda:
        s = _EMPTY_SENTINEL.get(t.device)
        if s is None:
            import torch
            s = _EMPTY_SENTINEL[t.device] = torch.zeros(64, dtype=torch.uint8, device=t.device)
        p = s.data_ptr()
    return ctypes.c_void_p(p)
