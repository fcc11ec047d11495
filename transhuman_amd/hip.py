"""ctypes binding of libtranshuman_hip.so (include/transhuman_hip.h).

PyTorch is used for device memory, streams and nothing else: every function
here takes CUDA(=HIP) tensors, hands raw device pointers + the current stream
to the C ABI and returns freshly allocated tensors.  There is NO fallback: if
the shared library is missing or a call fails this module raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TH_LIB_PATH: developer override to A/B an experimental build of the same ABI
LIB_PATH = os.environ.get("TH_LIB_PATH") or os.path.join(_HERE, "libtranshuman_hip.so")

c_float_p = C.POINTER(C.c_float)


class ThLinear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("out_f", C.c_int), ("in_f", C.c_int)]


class ThMlpWeights(C.Structure):
    _names = ["fc_0", "alpha_res_0", "key0", "val0", "key1", "val1", "fc_1", "fc_2", "fc_3", "alpha_fc",
              "feature_fc", "rgb_res_0", "view_fc", "rgb_res_1", "fc_4", "rgb_fc", "upsample_color"]
    _fields_ = [(n, ThLinear) for n in _names]


class ThSmplModel(C.Structure):
    _fields_ = [("v_template", C.c_void_p), ("shapedirs", C.c_void_p), ("posedirs", C.c_void_p),
                ("J_regressor", C.c_void_p), ("weights", C.c_void_p), ("parent", C.c_void_p), ("n_verts", C.c_int)]


class ThVitBlock(C.Structure):
    _fields_ = [("ln1_w", C.c_void_p), ("ln1_b", C.c_void_p), ("ln2_w", C.c_void_p), ("ln2_b", C.c_void_p),
                ("qkv", ThLinear), ("proj", ThLinear), ("fc1", ThLinear), ("fc2", ThLinear)]


class ThPoints(C.Structure):
    _fields_ = [("pts", C.c_void_p), ("ray_o", C.c_void_p), ("ray_d", C.c_void_p), ("near", C.c_void_p),
                ("far", C.c_void_p), ("t_vals", C.c_void_p), ("one_minus_t", C.c_void_p), ("R", C.c_int),
                ("S", C.c_int), ("z_vals", C.c_void_p), ("sigma_noise", C.c_void_p)]


class ThFrame(C.Structure):
    _fields_ = [("verts_world", C.c_void_p), ("n_verts", C.c_int), ("Rh", C.c_void_p), ("Th", C.c_void_p),
                ("cams", C.c_void_p), ("scale_xy", C.c_void_p), ("pixel_map_nhwc", C.c_void_p), ("V", C.c_int),
                ("H", C.c_int), ("W", C.c_int), ("map_channels", C.c_int), ("tokens", C.c_void_p), ("centres", C.c_void_p),
                ("rot", C.c_void_p), ("n_clusters", C.c_int), ("hull_thresh", C.c_float),
                ("small_frame_rays", C.c_int), ("map_source", C.c_void_p), ("map_fold", C.c_void_p)]


class ThMapSource(C.Structure):
    _fields_ = [("box", C.c_void_p), ("reach", C.c_float), ("img", C.c_void_p), ("lat0", C.c_void_p),
                ("lat1", C.c_void_p), ("lat2", C.c_void_p), ("dims", C.c_int32 * 6), ("demand", C.c_void_p)]


# every symbol include/transhuman_hip.h declares (tests/test_cabi.py checks the export table)
SYMBOLS = {
    "th_abi_version": (C.c_int, []),
    "th_last_error": (C.c_char_p, []),
    "th_sizeof": (C.c_size_t, [C.c_char_p]),
    "th_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "th_ctx_destroy": (None, [C.c_void_p]),
    "th_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "th_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "th_fused_cycles": (C.c_int, [C.c_void_p, C.c_void_p]),
    "th_host_wait_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "th_clock_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "th_set_mlp_weights": (C.c_int, [C.c_void_p, C.POINTER(ThMlpWeights), C.c_void_p]),
    "th_set_mlp_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "th_set_fused_waves": (C.c_int, [C.c_void_p, C.c_int]),
    "th_set_vit_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "th_set_tok_gather": (C.c_int, [C.c_void_p, C.c_int]),
    "th_set_tex_rows": (C.c_int, [C.c_void_p, C.c_int]),
    "th_map_fold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "th_pixel_texlist_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "th_pixel_texlist": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "th_range_snapshot": (C.c_int, [C.c_void_p, C.c_void_p]),
    "th_range_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]),
    "th_range_last_slot": (C.c_int, [C.c_void_p]),
    "th_set_chunk_samples": (C.c_int, [C.c_int]),
    "th_set_vit_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(ThVitBlock), C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "th_linear_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "th_linear_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(ThLinear), C.c_int,
                                    C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "th_hull_workspace_bytes": (C.c_size_t, [C.c_int]),
    "th_hull_mask": (C.c_int, [C.c_void_p, C.POINTER(ThPoints), C.c_void_p, C.c_int, C.c_float, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "th_paint_group": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "th_segment_mean_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_void_p]),
    "th_segment_mean_rot_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p]),
    "th_upsample_concat_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "th_upsample_concat_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "th_upsample_concat_split_box": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                               C.c_void_p]),
    "th_map_demand_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "th_upsample_concat_split_demand": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p]),
    "th_map_fold_demand": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "th_render_predemand": (C.c_int, [C.c_void_p, C.POINTER(ThFrame), C.POINTER(ThPoints), C.c_void_p, C.c_size_t, C.c_void_p,
                                      C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "th_map_box": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                             C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "th_paint_group_nhwc_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "th_paint_group_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(ThLinear), C.POINTER(ThLinear),
                                      C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "th_vit_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "th_vit_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_size_t, C.c_void_p]),
    "th_dparf_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "th_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p]),
    "th_pixel_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "th_pixel_gather_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "th_network_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "th_network_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    "th_composite": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(ThPoints), C.c_int, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "th_gen_rays": (C.c_int, [C.c_void_p, c_float_p, c_float_p, c_float_p, c_float_p, C.c_int, C.c_int, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "th_bound_mask": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "th_marching_cubes_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "th_marching_cubes_count": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                          C.c_size_t, C.POINTER(C.c_int64), C.c_void_p]),
    "th_marching_cubes_range": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.POINTER(C.c_int64), C.c_void_p]),
    "th_marching_cubes_emit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p,
                                         C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "th_smpl_workspace_bytes": (C.c_size_t, [C.c_int]),
    "th_smpl_lbs": (C.c_int, [C.c_void_p, C.POINTER(ThSmplModel), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "th_view_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "th_render_workspace_bytes": (C.c_size_t, [C.POINTER(ThFrame), C.c_int, C.c_int]),
    "th_shade_pool_bytes": (C.c_size_t, [C.c_void_p, C.POINTER(ThFrame), C.c_longlong, C.c_int]),
    "th_render_prepass_wait": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "th_render_rays": (C.c_int, [C.c_void_p, C.POINTER(ThFrame), C.POINTER(ThPoints), C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                 C.POINTER(C.c_int64), C.c_void_p]),
    "th_render_prepass": (C.c_int, [C.c_void_p, C.POINTER(ThFrame), C.POINTER(ThPoints), C.c_void_p, C.c_size_t,
                                    C.c_void_p]),
    "th_render_pregrid": (C.c_int, [C.c_void_p, C.POINTER(ThFrame), C.POINTER(ThPoints), C.c_void_p, C.c_size_t,
                                    C.c_void_p]),
    "th_render_pregather": (C.c_int, [C.c_void_p, C.POINTER(ThFrame), C.POINTER(ThPoints), C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "th_render_pregather_early": (C.c_int, [C.c_void_p, C.POINTER(ThFrame), C.POINTER(ThPoints), C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "th_conv_pack_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "th_conv_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                               C.POINTER(C.c_float), C.c_void_p]),
    "th_conv2d_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "th_conv2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_int,
                            C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "th_maxpool3x3s2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "th_bn_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "th_bn_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                            C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                            C.c_void_p]),
    "th_bn_act_eval": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "th_conv2d_stats_partials": (C.c_int, [C.c_int] * 7),
    "th_conv2d_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_int,
                                  C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "th_bn_act_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                  C.c_void_p]),
    "th_render_prepass_cancel": (C.c_int, [C.c_void_p]),
    "th_render_prepass_drop": (C.c_int, [C.c_void_p, C.c_void_p]),
    "th_sigma_grid_workspace_bytes": (C.c_size_t, [C.POINTER(ThFrame), C.c_int]),
    "th_eval_sigma_grid": (C.c_int, [C.c_void_p, C.POINTER(ThFrame), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_int64), C.c_void_p]),
}

_lib = None
_ctx = {}


class HipError(RuntimeError):
    pass


def load_library():
    """dlopen the in-tree library and bind prototypes.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipError(f"{LIB_PATH} not found: build it with `python -m transhuman_amd.build` "
                       "(there is no CPU/torch fallback for the rendering hot path)")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.th_abi_version() != 12:
        raise HipError("ABI version mismatch")
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise HipError(_lib.th_last_error().decode())


_ctx_by_device = {}      # torch.device with an explicit index -> handle (the per-call path: no torch.cuda queries)
_have_gpu = None


def _gpu_visible():
    """torch.cuda.is_available(), asked once (it reads environment variables on every call: ~70 calls per frame showed up
    in the host profile, tools/host_cost.py)"""
    global _have_gpu
    if _have_gpu is None:
        _have_gpu = bool(torch.cuda.is_available())
    return _have_gpu


def ctx(device=None):
    h = _ctx_by_device.get(device) if device is not None else None
    if h is not None:
        return h
    lib = load_library()
    if not _gpu_visible():
        raise HipError("no HIP device visible: the TransHuman hot path needs an MI355X (gfx950)")
    dev = torch.cuda.current_device() if device is None else torch.device(device).index or 0
    if dev not in _ctx:
        h = C.c_void_p()
        _check(lib.th_ctx_create(dev, C.byref(h)))
        _ctx[dev] = h
    if isinstance(device, torch.device) and device.index is not None:
        _ctx_by_device[device] = _ctx[dev]
    return _ctx[dev]


def _stream():
    # (torch.cuda.current_stream() builds a Stream object through three Python layers: 4-8 us, ~40 times per frame; the raw
    # handle of the current stream of the current device is one C call)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32(t):
    assert t.is_cuda, "expected a device tensor"
    if t.dtype is torch.float32 and t.is_contiguous():       # (the common case: no new tensor objects)
        return t.detach() if t.requires_grad else t
    return t.detach().to(torch.float32).contiguous()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _linear(w, b):
    w2 = _f32(w).reshape(w.shape[0], -1)
    b2 = _f32(b) if b is not None else None
    keep = (w2, b2)
    return ThLinear(_p(w2), _p(b2), w2.shape[0], w2.shape[1]), keep


# ---------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------
def set_mlp_weights(net):
    """Upload + repack the per-point MLP of a Network (cross_transformer.py:96-126)."""
    lib = load_library()
    keep = []
    W = ThMlpWeights()
    pairs = {"fc_0": net.fc_0, "alpha_res_0": net.alpha_res_0, "key0": net.spatial_key_value_0.key_embed,
             "val0": net.spatial_key_value_0.value_embed, "key1": net.spatial_key_value_1.key_embed,
             "val1": net.spatial_key_value_1.value_embed, "fc_1": net.fc_1, "fc_2": net.fc_2, "fc_3": net.fc_3,
             "alpha_fc": net.alpha_fc, "feature_fc": net.feature_fc, "rgb_res_0": net.rgb_res_0,
             "view_fc": net.view_fc, "rgb_res_1": net.rgb_res_1, "fc_4": net.fc_4, "rgb_fc": net.rgb_fc}
    # optional: the encoder's colour lift (encoder.py:95) -> colour-folded layers for the compact pixel map
    enc = getattr(net, "encoder", None)
    if enc is not None and hasattr(enc, "upsample_color"):
        pairs["upsample_color"] = enc.upsample_color
    for name, mod in pairs.items():
        lin, k = _linear(mod.weight, mod.bias)
        keep.append(k)
        setattr(W, name, lin)
    _check(lib.th_set_mlp_weights(ctx(net.fc_0.weight.device), C.byref(W), _stream()))


def set_vit_weights(vit):
    lib = load_library()
    keep = []
    blocks = (ThVitBlock * len(vit.blocks))()
    for i, blk in enumerate(vit.blocks):
        b = blocks[i]
        for name, t in (("ln1_w", blk.norm1.weight), ("ln1_b", blk.norm1.bias), ("ln2_w", blk.norm2.weight),
                        ("ln2_b", blk.norm2.bias)):
            tt = _f32(t)
            keep.append(tt)
            setattr(b, name, tt.data_ptr())
        for name, mod in (("qkv", blk.attn.qkv), ("proj", blk.attn.proj), ("fc1", blk.mlp.fc1), ("fc2", blk.mlp.fc2)):
            lin, k = _linear(mod.weight, mod.bias)
            keep.append(k)
            setattr(b, name, lin)
    nw, nb = _f32(vit.norm.weight), _f32(vit.norm.bias)
    _check(lib.th_set_vit_weights(ctx(nw.device), len(vit.blocks), vit.embed_dim, vit.num_heads, blocks, _p(nw), _p(nb),
                                  _stream()))


_ctx_owner = {}        # (device index, kind) -> (weakref to the module whose weights the ctx holds, version tuple)
_param_lists = {}      # id(module) -> [weakref(module), parameter list, calls since the last walk]


def _sync_weights(mod, kind):
    """Re-upload when the device context does not hold THIS module's current weights: a th_ctx has one MLP and
    one ViT weight image per device, so ownership is tracked per (device, kind) -- rendering with net A, then
    net B, then A again re-uploads A (a cache keyed by id(module) alone would find A's old version tuple and
    shade A with B's weights; the same for a new module that lands on a freed module's id()).  Any parameter
    change (load_state_dict, .cuda(), optimiser step) bumps the version tuple.
    The check runs every frame, so it is kept cheap: the module tree is walked once (and again every 256 calls,
    in case a Parameter object was replaced), per call only the in-place version counters of the cached
    parameter list and the addresses of its first / last tensor are read (25 us instead of 0.9 ms for the
    310-tensor Network: the walk used to leave the GPU idle in front of the per-sample stage)."""
    import weakref
    if not _gpu_visible():
        raise HipError("no HIP device visible: the TransHuman hot path needs an MI355X (gfx950); there is no CPU fallback")
    ent = _param_lists.get(id(mod))
    if ent is None or ent[0]() is not mod or ent[2] >= 256:
        ent = [weakref.ref(mod), list(mod.parameters()), 0]
        _param_lists[id(mod)] = ent
        if len(_param_lists) > 64:                       # drop entries of modules that are gone
            for k in [k for k, v in _param_lists.items() if v[0]() is None]:
                del _param_lists[k]
    ent[2] += 1
    pl = ent[1]
    ver = (tuple(p._version for p in pl), pl[0].data_ptr(), pl[-1].data_ptr(), len(pl))
    dev = pl[0].device
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), kind)
    own = _ctx_owner.get(key)
    if own is None or own[0]() is not mod or own[1] != ver:
        # The context's packed weight image is rewritten in place on the CURRENT stream.  Frame pipelines call this from a
        # side stream while kernels of earlier frames may still read the image on the shading stream (and the other way
        # round): a weight change is rare (load_state_dict, an optimiser step between evaluations), so it simply drains the
        # device first.
        if own is not None:
            torch.cuda.synchronize(dev)
        (set_mlp_weights if kind == "mlp" else set_vit_weights)(mod)
        _ctx_owner[key] = (weakref.ref(mod), ver)
        if kind == "vit":
            _vit_graph_epoch[0] += 1
        # new weights: back to the modes the caller asked for (the guard re-checks them).  The Network's parameter set
        # includes the encoder, so the stem convolutions return to the HIP kernels with it (th_set_mlp_weights clears
        # their sticky range slot, th_set_vit_weights TransHE's).
        if kind == "mlp":
            if _range_fallback.pop(key[0], None):
                _check(load_library().th_set_mlp_mode(ctx(dev), _user_mode.get(key[0], 1)))
            _conv_fallback.pop(key[0], None)
        elif _vit_fallback.pop(key[0], None):
            _vit_graph_epoch[0] += 1
            _check(load_library().th_set_vit_mode(ctx(dev), _user_vit_mode.get(key[0], 1)))


# ---------------------------------------------------------------------------
# range guard of the fp16 hi/lo split arithmetic (include/transhuman_hip.h: th_range_*)
# ---------------------------------------------------------------------------
RANGE_NAMES = ("f", "s", "p", "n", "inter", "fc4_in", "conv_in", "vit_in")
RANGE_FP16_LIMIT = 0x7B53          # 6.0e4 as an fp16 bit pattern (inf / NaN are larger)
RANGE_FP16_FLOOR = 0x2400          # 2^-6
RANGE_FP32_LIMIT = 0x476A6000      # 6.0e4 as an fp32 bit pattern (slot conv_in)
_user_mode = {}                    # device index -> mode requested through set_mlp_mode (default 1)
_range_fallback = {}               # device index -> True while the guard forces mode 0 on this context
_range_epoch = {}                  # device index -> number of fallbacks so far (frames queued earlier are re-rendered)
_conv_fallback = {}                # device index -> True once the stem convolutions' input left the fp16 range: stock convolutions
_user_vit_mode = {}                # device index -> the TransHE mode the caller asked for (set_vit_mode): restored with new weights
_vit_fallback = {}                 # device index -> True once an operand of TransHE's fp16-split GEMMs left the fp16 range: fp32 MFMA GEMMs
last_range = None                  # the last table read (debugging / tests)


def _dev_index(device):
    return torch.device(device).index if device is not None and torch.device(device).index is not None \
        else torch.cuda.current_device()


def conv_fallback(device=None):
    """True while the stem convolutions of this device run through the stock modules (range guard)."""
    return bool(_conv_fallback.get(_dev_index(device)))


def vit_fallback(device=None):
    """True while TransHE's dense layers of this device run on the fp32 MFMA GEMMs (range guard)."""
    return bool(_vit_fallback.get(_dev_index(device)))


def guard_state(device=None):
    """What the range guard has switched on this device: every entry False = the fast paths are in use (a tripped MLP
    guard means per-layer fp32 MFMA launches, ~7x slower frames)."""
    d = _dev_index(device)
    return {"mlp_fp32_fallback": bool(_range_fallback.get(d)), "conv_fallback": bool(_conv_fallback.get(d)),
            "vit_fp32_fallback": bool(_vit_fallback.get(d)), "epoch": _range_epoch.get(d, 0)}


def range_read(slot, device=None):
    """The launch-wide maxima of snapshot ``slot`` (th_range_read; waits for the work in front of the snapshot).
    None when the snapshot has been overwritten by later ones (the ring holds the 8 most recent)."""
    global last_range
    out = (C.c_uint32 * 8)()
    rc = load_library().th_range_read(ctx(device), int(slot), out)
    if rc == 2:
        return None
    _check(rc)
    last_range = list(out)
    return last_range


def range_verdict(vals):
    """-> None when every split tensor stayed inside the range the fp16 hi/lo arithmetic resolves, else a text."""
    bad = []
    for i in range(6):
        v = vals[i]
        if v >= RANGE_FP16_LIMIT:
            bad.append(f"{RANGE_NAMES[i]}: |x| reached the fp16 limit (bits 0x{v:04x})")
        elif 0 < v < RANGE_FP16_FLOOR:
            bad.append(f"{RANGE_NAMES[i]}: max |x| below 2^-6 (bits 0x{v:04x})")
    return "; ".join(bad) if bad else None


def _conv_range_bad(vals):
    return vals[6] >= RANGE_FP32_LIMIT


def _enter_fallback(device, why):
    """Force the per-layer fp32 MFMA form of the MLP on this device until new weights are uploaded."""
    import warnings
    d = _dev_index(device)
    warnings.warn("transhuman_amd: activations left the range of the fp16 hi/lo split MFMA kernel (" + why +
                  "); re-rendering and continuing on the per-layer fp32 MFMA path for these weights", RuntimeWarning)
    _check(load_library().th_set_mlp_mode(ctx(device), 0))
    _range_fallback[d] = True
    _range_epoch[d] = _range_epoch.get(d, 0) + 1


def force_conv_fallback(device, why):
    """Switch the stem convolutions of this device to the stock modules (what the range guard does when slot conv_in trips)
    on evidence from outside the context's own range table -- e.g. non-finite stem latents received from another rank
    (dist.StemExchange).  Bumps the epoch: frames built before are rebuilt."""
    import warnings
    d = _dev_index(device)
    if not _conv_fallback.get(d):
        warnings.warn("transhuman_amd: " + why + "; using the stock convolutions until new weights are uploaded", RuntimeWarning)
        _conv_fallback[d] = True
        _range_epoch[d] = _range_epoch.get(d, 0) + 1


def range_epoch(device=None):
    return _range_epoch.get(_dev_index(device), 0)


def _guard(device, slot):
    """True when the snapshot is clean.  Otherwise the context has been switched to the fp32 path (and / or the
    stem convolutions to the stock modules, TransHE to the fp32 GEMMs): the caller re-runs its work.  Every switch
    bumps the device's epoch: the conv / TransHE slots are sticky and written by the stream that computes the NEXT
    frames' constants, so an overflow of frame j+1 usually shows up in the snapshot of frame j -- frames whose
    constants were built before the switch are recognised by their older epoch and rebuilt (render_sequence)."""
    if slot is None or slot < 0:
        return True
    vals = range_read(slot, device)
    if vals is None:             # overwritten snapshot: the frame is unchecked -> render it again (its own snapshot is read at once)
        return False
    d = _dev_index(device)
    ok = True
    if vals[7] >= RANGE_FP16_LIMIT and not _vit_fallback.get(d):
        import warnings
        warnings.warn("transhuman_amd: an operand of TransHE's dense layers left the fp16 range; using the fp32 MFMA "
                      "GEMMs until new weights are uploaded", RuntimeWarning)
        _check(load_library().th_set_vit_mode(ctx(device), 0))
        _vit_graph_epoch[0] += 1
        _vit_fallback[d] = True
        _range_epoch[d] = _range_epoch.get(d, 0) + 1
        ok = False
    if _conv_range_bad(vals) and not _conv_fallback.get(d):
        import warnings
        warnings.warn("transhuman_amd: the ResNet-stem convolution input left the fp16 range; using the stock "
                      "convolutions until new weights are uploaded", RuntimeWarning)
        _conv_fallback[d] = True
        _range_epoch[d] = _range_epoch.get(d, 0) + 1
        ok = False
    why = range_verdict(vals)
    if why is not None and not _range_fallback.get(d):
        _enter_fallback(device, why)
        ok = False
    return ok


# ---------------------------------------------------------------------------
# building blocks (used by tests and by the Network/Renderer classes)
# ---------------------------------------------------------------------------
def linear(x, weight, bias=None, act=0):
    """C = act(x @ W^T + b) on the fp32 MFMA pipe.  x [M,K] (row stride % 4 == 0)."""
    lib = load_library()
    x = _f32(x)
    assert x.dim() == 2
    if x.shape[1] % 4:
        x = torch.nn.functional.pad(x, (0, 4 - x.shape[1] % 4))
    lin, keep = _linear(weight, bias)
    out = torch.empty((x.shape[0], lin.out_f), dtype=torch.float32, device=x.device)
    ws = _ws(lib.th_linear_workspace_bytes(lin.out_f, lin.in_f), x.device)
    _check(lib.th_linear_forward(ctx(x.device), _p(x), x.stride(0), x.shape[0], C.byref(lin), act, _p(out),
                                 out.stride(0), _p(ws), ws.numel(), _stream()))
    return out


class Points:
    """Host-side holder of a th_points (keeps the tensors alive)."""

    def __init__(self, ray_o=None, ray_d=None, near=None, far=None, n_samples=1, pts=None, z_vals=None, sigma_noise=None):
        """``z_vals`` [R,S]: explicit sample depths (the reference's stratified jitter, if_clight_renderer.py:276-283) instead of
        near (1 - t) + far t; ``sigma_noise`` [R,S]: added to sigma in front of raw2alpha's relu (nerf_net_utils.py:39-44)."""
        if pts is not None:
            self.pts = _f32(pts).reshape(-1, 3)
            self.R, self.S = self.pts.shape[0], 1
            self.keep = (self.pts,)
            self.c = ThPoints(_p(self.pts), None, None, None, None, None, None, self.R, 1)
            return
        self.ray_o, self.ray_d = _f32(ray_o).reshape(-1, 3), _f32(ray_d).reshape(-1, 3)
        self.near, self.far = _f32(near).reshape(-1), _f32(far).reshape(-1)
        dev = self.ray_o.device
        # t_vals exactly as torch.linspace builds them (if_clight_renderer.py:273-274); cached per (S, device)
        self.t, self.omt = _t_vals(n_samples, dev)
        self.R, self.S = self.ray_o.shape[0], n_samples
        self.z_vals = None if z_vals is None else _f32(z_vals).reshape(self.R, self.S)
        self.sigma_noise = None if sigma_noise is None else _f32(sigma_noise).reshape(self.R, self.S)
        self.c = ThPoints(None, _p(self.ray_o), _p(self.ray_d), _p(self.near), _p(self.far), _p(self.t), _p(self.omt),
                          self.R, self.S, _p(self.z_vals), _p(self.sigma_noise))


_t_cache = {}


def _t_vals(n_samples, dev):
    key = (int(n_samples), str(dev))
    if key not in _t_cache:
        t = torch.linspace(0.0, 1.0, steps=n_samples)
        _t_cache[key] = (t.to(dev), (1.0 - t).to(dev))
    return _t_cache[key]


def hull_mask(points, verts_world, thresh=0.1):
    lib = load_library()
    v = _f32(verts_world).reshape(-1, 3)
    P = points.R * points.S
    mask = torch.empty(P, dtype=torch.uint8, device=v.device)
    hit = torch.empty(points.R, dtype=torch.int32, device=v.device)
    ws = _ws(lib.th_hull_workspace_bytes(v.shape[0]), v.device)
    _check(lib.th_hull_mask(ctx(v.device), C.byref(points.c), _p(v), v.shape[0], thresh, _p(mask), _p(hit), _p(ws),
                            ws.numel(), _stream()))
    return mask.view(points.R, points.S).bool(), hit.bool()


def pack_cams(R, T, K):
    """[V,3,3],[V,3,1],[V,3,3] -> [V,21] fp32 (R | T | K row-major)."""
    V = R.shape[0]
    return torch.cat([_f32(R).reshape(V, 9), _f32(T).reshape(V, 3), _f32(K).reshape(V, 9)], dim=1).contiguous()


def feat_scale(scale_np, image_shape, device):
    """sample_from_feature_map, if_clight_renderer.py:193-195 (float64 divide, cast to fp32)."""
    import numpy as np
    s = np.asarray(scale_np, dtype=np.float64) / np.asarray(image_shape, dtype=np.float64)
    key = (float(s[0]), float(s[1]), str(device))
    if key not in _scale_cache:
        _scale_cache[key] = torch.tensor(s).to(dtype=torch.float32, device=device)
    return _scale_cache[key]


_scale_cache = {}


def csr_to_device(offsets, members, device):
    return (torch.as_tensor(offsets, dtype=torch.int32, device=device).contiguous(),
            torch.as_tensor(members, dtype=torch.int32, device=device).contiguous())


def paint_group(holder_map, verts_world, cams, scale_xy, vizmap, off, mem, return_painted=False):
    lib = load_library()
    m = _f32(holder_map)
    V, Cc, H, W = m.shape
    v = _f32(verts_world).reshape(-1, 3)
    nc = off.numel() - 1
    viz = vizmap.to(torch.uint8).contiguous() if vizmap is not None else None
    painted = torch.empty((V, v.shape[0], Cc), dtype=torch.float32, device=m.device)
    tokens = torch.empty((V, nc, Cc), dtype=torch.float32, device=m.device)
    _check(lib.th_paint_group(ctx(m.device), _p(m), V, Cc, H, W, _p(v), v.shape[0], _p(cams), _p(scale_xy), _p(viz),
                              _p(off), _p(mem), nc, _p(painted), _p(tokens), _stream()))
    return (tokens, painted) if return_painted else tokens


class SplitMap:
    """The compact pixel map in the TH_MAP_SPLIT layout: one buffer holding [V,H,W,256] latents (1 KiB rows: a corner
    texel is ONE aligned wave load; the interleaved 260-channel rows straddle nine 128-byte lines instead of eight and
    cost a second load instruction for their 65th float4) followed by [V,H,W,4] (r, g, b, 0)."""

    def __init__(self, buf, V, H, W, source=None):
        self.buf, self.V, self.H, self.W = buf, V, H, W
        self.shape = (V, H, W, 256)
        self.device = buf.device
        # cropped map (upsample_concat_split(box=...)): (ThMapSource, the tensors it points to); texels outside the
        # per-view box are NOT written until a frame-level call needs them (th_frame.map_source)
        self.source = source
        # th_map_fold's output for this map ([2, V, H, W, 256] fp32; hip.map_fold), or None: frames built on the map then take
        # the texel hand-over (th_frame.map_fold)
        self.fold = None

    @property
    def box(self):
        """device int32 [V,4] (x0, y0, x1, y1 inclusive) of a cropped map, else None"""
        return self.source[1][0] if self.source is not None else None

    @property
    def demand(self):
        """the render_predemand buffer a demand-driven map was written for, else None"""
        return self.source[1][5] if self.source is not None and len(self.source[1]) > 5 else None

    def data_ptr(self):
        return self.buf.data_ptr()

    @property
    def latents(self):
        return self.buf[: self.V * self.H * self.W * 256].view(self.V, self.H, self.W, 256)

    @property
    def rgb0(self):
        return self.buf[self.V * self.H * self.W * 256:].view(self.V, self.H, self.W, 4)

    def interleaved(self):
        """the same map as one [V,H,W,260] tensor (tests / A-B)"""
        return torch.cat([self.latents, self.rgb0], dim=-1).contiguous()


def map_fold(net, split_map):
    """th_map_fold: alpha_res_0 / rgb_res_0 (under the folded view_fc) / rgb_res_1 of ``net`` applied to the texels of
    ``split_map`` (inside its crop box), on the current stream; the result is kept as ``split_map.fold``."""
    assert isinstance(split_map, SplitMap)
    _sync_weights(net, "mlp")
    V, H, W = split_map.V, split_map.H, split_map.W
    fold = torch.empty((2, V, H, W, 256), dtype=torch.float32, device=split_map.device)
    if split_map.demand is not None:          # demand-driven map: the texels the frame's samples read, as a compacted list
        _check(load_library().th_map_fold_demand(ctx(split_map.device), _p(split_map), V, H, W, _p(split_map.demand), _p(fold), _stream()))
        split_map.fold = fold
        return fold
    if split_map.box is not None:
        _box_checked(split_map.box, H)
    _check(load_library().th_map_fold(ctx(split_map.device), _p(split_map), V, H, W, _p(split_map.box), _p(fold), _stream()))
    split_map.fold = fold
    return fold


def map_box(verts_a, verts_b, cams, scale_xy, H, W, reach):
    """th_map_box: per view the texel box (device int32 [V,4]: x0, y0, x1, y1 inclusive) that holds every texel a
    bilinear gather at a point within ``reach`` (per axis) of a vertex of verts_a / verts_b can read."""
    lib = load_library()
    a = _f32(verts_a).reshape(-1, 3)
    b = _f32(verts_b).reshape(-1, 3) if verts_b is not None else None
    V = cams.shape[0]
    # [V,4] boxes followed by [V,H,2] row spans (x0, x1 of every image row: the outline of the body inside the box); the
    # returned tensor is the [V,4] head of that buffer, `map_spans(box, H)` views the rest
    buf = torch.empty(V * 4 + V * int(H) * 2 + V, dtype=torch.int32, device=a.device)      # (+ V flag words of the kernels)
    box = buf[: V * 4].view(V, 4)
    box._th_map_box = (buf, int(H))              # (a clone / an exchanged [V,4] copy has no spans behind it: _box_checked)
    _check(lib.th_map_box(ctx(a.device), _p(a), a.shape[0], _p(b), b.shape[0] if b is not None else 0, _p(cams), V,
                          _p(scale_xy), int(H), int(W), float(reach), _p(box), _stream()))
    return box


def _box_checked(box, H):
    """``box`` must be the tensor hip.map_box returned (the row spans and flag words th_map_box wrote sit BEHIND it in the
    same buffer and the C side reads them through the box pointer): a copy of the [V,4] values is refused"""
    tag = getattr(box, "_th_map_box", None)
    if tag is None or tag[0].data_ptr() != box.data_ptr() or tag[1] != int(H):
        raise HipError("box must be the tensor hip.map_box returned for this image height (its row spans follow it in memory)")
    return box


def map_spans(box, H):
    """the [V,H,2] row spans (x0, x1 inclusive; x1 < x0: empty row) th_map_box wrote behind the boxes of ``box``"""
    _box_checked(box, H)
    V = box.shape[0]
    return torch.as_strided(box, (V, int(H), 2), (int(H) * 2, 2, 1), storage_offset=box.storage_offset() + V * 4)


def upsample_concat_split(images, lat0, lat1, lat2, box=None, reach=0.0, demand=None):
    """th_upsample_concat_split(_box): the compact map (colour lift folded into the consumers) in the split layout.
    ``box`` (map_box, computed with ``reach``): only the texels of each view's box are written -- the returned SplitMap
    carries what it was made from (``source``) and hip.Frame hands that to the C side, which completes the map on its
    own if a call gathers outside the hull's reach (un-masked small-frame branch, no hull test)."""
    lib = load_library()
    img, l0, l1, l2 = _f32(images), _f32(lat0), _f32(lat1), _f32(lat2)
    V, _, H, W = img.shape
    assert l0.shape[1] == 64 and l1.shape[1] == 64 and l2.shape[1] == 128
    dims = (C.c_int32 * 6)(l0.shape[2], l0.shape[3], l1.shape[2], l1.shape[3], l2.shape[2], l2.shape[3])
    buf = torch.empty(V * H * W * 260, dtype=torch.float32, device=img.device)
    if demand is not None:
        # demand-driven map (render_predemand): only the texels the frame's valid samples / painted vertices read are written
        _check(lib.th_upsample_concat_split_demand(ctx(img.device), _p(img), _p(l0), _p(l1), _p(l2), dims, V, H, W, _p(buf), None,
                                                   _p(demand), _stream()))
        src = ThMapSource(None, float(reach), _p(img), _p(l0), _p(l1), _p(l2), dims, _p(demand))
        return SplitMap(buf, V, H, W, source=(src, (None, img, l0, l1, l2, demand)))
    if box is None:
        _check(lib.th_upsample_concat_split(ctx(img.device), _p(img), _p(l0), _p(l1), _p(l2), dims, V, H, W, _p(buf), _stream()))
        return SplitMap(buf, V, H, W)
    assert box.dtype == torch.int32 and tuple(box.shape) == (V, 4) and box.is_contiguous()
    # (th_map_box's buffer: the row spans follow the boxes)
    _box_checked(box, H)
    _check(lib.th_upsample_concat_split_box(ctx(img.device), _p(img), _p(l0), _p(l1), _p(l2), dims, V, H, W, _p(buf),
                                            _p(box), _stream()))
    src = ThMapSource(_p(box), float(reach), _p(img), _p(l0), _p(l1), _p(l2), dims, None)
    return SplitMap(buf, V, H, W, source=(src, (box, img, l0, l1, l2)))


def upsample_concat_nhwc(images, lat0, lat1, lat2, color_w=None, color_b=None):
    """Encoder tail (encoder.py:133-146) -> channels-last pixel_feat_map [V,H,W,384], or with
    color_w=None the compact map [V,H,W,260] (256 latent | r g b | 0; the lift is folded into the consumers)."""
    lib = load_library()
    img, l0, l1, l2 = _f32(images), _f32(lat0), _f32(lat1), _f32(lat2)
    V, _, H, W = img.shape
    assert l0.shape[1] == 64 and l1.shape[1] == 64 and l2.shape[1] == 128
    dims = (C.c_int32 * 6)(l0.shape[2], l0.shape[3], l1.shape[2], l1.shape[3], l2.shape[2], l2.shape[3])
    cw = _f32(color_w).reshape(128, 3) if color_w is not None else None
    cb = _f32(color_b) if color_w is not None else None
    out = torch.empty((V, H, W, 384 if color_w is not None else 260), dtype=torch.float32, device=img.device)
    _check(lib.th_upsample_concat_nhwc(ctx(img.device), _p(img), _p(l0), _p(l1), _p(l2), dims, V, H, W, _p(cw), _p(cb),
                                       _p(out), _stream()))
    return out


def paint_group_nhwc(map_nhwc, verts_world, cams, scale_xy, vizmap, red_w, red_b, off, mem, color_w=None,
                     color_b=None):
    """Sample the channels-last map at the projected vertices, apply reduction_layer there, mask, pool.
    A compact (260-channel) map needs the colour lift (color_w, color_b) to fold into the reduction layer."""
    lib = load_library()
    V, H, W, Cc = map_nhwc.shape                      # (a SplitMap reports 256 channels: TH_MAP_SPLIT)
    v = _f32(verts_world).reshape(-1, 3)
    nc = off.numel() - 1
    viz = vizmap.to(torch.uint8).contiguous() if vizmap is not None else None
    lin, keep = _linear(red_w, red_b)
    lift, keep2 = _linear(color_w, color_b) if color_w is not None else (None, None)
    tokens = torch.empty((V, nc, lin.out_f), dtype=torch.float32, device=v.device)
    ws = _ws(lib.th_paint_group_nhwc_workspace_bytes(V, v.shape[0], Cc, lin.out_f), v.device)
    _check(lib.th_paint_group_nhwc(ctx(v.device), _p(map_nhwc), V, H, W, Cc, _p(v), v.shape[0], _p(cams), _p(scale_xy),
                                   _p(viz), C.byref(lin), C.byref(lift) if lift is not None else None, _p(off), _p(mem),
                                   nc, _p(tokens), _p(ws), ws.numel(), _stream()))
    return tokens


_conv_cache = {}


def conv2d_supported(conv):
    """True for the nn.Conv2d shapes th_conv2d is built for (the bias-free convolutions of the ResNet18 stem)."""
    ks, st, pd = conv.kernel_size, conv.stride, conv.padding
    return (not conv_fallback(conv.weight.device) and conv.bias is None and ks[0] == ks[1] and st[0] == st[1] and pd[0] == pd[1] == ks[0] // 2 and
            conv.groups == 1 and conv.dilation == (1, 1) and
            bool(load_library().th_conv2d_supported(conv.in_channels, conv.out_channels, ks[0], st[0])))


def conv2d(x, conv, stats=False):
    """th_conv2d: y = conv(x) for a supported bias-free nn.Conv2d (NCHW fp32), fp16-split MFMA implicit GEMM.
    The packed weight image is cached per module and rebuilt when the weight tensor changes.
    ``stats=True`` (th_conv2d_stats): also returns the per-channel BatchNorm partial sums the epilogue leaves, as
    (y, (buffer, n_partials)) for bn_act(..., conv_stats=...)."""
    lib = load_library()
    assert x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous()
    w = conv.weight
    key = id(conv)
    ent = _conv_cache.get(key)
    if ent is None or ent[0] is not w or ent[1] != w._version or ent[2].device != x.device:
        co, ci, ks, _ = w.shape
        buf = torch.empty(lib.th_conv_pack_bytes(co, ci, ks), dtype=torch.uint8, device=x.device)
        inv = C.c_float()
        _check(lib.th_conv_pack(ctx(x.device), _p(_f32(w)), co, ci, ks, _p(buf), buf.numel(), C.byref(inv), _stream()))
        ent = (w, w._version, buf, float(inv.value))
        _conv_cache[key] = ent
    N, ci, H, W = x.shape
    co, ks, st = conv.out_channels, conv.kernel_size[0], conv.stride[0]
    pd = ks // 2
    Ho, Wo = (H + 2 * pd - ks) // st + 1, (W + 2 * pd - ks) // st + 1
    y = torch.empty((N, co, Ho, Wo), dtype=torch.float32, device=x.device)
    if stats:
        npart = int(lib.th_conv2d_stats_partials(N, ci, H, W, co, ks, st))
        sbuf = torch.empty((co, npart, 2), dtype=torch.float32, device=x.device)
        _check(lib.th_conv2d_stats(ctx(x.device), _p(x), N, ci, H, W, _p(ent[2]), ent[3], co, ks, st, _p(y), _p(sbuf),
                                   sbuf.numel() * 4, _stream()))
        return y, (sbuf, npart)
    _check(lib.th_conv2d(ctx(x.device), _p(x), N, ci, H, W, _p(ent[2]), ent[3], co, ks, st, _p(y), _stream()))
    return y


def maxpool3x3s2(x):
    """th_maxpool3x3s2: nn.MaxPool2d(3, 2, 1) on NCHW fp32."""
    assert x.dim() == 4 and x.dtype == torch.float32 and x.is_contiguous()
    N, Cc, H, W = x.shape
    y = torch.empty((N, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    _check(load_library().th_maxpool3x3s2(ctx(x.device), _p(x), N * Cc, H, W, _p(y), _stream()))
    return y


def bn_act(x, bn, residual=None, relu=True, conv_stats=None):
    """th_bn_act: train-mode nn.BatchNorm2d `bn` on x [N,C,H,W] (+ residual) (+ ReLU) in two launches; updates
    bn.running_mean / running_var like the module (num_batches_tracked is left to the caller).  A module in eval() mode
    normalises with its running statistics in one launch (th_bn_act_eval).  ``conv_stats`` = the (buffer, n_partials) of
    conv2d(..., stats=True) that produced x: the statistics pass is skipped (th_bn_act_stats, one launch)."""
    lib = load_library()
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
    N, Cc, H, W = x.shape
    r = None if residual is None else residual.contiguous()
    if not bn.training:             # eval(): running statistics, one launch (th_bn_act_eval)
        assert bn.track_running_stats and bn.running_mean is not None, "eval-mode BatchNorm without running statistics"
        y = torch.empty_like(x)
        _check(lib.th_bn_act_eval(ctx(x.device), _p(x), None if r is None else _p(r), N, Cc, H * W,
                                  None if bn.weight is None else _p(bn.weight), None if bn.bias is None else _p(bn.bias),
                                  float(bn.eps), _p(bn.running_mean), _p(bn.running_var), int(relu), _p(y), _stream()))
        return y
    y = torch.empty_like(x)
    track = bn.track_running_stats and bn.running_mean is not None
    assert bn.momentum is not None, "cumulative-average BatchNorm (momentum=None) is not handled here"
    mom = float(bn.momentum)
    r = None if residual is None else residual.contiguous()
    if conv_stats is not None:
        sbuf, npart = conv_stats
        assert sbuf.shape == (Cc, npart, 2) and sbuf.dtype == torch.float32 and sbuf.device == x.device
        _check(lib.th_bn_act_stats(ctx(x.device), _p(x), None if r is None else _p(r), N, Cc, H * W, _p(sbuf), npart,
                                   None if bn.weight is None else _p(bn.weight), None if bn.bias is None else _p(bn.bias),
                                   float(bn.eps), mom, _p(bn.running_mean) if track else None,
                                   _p(bn.running_var) if track else None, int(relu), _p(y), _stream()))
        return y
    ws = _ws(lib.th_bn_workspace_bytes(N, Cc, H * W), x.device)
    _check(lib.th_bn_act(ctx(x.device), _p(x), None if r is None else _p(r), N, Cc, H * W,
                         None if bn.weight is None else _p(bn.weight), None if bn.bias is None else _p(bn.bias),
                         float(bn.eps), mom, _p(bn.running_mean) if track else None,
                         _p(bn.running_var) if track else None, int(relu), _p(y), _p(ws), ws.numel(), _stream()))
    return y


def segment_mean(src, off, mem):
    lib = load_library()
    s = _f32(src)
    width = int(s[0].numel())
    s2 = s.reshape(s.shape[0], width)
    nc = off.numel() - 1
    out = torch.empty((nc, width), dtype=torch.float32, device=s.device)
    _check(lib.th_segment_mean_f32(ctx(s.device), _p(s2), width, _p(off), _p(mem), nc, _p(out), _stream()))
    return out.reshape(nc, *s.shape[1:])


def segment_mean_rot(blend, off, mem):
    """blend [n_verts,4,4] (float64 kept as is) -> fp32 [N_c,3,3]."""
    lib = load_library()
    b = blend.detach().to(torch.float64).contiguous().reshape(-1, 16)
    nc = off.numel() - 1
    out = torch.empty((nc, 9), dtype=torch.float32, device=b.device)
    _check(lib.th_segment_mean_rot_f64(ctx(b.device), _p(b), _p(off), _p(mem), nc, _p(out), _stream()))
    return out


_graphs_off = [False]


def graph_capture(fn):
    """Record the launches ``fn()`` queues on the current stream into a hipGraph (torch.cuda.CUDAGraph, private memory pool)
    -> (graph, fn's result).  Capture mode "thread_local": API calls other threads make meanwhile (the process group's
    watchdog polling its events in a multi-rank job) neither fail nor invalidate the capture.  A capture that fails raises
    GraphCaptureFailed after switching the replayed-graph forms off for the process (callers run their eager form)."""
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = fn()
    except Exception as e:           # (nothing of fn ran; the capture is over either way)
        _graphs_off[0] = True
        import warnings
        warnings.warn(f"transhuman_amd: hipGraph capture failed ({type(e).__name__}: {e}); the stem and TransHE run as "
                      "separate launches from here on", RuntimeWarning)
        raise GraphCaptureFailed(str(e)) from e
    return g, out


class GraphCaptureFailed(RuntimeError):
    pass


def graphs_enabled():
    return not _graphs_off[0]


_vit_graphs = {}
def graph_epoch():
    """counter bumped whenever captured graphs are dropped for new weights / arithmetic (frame pipelines key their priming on it)"""
    return _vit_graph_epoch[0]


_vit_graph_epoch = [0]      # bumped whenever the context's TransHE weights or arithmetic change: captured graphs are dropped
VIT_GRAPH_RING = 4


def _vit_forward_graphed(vit, x, pe):
    """TransHE's 63 launches replayed as one hipGraph (the frame paths: Renderer.render_fast / render_sequence).  Per (module,
    shape, device): the first call of a version runs eagerly and captures VIT_GRAPH_RING instances (static input copy + output
    in the graph's private pool), later calls copy the input into the next instance and replay it.  The tokens returned stay valid
    until VIT_GRAPH_RING - 1 further graphed calls of the shape have been made (the frame pipeline holds three frames).
    The captured forward holds kernel nodes only (k_vit.hip: zero16_kernel, profiles/r05_l_vit_graph_memset_node.txt).
    None = not applicable (TH_VIT_GRAPH=0, autograd input)."""
    if os.environ.get("TH_VIT_GRAPH", "1") == "0" or _graphs_off[0] or (torch.is_grad_enabled() and x.requires_grad):
        return None
    key = (id(vit), tuple(x.shape), str(x.device))
    ver = (_vit_graph_epoch[0], pe.data_ptr(), _ctx_owner.get((_dev_index(x.device), "vit"), (None, None))[1])
    st = _vit_graphs.get(key)
    if st is not None and st.get("eager_only"):
        st["same"] = st["same"] + 1 if st["ver"] == ver else 0          # (a version that has settled gets its graphs back)
        st["ver"] = ver
        if st["same"] < 2:
            return None
        st = None
    if st is None or st["ver"] != ver:
        # first call of a version: eagerly (the kernels' first launches must not happen under capture), then every instance of
        # the ring is captured at once -- the cost of capturing (a device synchronisation each) lands in this one call, not in
        # the next VIT_GRAPH_RING frames of a stream (a rank of N computes TransHE every N-th frame only)
        # (a version that changes with every call -- weights updated between frames, a positional table converted anew each time --
        # would capture on every call: after three captures that were never replayed the module stays on separate launches)
        thrash = 0 if st is None or st["replays"] > 0 else st["thrash"] + 1
        if thrash >= 3:
            _vit_graphs[key] = {"eager_only": True, "mod": vit, "ver": ver, "same": 0}
            return None
        if len(_vit_graphs) > 8:
            _vit_graphs.clear()
        st = _vit_graphs[key] = {"ver": ver, "inst": [], "next": 0, "mod": vit, "pe": pe,     # (the graphs hold pe's address)
                                 "replays": 0, "thrash": thrash}
        out = vit_forward(vit, x, pe, graph=False, _checked=True)
        try:
            for _ in range(VIT_GRAPH_RING):
                xs = torch.empty_like(x)
                g, o = graph_capture(lambda: vit_forward(vit, xs, pe, graph=False, _checked=True))
                st["inst"].append((g, xs, o))
        except GraphCaptureFailed:
            _vit_graphs.pop(key, None)
        return out
    k = st["next"]
    st["next"] = (k + 1) % VIT_GRAPH_RING
    st["replays"] += 1
    g, xs, out = st["inst"][k]
    xs.copy_(x)
    g.replay()
    return out


def vit_forward(vit, x, pe, graph=False, _checked=False):
    lib = load_library()
    if not _checked:
        _sync_weights(vit, "vit")
    x, pe = _f32(x), _f32(pe)
    if graph:
        out = _vit_forward_graphed(vit, x, pe)
        if out is not None:
            return out
    V, N, D = x.shape
    out = torch.empty_like(x)
    ws = _ws(lib.th_vit_workspace_bytes(V, N, D, vit.num_heads), x.device)
    _check(lib.th_vit_forward(ctx(x.device), _p(x), _p(pe), V, N, _p(out), _p(ws), ws.numel(), _stream()))
    return out


def dparf_encode(pts_smpl, centres, rot, tokens, sel=None):
    lib = load_library()
    p, c, r, t = _f32(pts_smpl).reshape(-1, 3), _f32(centres).reshape(-1, 3), _f32(rot).reshape(-1, 9), _f32(tokens)
    P = p.shape[0] if sel is None else sel.numel()
    V, nc = t.shape[0], t.shape[1]
    out = torch.empty((P, V, 256), dtype=torch.float32, device=p.device)
    _check(lib.th_dparf_encode(ctx(p.device), _p(p), _p(sel), P, _p(c), _p(r), _p(t), V, nc, _p(out), _stream()))
    return out


def nchw_to_nhwc(m):
    lib = load_library()
    m = _f32(m)
    V, Cc, H, W = m.shape
    out = torch.empty((V, H, W, Cc), dtype=torch.float32, device=m.device)
    _check(lib.th_nchw_to_nhwc(ctx(m.device), _p(m), V, Cc, H, W, _p(out), _stream()))
    return out


def pixel_gather(map_nhwc, pts_world, cams, scale_xy, sel=None, row_floats=None):
    """-> [P, V, row_floats] (default: the map's channel count; wider rows are zero padded)."""
    lib = load_library()
    V, H, W, Cc = map_nhwc.shape
    p = _f32(pts_world).reshape(-1, 3)
    P = p.shape[0] if sel is None else sel.numel()
    ldo = Cc if row_floats is None else int(row_floats)
    out = torch.empty((P, V, ldo), dtype=torch.float32, device=p.device)
    _check(lib.th_pixel_gather(ctx(p.device), _p(map_nhwc), V, Cc, H, W, _p(p), _p(sel), P, _p(cams), _p(scale_xy),
                               _p(out), ldo, _stream()))
    return out


def pixel_gather_split(split_map, pts_world, cams, scale_xy, sel=None):
    """th_pixel_gather_split: the split-row form of the pixel-aligned gather (what the frame-level entry points run) ->
    (hi, lo) fp16 tensors [P, V, 272] (x = hi + lo), decoded from the [8 hi | 8 lo] groups of the rows."""
    lib = load_library()
    assert isinstance(split_map, SplitMap)
    V, H, W = split_map.V, split_map.H, split_map.W
    p = _f32(pts_world).reshape(-1, 3)
    P = p.shape[0] if sel is None else sel.numel()
    out = torch.zeros((P, V, 272 * 2), dtype=torch.float16, device=p.device)
    _check(lib.th_pixel_gather_split(ctx(p.device), _p(split_map), V, H, W, _p(p), _p(sel), P, _p(cams), _p(scale_xy),
                                     _p(out), 272, _stream()))
    g = out.view(P, V, 34, 2, 8)
    return g[:, :, :, 0].reshape(P, V, 272), g[:, :, :, 1].reshape(P, V, 272)


def pixel_texlist(split_map, pts_world, cams, scale_xy, sel=None):
    """th_pixel_texlist: K5t on its own -> dict(lists [T,4,128] int32, records [T,V,32,8] int32 (float bits / byte offsets)),
    T = ceil(P / 32) tiles of 32 consecutive samples."""
    lib = load_library()
    assert isinstance(split_map, SplitMap)
    V, H, W = split_map.V, split_map.H, split_map.W
    p = _f32(pts_world).reshape(-1, 3)
    P = p.shape[0] if sel is None else sel.numel()
    T = (P + 31) // 32
    nb = int(lib.th_pixel_texlist_bytes(V, P))
    out = torch.zeros(nb // 4, dtype=torch.int32, device=p.device)
    _check(lib.th_pixel_texlist(ctx(p.device), _p(split_map), V, H, W, _p(p), _p(sel), P, _p(cams), _p(scale_xy), _p(out), nb,
                                _stream()))
    a, b = T * 512, T * 512 + T * V * 32 * 8
    return dict(lists=out[:a].view(T, 4, 128), records=out[a:b].view(T, V, 32, 8))


def network_forward(net, pixel_feat, viewdir, pts_smpl, centres, rot, tokens, mask=None):
    """Network.forward on gathered inputs -> raw [P,4]."""
    lib = load_library()
    _sync_weights(net, "mlp")
    pf, vd, ps = _f32(pixel_feat), _f32(viewdir).reshape(-1, 27), _f32(pts_smpl).reshape(-1, 3)
    V, Cc, P = pf.shape
    assert Cc == 384 and vd.shape[0] == P and ps.shape[0] == P
    c, r, t = _f32(centres).reshape(-1, 3), _f32(rot).reshape(-1, 9), _f32(tokens)
    m = mask.reshape(-1).to(torch.uint8).contiguous() if mask is not None else None
    raw = torch.empty((P, 4), dtype=torch.float32, device=pf.device)
    if P == 0:
        return raw
    ws = _ws(lib.th_network_workspace_bytes(V, P), pf.device)
    for _ in range(3):
        _check(lib.th_network_forward(ctx(pf.device), _p(pf), _p(vd), _p(ps), _p(m), P, _p(c), _p(r), _p(t), V, t.shape[1],
                                      _p(raw), _p(ws), ws.numel(), _stream()))
        if _guard(pf.device, lib.th_range_last_slot(ctx(pf.device))):
            break                                    # (else: the context is on the fp32 path now -- run again)
    return raw


def composite(raw, z, ray_d, white_bkgd=False, return_weights=False):
    lib = load_library()
    raw, z, d = _f32(raw), _f32(z), _f32(ray_d).reshape(-1, 3)
    R, S = z.shape
    pts = ThPoints(None, None, _p(d), None, None, None, None, R, S)
    rgb = torch.empty((R, 3), dtype=torch.float32, device=raw.device)
    acc = torch.empty(R, dtype=torch.float32, device=raw.device)
    dep = torch.empty(R, dtype=torch.float32, device=raw.device)
    w = torch.empty((R, S), dtype=torch.float32, device=raw.device) if return_weights else None
    _check(lib.th_composite(ctx(raw.device), _p(raw), _p(z), C.byref(pts), int(white_bkgd), _p(rgb), _p(acc), _p(dep),
                            _p(w), _stream()))
    return (rgb, acc, dep, w) if return_weights else (rgb, acc, dep)


def gen_rays(K, R, T, bounds, H, W, device=None, compact=True):
    """Rays of one target camera + box near/far, if_nerf_data_utils.py:11-30, :65-97 (test split of
    sample_ray_h36m :271-283).  K [3,3], R [3,3], T [3,1], bounds [2,3]: float32 numpy / torch (host values).
    compact=True returns the reference's masked ray list {ray_o [R',3], ray_d [R',3], near [R'], far [R'],
    mask_at_box bool [H*W]}; compact=False the dense per-pixel arrays."""
    import numpy as np
    lib = load_library()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)

    def host(a, n):
        a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        assert a.size == n
        return a

    k, r, t, b = host(K, 9), host(R, 9), host(T, 3), host(bounds, 6)
    n = H * W
    ray_o = torch.empty((n, 3), dtype=torch.float32, device=dev)
    ray_d = torch.empty((n, 3), dtype=torch.float32, device=dev)
    near = torch.empty(n, dtype=torch.float32, device=dev)
    far = torch.empty(n, dtype=torch.float32, device=dev)
    mask = torch.empty(n, dtype=torch.uint8, device=dev)
    fp = lambda a: a.ctypes.data_as(c_float_p)
    _check(lib.th_gen_rays(ctx(dev), fp(k), fp(r), fp(t), fp(b), H, W, _p(ray_o), _p(ray_d), _p(near), _p(far), _p(mask),
                           _stream()))
    m = mask.bool()
    if not compact:
        return dict(ray_o=ray_o, ray_d=ray_d, near=near, far=far, mask_at_box=m)
    return dict(ray_o=ray_o[m], ray_d=ray_d[m], near=near[m], far=far[m], mask_at_box=m)


def bound_corners_2d(bounds, K, pose):
    """The eight corners of ``bounds`` [2,3] projected into the camera (K [3,3], pose = [R|T] [3,4]) and rounded
    to integer pixels exactly like if_nerf_data_utils.py:33-53 (get_bound_corners, base_utils.project :178-187,
    np.round(..).astype(int)); the dtype of the inputs is kept, as numpy would."""
    import numpy as np
    bounds, K, pose = np.asarray(bounds), np.asarray(K), np.asarray(pose)
    mn, mx = bounds[0], bounds[1]
    corners = np.array([[mn[0], mn[1], mn[2]], [mn[0], mn[1], mx[2]], [mn[0], mx[1], mn[2]], [mn[0], mx[1], mx[2]],
                        [mx[0], mn[1], mn[2]], [mx[0], mn[1], mx[2]], [mx[0], mx[1], mn[2]], [mx[0], mx[1], mx[2]]])
    xyz = np.dot(corners, pose[:, :3].T) + pose[:, 3:].T
    xyz = np.dot(xyz, K.T)
    xy = xyz[:, :2] / xyz[:, 2:]
    return np.round(xy).astype(int)


def bound_2d_mask(bounds, K, pose, H, W, device=None):
    """th_bound_mask: get_bound_2d_mask (if_nerf_data_utils.py:49-62) -> uint8 [H, W] device tensor."""
    import numpy as np
    lib = load_library()
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    c2 = np.ascontiguousarray(bound_corners_2d(bounds, K, pose), dtype=np.int32)
    mask = torch.empty((H, W), dtype=torch.uint8, device=dev)
    _check(lib.th_bound_mask(ctx(dev), c2.ctypes.data_as(C.POINTER(C.c_int32)), H, W, _p(mask), _stream()))
    return mask


def marching_cubes(cube, iso, scale=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), x_range=None):
    """th_marching_cubes_*: mcubes.marching_cubes(cube, iso) on the device (if_mesh_renderer.py:103) with the
    index -> world transform of :106-108 applied (vertex = index * scale + origin, float64).
    cube: [X,Y,Z] fp32 device tensor.  -> (vertices float64 [nv,3], triangles int32 [nt,3]) device tensors.
    x_range=(x0, x1): only the grid slab x0 <= x < x1 is emitted (multi-GPU z-slab style sharding); returns
    (vertices, triangles, (v0, v1), (t0, t1)) where the slab filled rows [v0, v1) / [t0, t1) of the full arrays."""
    lib = load_library()
    c = _f32(cube)
    assert c.dim() == 3
    X, Y, Z = c.shape
    dev = c.device
    ws = _ws(lib.th_marching_cubes_workspace_bytes(X, Y, Z), dev)
    counts = (C.c_int64 * 2)()
    _check(lib.th_marching_cubes_count(ctx(dev), _p(c), X, Y, Z, float(iso), _p(ws), ws.numel(), counts, _stream()))
    nv, nt = int(counts[0]), int(counts[1])
    verts = torch.zeros((nv, 3), dtype=torch.float64, device=dev)
    tris = torch.zeros((nt, 3), dtype=torch.int32, device=dev)
    sc = (C.c_double * 3)(*[float(v) for v in scale])
    og = (C.c_double * 3)(*[float(v) for v in origin])
    x0, x1 = (0, X) if x_range is None else (int(x_range[0]), int(x_range[1]))
    if nv > 0:
        _check(lib.th_marching_cubes_emit(ctx(dev), _p(c), X, Y, Z, float(iso), _p(ws), x0, x1, sc, og,
                                          _p(verts), _p(tris) if nt > 0 else _p(verts), _stream()))
    if x_range is None:
        return verts, tris
    a, b = (C.c_int64 * 2)(), (C.c_int64 * 2)()
    _check(lib.th_marching_cubes_range(ctx(dev), _p(ws), X, Y, Z, max(x0, 0), a, _stream()))
    _check(lib.th_marching_cubes_range(ctx(dev), _p(ws), X, Y, Z, min(x1, X), b, _stream()))
    return verts, tris, (int(a[0]), int(b[0])), (int(a[1]), int(b[1]))


class SmplModel:
    """Device copy of the SMPL model arrays (the fields lib/utils/SMPL.py:83-89 reads), float64."""

    def __init__(self, arrays, device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        f64 = lambda k: torch.as_tensor(arrays[k], dtype=torch.float64).contiguous().to(dev)
        self.v_template, self.shapedirs, self.posedirs = f64("v_template"), f64("shapedirs"), f64("posedirs")
        self.J_regressor, self.weights = f64("J_regressor"), f64("weights")
        self.parent = torch.as_tensor(arrays["parent"], dtype=torch.int32).contiguous().to(dev)
        self.nv = self.v_template.shape[0]
        assert self.shapedirs.shape == (self.nv, 3, 10) and self.posedirs.shape == (self.nv, 3, 207)
        assert self.J_regressor.shape == (24, self.nv) and self.weights.shape == (self.nv, 24)
        self.c = ThSmplModel(_p(self.v_template), _p(self.shapedirs), _p(self.posedirs), _p(self.J_regressor),
                             _p(self.weights), _p(self.parent), self.nv)

    def __call__(self, pose, beta):
        """SMPL.__call__ (lib/utils/SMPL.py:107-186): pose = 72 axis-angle values or [24,3,3] rotation matrices;
        beta [10].  -> (v [nv,3], joints [24,3], T [nv,4,4]) float64 device tensors."""
        lib = load_library()
        dev = self.v_template.device
        p = torch.as_tensor(pose, dtype=torch.float32).contiguous().to(dev)
        b = torch.as_tensor(beta, dtype=torch.float64).reshape(10).contiguous().to(dev)
        is_rot = tuple(p.shape) == (24, 3, 3)
        assert is_rot or p.numel() == 72, "Unsupported Pose Inputs"
        v = torch.empty((self.nv, 3), dtype=torch.float64, device=dev)
        j = torch.empty((24, 3), dtype=torch.float64, device=dev)
        T = torch.empty((self.nv, 4, 4), dtype=torch.float64, device=dev)
        ws = _ws(lib.th_smpl_workspace_bytes(self.nv), dev)
        _check(lib.th_smpl_lbs(ctx(dev), C.byref(self.c), None if is_rot else _p(p), _p(p) if is_rot else None, _p(b),
                               _p(v), _p(j), _p(T), _p(ws), ws.numel(), _stream()))
        return v, j, T


def view_embed(ray_d, view_res=4):
    lib = load_library()
    d = _f32(ray_d).reshape(-1, 3)
    out = torch.empty((d.shape[0], 3 + 6 * view_res), dtype=torch.float32, device=d.device)
    _check(lib.th_view_embed(ctx(d.device), _p(d), d.shape[0], view_res, _p(out), _stream()))
    return out


class Frame:
    """Per-frame constants of the per-sample stage (th_frame) + owners."""

    def __init__(self, verts_world, Rh, Th, cams, scale_xy, pixel_map_nhwc, tokens, centres, rot,
                 hull_thresh=0.1, small_frame_rays=2400):
        self.verts = _f32(verts_world).reshape(-1, 3)
        self.Rh, self.Th = _f32(Rh).reshape(9), _f32(Th).reshape(3)
        self.cams, self.scale = cams, scale_xy
        self.map = pixel_map_nhwc
        self.centres, self.rot = _f32(centres).reshape(-1, 3), _f32(rot).reshape(-1, 9)
        self.tokens = _f32(tokens) if tokens is not None else None       # None: set_tokens() before the frame is rendered
        V, H, W, Cc = pixel_map_nhwc.shape
        assert Cc in (384, 260) or isinstance(pixel_map_nhwc, SplitMap), \
            "pixel map must be the full (384) or the compact (260 interleaved / SplitMap) channels-last map"
        src = getattr(pixel_map_nhwc, "source", None)
        self.c = ThFrame(_p(self.verts), self.verts.shape[0], _p(self.Rh), _p(self.Th), _p(self.cams), _p(self.scale),
                         _p(self.map), V, H, W, Cc, _p(self.tokens), _p(self.centres), _p(self.rot),
                         self.centres.shape[0], hull_thresh, small_frame_rays,
                         C.cast(C.pointer(src[0]), C.c_void_p).value if src is not None else None,
                         _p(getattr(pixel_map_nhwc, "fold", None)))

    def set_tokens(self, tokens):
        """TransHE output [V, N_c, 192] of a frame that was built without it (render_pregather runs beside TransHE)"""
        self.tokens = _f32(tokens)
        assert self.tokens.shape[1] == self.centres.shape[0]
        self.c.tokens = self.tokens.data_ptr()


_ws_cache = {}


def _cached_ws(nbytes, device, slot=0):
    """Render workspace, grown on demand.  ``slot`` selects one of several independent workspaces (the frame
    pipeline of Renderer.render_sequence runs the hull stage of frame i+1 while frame i is still shading)."""
    key = (str(device), slot)
    cur = _ws_cache.get(key)
    if cur is None or cur.numel() < nbytes:
        _ws_cache[key] = None
        cur = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _ws_cache[key] = cur
    return cur


_pool_cache = {}


def _shade_pool(frame_c, n_valid, with_pregather, device, slot=None):
    """The context's shading pool (include/transhuman_hip.h: th_shade_pool_bytes), ONE per device, grown on demand to what
    ``n_valid`` valid samples need in the context's current mode.  Growth allocates on the current stream; the old pool
    goes back to torch's allocator, which hands it out again only behind the work already queued on it."""
    need = int(load_library().th_shade_pool_bytes(ctx(device), C.byref(frame_c), int(n_valid), int(with_pregather)))
    # (slot: a frame pipeline that forms the NEXT frame's records while this frame shades gives every frame in flight a pool of
    # its own -- Renderer.render_sequence; None: the context's one shared pool)
    key = str(device) if slot is None else (str(device), int(slot))
    cur = _pool_cache.get(key)
    if cur is None or cur.numel() < need:
        _pool_cache[key] = None
        # (headroom: frames of a sequence differ by a few per cent; a pool that grows every other frame would thrash)
        cur = torch.empty(int(need * 1.0625) + 4096, dtype=torch.uint8, device=device)
        _pool_cache[key] = cur
    cur.record_stream(torch.cuda.current_stream(device))
    return cur


def _prepass_counts(ws, device):
    """(hit_rays, unmasked, n_valid) of the prepass pending in workspace ``ws`` (host wait), or None"""
    out = (C.c_int64 * 3)()
    rc = load_library().th_render_prepass_wait(ctx(device), _p(ws), out)
    if rc == 1:
        return None
    _check(rc)
    return int(out[0]), int(out[1]), int(out[2])


def drop_workspaces(device=None):
    """Forget the cached render workspaces and the shading pool (GBs at full frame size) -- e.g. between unrelated
    workloads of one process; pending prepasses of those workspaces are cancelled."""
    lib = load_library()
    for key in list(_pool_cache.keys()):
        dev = key if isinstance(key, str) else key[0]
        if device is None or str(device) == dev:
            _pool_cache.pop(key, None)
    for (dev, _slot) in list(_ws_cache.keys()):
        if device is None or str(device) == dev:
            _ws_cache.pop((dev, _slot), None)
    try:
        _check(lib.th_render_prepass_cancel(ctx(device)))
    except HipError:
        pass


def render_prepass(points, verts_world, V, hull_thresh=0.1, small_frame_rays=2400, n_clusters=0, slot=0):
    """th_render_prepass: queue the ray-only front of render_rays (hull mask, compaction, ...) before the
    per-frame constants exist.  The following render_rays on the same `points` picks it up (and shades in the
    workspace ``slot`` the prepass wrote)."""
    lib = load_library()
    v = _f32(verts_world).reshape(-1, 3)
    dev = v.device
    f = ThFrame()
    f.verts_world, f.n_verts, f.V = v.data_ptr(), v.shape[0], V
    f.hull_thresh, f.small_frame_rays, f.map_channels = hull_thresh, small_frame_rays, 384
    f.n_clusters = n_clusters                  # (sizes the workspace exactly like the frame that follows)
    ws = _cached_ws(lib.th_render_workspace_bytes(C.byref(f), points.R, points.S), dev, slot)
    points._prepass_keep = (v, ws)
    points._prepass_nc = int(n_clusters)
    _check(lib.th_render_prepass(ctx(dev), C.byref(f), C.byref(points.c), _p(ws), ws.numel(), _stream()))
    points._prepass_pending = True          # only THIS Points object (it keeps the ray tensors alive) may consume it


def render_predemand(points, cams, scale_xy, V, H, W, verts_paint=None):
    """th_render_predemand: behind the pending render_prepass of ``points`` (on the current stream, which is ordered behind it),
    mark the map texels its valid samples -- and the ``verts_paint`` [n,3] vertices, if given -- read in the V views.  -> the demand
    buffer (device uint8 tensor) for upsample_concat_split(demand=...) / map_fold, or None (no pending prepass, or a map
    width that is not a multiple of 64)."""
    if not getattr(points, "_prepass_pending", False) or int(W) % 64 != 0 or V > 3:
        return None
    lib = load_library()
    v, ws = points._prepass_keep
    dev = v.device
    f = ThFrame()
    f.verts_world, f.n_verts, f.V, f.H, f.W = v.data_ptr(), v.shape[0], int(V), int(H), int(W)
    f.cams, f.scale_xy = cams.data_ptr(), scale_xy.data_ptr()
    f.n_clusters, f.map_channels = points._prepass_nc, 384
    vp = _f32(verts_paint).reshape(-1, 3) if verts_paint is not None else None
    nb = int(lib.th_map_demand_bytes(int(V), int(H), int(W)))
    demand = torch.empty(nb, dtype=torch.uint8, device=dev)
    rc = lib.th_render_predemand(ctx(dev), C.byref(f), C.byref(points.c), _p(ws), ws.numel(), _p(vp), vp.shape[0] if vp is not None else 0,
                                 _p(demand), nb, _stream())
    if rc == 1:
        return None
    _check(rc)
    demand._keep = (vp, cams, scale_xy)
    return demand


def render_pregrid(frame, points):
    """th_render_pregrid: the candidate grid of K4's 7-neighbour search for the frame's token centres, into the workspace of
    the pending render_prepass of ``points`` -- on the current stream (the one that produced the centres)."""
    if not getattr(points, "_prepass_pending", False):
        return
    _, ws = points._prepass_keep
    _check(load_library().th_render_pregrid(ctx(frame.verts.device), C.byref(frame.c), C.byref(points.c), _p(ws), ws.numel(),
                                            _stream()))


def render_pregather(net, frame, points, slot=0, early=False, pool_slot=None):
    """th_render_pregather: behind a render_prepass of the same ``points`` / workspace ``slot``, queue the pixel-feature
    gather and the neighbour records of the first chunks -- ``frame`` may still lack its tokens (Frame(tokens=None)),
    so TransHE can run on another stream meanwhile.  ``early=True`` (th_render_pregather_early): the caller has made the
    current stream wait for this frame's front BEFORE it queued the previous frame's render_rays, so the neighbour
    records may start as soon as that frame's per-sample stage is done, beside its compositing."""
    if not getattr(points, "_prepass_pending", False):
        return
    lib = load_library()
    _sync_weights(net, "mlp")
    dev = frame.verts.device
    _, ws = points._prepass_keep
    cnt = _prepass_counts(ws, dev)
    if cnt is None:
        return
    pool = _shade_pool(frame.c, cnt[2], True, dev, pool_slot)
    points._pregathered = True
    fn = lib.th_render_pregather_early if early else lib.th_render_pregather
    _check(fn(ctx(dev), C.byref(frame.c), C.byref(points.c), _p(ws), ws.numel(), _p(pool), pool.numel(), _stream()))


def render_rays(net, frame, points, white_bkgd=False, defer_guard=False, small_frame_rays=None, pool_slot=None):
    """th_render_rays: rays -> (rgb [R,3], acc [R], depth [R], stats).
    Range guard: the frame's snapshot of the split-arithmetic maxima is read back after the call (a host wait for
    the frame) and, if a tensor left the resolvable range, the frame is rendered again on the fp32 path.
    ``defer_guard=True`` returns a fifth value, a callable ``check() -> bool`` (True = clean), instead: a frame
    pipeline calls it after queuing the next frame so the host never idles the device.
    ``small_frame_rays``: the R' threshold of if_clight_renderer.py:551 for THIS call (applied to a copy of the
    frame descriptor; None = the frame's own value)."""
    lib = load_library()
    _sync_weights(net, "mlp")
    dev = frame.verts.device
    R = points.R
    rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
    acc = torch.empty(R, dtype=torch.float32, device=dev)
    dep = torch.empty(R, dtype=torch.float32, device=dev)
    if R == 0:                                   # empty ray list: nothing to launch (zero-size tensors have no address)
        st0 = dict(hit_rays=0, valid_samples=0, unmasked=0)
        return (rgb, acc, dep, st0, lambda: True) if defer_guard else (rgb, acc, dep, st0)
    fc = frame.c
    if small_frame_rays is not None and int(small_frame_rays) != fc.small_frame_rays:
        fc = ThFrame.from_buffer_copy(frame.c)
        fc.small_frame_rays = int(small_frame_rays)
    need = lib.th_render_workspace_bytes(C.byref(fc), R, points.S)
    n_bound, pre = R * points.S, False
    if getattr(points, "_prepass_pending", False) and points._prepass_keep[1].numel() >= need:
        points._prepass_pending = False
        ws = points._prepass_keep[1]                        # the workspace its prepass ran in
        cnt = _prepass_counts(ws, dev)                      # (on the host long ago in a frame pipeline)
        if cnt is not None:
            n_bound, pre = cnt[2], bool(getattr(points, "_pregathered", False))
    else:
        ws = _cached_ws(need, dev)
        _check(lib.th_render_prepass_drop(ctx(dev), _p(ws)))  # a token queued there for other (possibly freed) rays
    points._pregathered = False
    # the shading pool: sized from the valid-sample count when a prepass has put it on the host, else for the chunk
    # buffers only (the count bounds the chunk, not the pool)
    pool = _shade_pool(fc, n_bound, pre, dev, pool_slot)
    stats = (C.c_int64 * 4)()
    _check(lib.th_render_rays(ctx(dev), C.byref(fc), C.byref(points.c), _p(rgb), _p(acc), _p(dep), int(white_bkgd),
                              _p(ws), ws.numel(), _p(pool), pool.numel(), stats, _stream()))
    st = dict(hit_rays=stats[0], valid_samples=stats[1], unmasked=stats[3])
    slot = int(stats[2])
    if defer_guard:
        return rgb, acc, dep, st, (lambda: _guard(dev, slot))
    # a snapshot that is not clean (or was overwritten before it could be read) -> the paths the guard switched are in
    # effect now: render again and check THAT frame's snapshot too (bounded: every switch is one-way, three tries cover
    # MLP + stem + TransHE tripping one after the other)
    for _ in range(3):
        if _guard(dev, slot):
            break
        if (conv_fallback(dev) or vit_fallback(dev)) and getattr(frame, "rebuild", None) is not None:
            frame = frame.rebuild()                  # frame constants again, through the stock convolutions
            keep_sfr = fc.small_frame_rays
            fc = ThFrame.from_buffer_copy(frame.c)
            fc.small_frame_rays = keep_sfr
        _check(lib.th_render_prepass_drop(ctx(dev), _p(ws)))
        pool = _shade_pool(fc, st["valid_samples"], False, dev, pool_slot)   # (the guard may have changed the mode: other row formats)
        _check(lib.th_render_rays(ctx(dev), C.byref(fc), C.byref(points.c), _p(rgb), _p(acc), _p(dep),
                                  int(white_bkgd), _p(ws), ws.numel(), _p(pool), pool.numel(), stats, _stream()))
        st = dict(hit_rays=stats[0], valid_samples=stats[1], unmasked=stats[3])
        slot = int(stats[2])
    return rgb, acc, dep, st


def eval_sigma_grid(net, frame, pts):
    lib = load_library()
    _sync_weights(net, "mlp")
    p = _f32(pts).reshape(-1, 3)
    P = p.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=p.device)
    if P == 0:
        return out, dict(valid_samples=0)
    ws = _cached_ws(lib.th_sigma_grid_workspace_bytes(C.byref(frame.c), P), p.device)
    stats = (C.c_int64 * 4)()
    for _ in range(3):
        pool = _shade_pool(frame.c, P, False, p.device)        # (inside the loop: the guard may have changed the mode)
        _check(lib.th_eval_sigma_grid(ctx(p.device), C.byref(frame.c), _p(p), P, _p(out), _p(ws), ws.numel(), _p(pool),
                                      pool.numel(), stats, _stream()))
        if _guard(p.device, int(stats[2])):
            break
        if (conv_fallback(p.device) or vit_fallback(p.device)) and getattr(frame, "rebuild", None) is not None:
            frame = frame.rebuild()
    return out, dict(valid_samples=stats[1])


def set_tok_gather(on, device=None):
    """Token-branch hand-over K4 -> K6 on the fused path: True (default) = neighbour records, the fused kernel blends the
    rows of the per-frame token table on the matrix pipe; False = K4 blends them in fp32 (3.3 KB per sample through HBM;
    a ray shard then equals the whole frame bit for bit instead of to fp32 rounding)."""
    _check(load_library().th_set_tok_gather(ctx(device), 1 if on else 0))


_tex_rows = {}                     # device index -> mode set through set_tex_rows (default: env TH_ROWS_TEX, else on)


def tex_rows_enabled(device=None):
    """Whether new frames of this device's context take the texel hand-over (include/transhuman_hip.h: th_set_tex_rows); it
    applies to the fused path on frames with a split map (Renderer.prepare_frame's default)."""
    if _tex_rows:                      # (only a set_tex_rows call needs the device: no CUDA call on a host without one)
        d = _dev_index(device)
        if d in _tex_rows:
            return _tex_rows[d]
    return os.environ.get("TH_ROWS_TEX", "1")[:1] != "0"


def mlp_is_fused(device=None):
    """the fused fp16-split kernel shades this device's frames (mode 1 requested and the range guard has not switched it off)"""
    d = _dev_index(device)
    return _user_mode.get(d, 1) == 1 and not _range_fallback.get(d)


def set_tex_rows(on, device=None):
    """Pixel-feature hand-over K5 -> K6 on the fused path (split map): True (default) = texel lists per tile, the fused kernel
    copies the distinct texels into LDS and blends them itself (same operand bits, 160 B instead of 3.3 KB per sample through
    HBM); False = K5 writes the rows.  The shading pool is sized per mode (cached pools are re-made on demand)."""
    _check(load_library().th_set_tex_rows(ctx(device), 1 if on else 0))
    _tex_rows[_dev_index(device)] = bool(on)


def set_mlp_mode(mode, device=None):
    """1 = fused fp16-split MFMA kernel (default), 0 = layer-by-layer fp32 MFMA GEMMs."""
    _check(load_library().th_set_mlp_mode(ctx(device), int(mode)))
    d = _dev_index(device)
    _user_mode[d] = int(mode)
    _range_fallback.pop(d, None)


_fused_waves = {}


def set_fused_waves(waves, device=None):
    """Form of the fused MLP kernel on the frame-level path: 8 = two waves per SIMD (512-thread workgroups, default), 4 = the
    256-thread form of rounds 2-5."""
    _check(load_library().th_set_fused_waves(ctx(device), int(waves)))
    _fused_waves[_dev_index(device)] = int(waves)


def fused_waves(device=None):
    """what set_fused_waves last chose on this device (default: TH_FUSED_WAVES, else 8)"""
    dflt = 4 if os.environ.get("TH_FUSED_WAVES", "8")[:1] == "4" else 8
    if device is None and not torch.cuda.is_available():
        return dflt
    return _fused_waves.get(_dev_index(device), dflt)


def set_chunk_samples(n):
    """Samples per pass of the per-sample stage (default 524288; results are invariant to it)."""
    _check(load_library().th_set_chunk_samples(int(n)))


PROF_PHASES = ("hull", "dparf", "gather", "mlp", "composite", "vit", "fold", "_7")


def profile_enable(on=True, device=None):
    _check(load_library().th_profile_enable(ctx(device), int(on)))


def profile_read(device=None):
    """-> {phase: (milliseconds, launches)} accumulated since the last read."""
    ms = (C.c_double * 8)()
    cnt = (C.c_int64 * 8)()
    _check(load_library().th_profile_read(ctx(device), ms, cnt))
    return {PROF_PHASES[i]: (ms[i], cnt[i]) for i in range(7)}


def clock_probe(out):
    """th_clock_probe: queue the shader-clock probe on the current stream; ``out`` = int64 device tensor of >= 3 words
    (ticks, 10 ns units, scratch).  GHz = out[0] / (10 * out[1]) once the stream has passed it."""
    assert out.dtype == torch.int64 and out.numel() >= 3 and out.is_cuda
    _check(load_library().th_clock_probe(ctx(out.device), _p(out), _stream()))


def fused_cycles(counters, device=None):
    """th_fused_cycles: cycle accounting of the fused MLP kernel into ``counters`` (64 zeroed int64 words on the device), or None to
    switch it off.  [0] sampled tiles, [1..61] shader cycles per phase, [62] / [63] shader cycles / 100 MHz ticks per tile."""
    if counters is not None:
        assert counters.dtype == torch.int64 and counters.numel() >= 64 and counters.is_cuda
        device = counters.device
    _check(load_library().th_fused_cycles(ctx(device), _p(counters) if counters is not None else None))


def host_wait_read(device=None):
    """th_host_wait_read: host ms spent in the blocking waits of the entry points since the last call (reads and clears)"""
    ms = C.c_double()
    _check(load_library().th_host_wait_read(ctx(device), C.byref(ms)))
    return float(ms.value)


def set_vit_mode(mode, device=None):
    """th_set_vit_mode: 0 = TransHE's dense layers on the fp32 MFMA GEMMs, 1 (default) = fp16-split arithmetic."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    _check(load_library().th_set_vit_mode(ctx(dev), int(mode)))
    _vit_graph_epoch[0] += 1
    _user_vit_mode[_dev_index(dev)] = int(mode)
    if int(mode) != 0:
        _vit_fallback.pop(_dev_index(dev), None)
