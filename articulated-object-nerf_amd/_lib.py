"""ctypes binding of libaon_hip.so (C ABI declared in include/aon_hip.h).

There is NO fallback: if the shared library is missing or fails to load, importing this module raises, and so
does every op built on it.  Build it with ``python articulated-object-nerf_amd/build.py`` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os

# torch FIRST: it carries its own copy of the HIP runtime (torch/lib/libamdhip64.so), and the process must run on ONE runtime --
# the one that owns torch's allocations and streams.  Loading libaon_hip.so before torch binds it to /opt/rocm's copy instead, and
# its first launch on torch's memory fails with "no ROCm-capable device is detected" (seen when __graft_entry__.build() and
# smoke() ran in one process on a GPU box).
import torch  # noqa: F401

_PKG = os.path.dirname(os.path.abspath(__file__))
# AON_HIP_LIB: an alternative build of the SAME library (A/B experiments, tools/kernel_bench.py); never a different backend
LIB_PATH = os.environ.get("AON_HIP_LIB") or os.path.join(_PKG, "libaon_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the HIP kernels are the product path and there is no CPU/eager fallback. "
        "Build them with `python articulated-object-nerf_amd/build.py` (needs /opt/rocm/bin/hipcc)."
    )

lib = C.CDLL(LIB_PATH)

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_SIGS = {
    "aon_abi_version": (_i, []),
    "aon_last_error": (C.c_char_p, []),
    "aon_raygen": (_i, [_p, _i, _i, _f, _l, _l, _p, _p, _p, _p]),
    "aon_train_loss_fwd": (_i, [_p, _p, _p, _l, _p, _p, _f, _p, _p, _p]),
    "aon_train_loss_bwd": (_i, [_p, _p, _p, _l, _p, _p, _f, _p, _p, _p, _p, _p]),
    "aon_ray_directions": (_i, [_i, _i, _f, _p, _p]),
    "aon_get_rays": (_i, [_p, _p, _l, _p, _p, _p, _p]),
    "aon_ray_radii": (_i, [_p, _p, _i, _i, _p, _p]),
    "aon_cast_rays": (_i, [_p, _p, _p, _l, _i, _p, _p]),
    "aon_sample_along_rays": (_i, [_p, _p, _l, _i, _f, _f, _p, _p, _p, _p]),
    "aon_pos_enc": (_i, [_p, _l, _i, _i, _p, _p]),
    "aon_mlp_packed_bytes": (_l, []),
    "aon_pack_vanilla_mlp": (_i, [_p, _p, _p]),
    "aon_mlp_fwd": (_i, [_p, _p, _p, _p, _p, _l, _i, _p, _p]),
    "aon_mlp_fwd_enc": (_i, [_p, _p, _p, _l, _i, _p, _p]),
    "aon_composite": (_i, [_p, _i, _p, _i, _p, _p, _l, _i, _i, _i, _p, _p, _p, _p, _p]),
    "aon_sample_pdf": (_i, [_p, _p, _l, _p, _p, _l, _l, _p, _p, _p]),
    "aon_art_packed_bytes": (_l, []),
    "aon_art_small_bytes": (_l, []),
    "aon_pack_art_mlp": (_i, [_p, _p, _p]),
    "aon_art_prepare": (_i, [_p, _p, _p, _p, _p, _p]),
    "aon_pack_art_mlp_deg": (_i, [_p, _i, _i, _i, _p, _p]),
    "aon_art_prepare_deg": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "aon_pack_art_mlp_bwd_deg": (_i, [_p, _i, _i, _i, _p, _p]),
    "aon_art_mlp_fwd": (_i, [_p, _p, _p, _p, _p, _p, _l, _i, _p, _p]),
    "aon_art_mlp_fwd_pos": (_i, [_p, _p, _p, _p, _l, _i, _p, _p]),
    "aon_art_render_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _f, _f, _i, _i, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _l, _p]),
    "aon_train_plane_rows": (_l, []),
    "aon_bwd_packed_bytes": (_l, []),
    "aon_wgrad_workspace_bytes": (_l, []),
    "aon_pack_vanilla_mlp_bwd": (_i, [_p, _p, _p]),
    "aon_train_mask_bytes": (_l, [_l]),
    "aon_mlp_fwd_train": (_i, [_p, _p, _p, _p, _p, _l, _i, _p, _p, _p, _p]),
    "aon_composite_bwd": (_i, [_p, _p, _p, _p, _p, _p, _l, _i, _i, _i, _p, _p]),
    "aon_mlp_bwd_chain": (_i, [_p, _p, _p, _p, _p, _l, _p]),
    "aon_vanilla_wgrad": (_i, [_p, _p, _p, _l, _p, _p, _l, _p, _p]),
    "aon_wgrad_plan": (_i, [_i, _l, _i, _p, _i, _p]),
    "aon_wgrad_plan_segment": (_i, [_i, _l, _i, _i, _i, _p]),
    "aon_wgrad_kind_bench": (_i, [_i, _i, _p, _p, _i, _l, _p, _l, _p]),
    "aon_art_train_plane_rows": (_l, []),
    "aon_art_train_mask_bytes": (_l, [_l]),
    "aon_art_bwd_packed_bytes": (_l, []),
    "aon_pack_art_mlp_bwd": (_i, [_p, _p, _p]),
    "aon_art_mlp_fwd_train": (_i, [_p, _p, _p, _p, _p, _p, _l, _i, _p, _p, _p, _p]),
    "aon_art_bwd_chain": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _p]),
    "aon_art_wgrad": (_i, [_p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p]),
    "aon_art_wgrad_deg": (_i, [_p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _p, _i, _i, _i, _p]),
    "aon_set_bottleneck_fold": (_i, [_i]),
    "aon_get_bottleneck_fold": (_i, []),
    "aon_stream_is_folded": (_i, [_p]),
    "aon_stream_form": (_i, [_p]),
    "aon_declare_stream_form": (_i, [_p, _i]),
    "aon_adam_step": (_i, [_p, _p, _p, _p, _l, C.c_double, C.c_double, C.c_double, C.c_double, _l, _p]),
    "aon_code_library_fwd": (_i, [_p, _p, _p, _p, _p, _p]),
    "aon_code_library_bwd": (_i, [_p, _p, _p, _p, _p, _p]),
    "aon_art_pack_step": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "aon_vanilla_pack_step": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _p, _p]),
    "aon_set_bwd_early_heads": (_i, [_i]),
    "aon_set_view_bias": (_i, [_i]),
    "aon_get_view_bias": (_i, []),
    "aon_view_bias": (_i, [_p, _p, _l, _p, _p]),
    "aon_set_bwd_overlap": (_i, [_i]),
    "aon_set_fwd_overlap": (_i, [_i]),
    "aon_set_fwd_merge": (_i, [_i]),
    "aon_set_bwd_merge": (_i, [_i]),
    "aon_set_wgrad_probe": (_i, [_p]),
    "aon_train_workspace_bytes": (_l, [_l, _i, _i]),
    "aon_train_scratch_bytes": (_l, [_l, _i, _i]),
    "aon_render_fwd_train": (_i, [_p, _p, _p, _p, _p, _l, _f, _f, _i, _i, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _l, _p]),
    "aon_render_bwd": (_i, [_p, _p, _p, _p, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _l, _p, _l, _p]),
    "aon_art_render_fwd_train": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _f, _f, _i, _i, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _l, _p]),
    "aon_art_render_bwd": (_i, [_p, _p, _p, _p, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _p, _l, _p]),
    "aon_profile_begin": (_i, []),
    "aon_profile_end": (_i, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "aon_profile_class": (_i, [_i, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "aon_composite_pdf": (_i, [_p, _p, _p, _l, _i, _i, _p, _l, _p, _p, _p, _p, _p, _p]),
    "aon_set_coarse_fusion": (_i, [_i]),
    "aon_render_workspace_bytes": (_l, [_l]),
    "aon_render_fwd": (_i, [_p, _p, _p, _p, _p, _l, _f, _f, _i, _i, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _l, _p]),
    # constructor arguments beyond the defaults (aon_render_opts; the *_ex forms take the struct pointer last)
    "aon_render_opts_init": (None, [_p]),
    "aon_pack_vanilla_mlp_deg": (_i, [_p, _i, _i, _i, _p, _p]),
    "aon_pack_vanilla_mlp_bwd_deg": (_i, [_p, _i, _i, _i, _p, _p]),
    "aon_sample_along_rays_ex": (_i, [_p, _p, _l, _i, _f, _f, _i, _f, _f, _p, _p, _p, _p]),
    "aon_composite_ex": (_i, [_p, _i, _p, _i, _p, _p, _l, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "aon_sample_pdf_n": (_i, [_p, _p, _l, _p, _p, _l, _l, _i, _i, _i, _p, _p, _p]),
    "aon_render_workspace_bytes_ex": (_l, [_l, _p]),
    "aon_render_fwd_ex": (_i, [_p, _p, _p, _p, _p, _l, _f, _f, _i, _i, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p]),
    "aon_art_render_fwd_ex": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _f, _f, _i, _i, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p]),
    "aon_train_workspace_bytes_ex": (_l, [_l, _i, _i, _p]),
    "aon_train_scratch_bytes_ex": (_l, [_l, _i, _i, _p]),
    "aon_render_fwd_train_ex": (_i, [_p, _p, _p, _p, _p, _l, _f, _f, _i, _i, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p]),
    "aon_render_bwd_ex": (_i, [_p, _p, _p, _p, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _l, _p, _l, _p, _p]),
    "aon_art_render_fwd_train_ex": (_i, [_p, _p, _p, _p, _p, _p, _p, _l, _f, _f, _i, _i, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p]),
    "aon_art_render_bwd_ex": (_i, [_p, _p, _p, _p, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _p, _l, _p, _p]),
    # NeRFMLP of any constructor geometry (aon_mlp_geometry first)
    "aon_mlp_geometry_init": (None, [_p]),
    "aon_gmlp_param_count": (_i, [_p]),
    "aon_gmlp_workspace_bytes": (_l, [_p, _l]),
    "aon_gmlp_fwd": (_i, [_p, _p, _p, _p, _l, _i, _p, _p, _p, _l, _p]),
    "aon_grender_workspace_bytes": (_l, [_p, _l, _p]),
    "aon_grender_fwd": (_i, [_p, _p, _p, _p, _p, _p, _l, _f, _f, _i, _i, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p]),
    "aon_grender_train_workspace_bytes": (_l, [_p, _l, _i, _p]),
    "aon_grender_train_scratch_bytes": (_l, [_p, _l, _i, _p]),
    "aon_grender_fwd_train": (_i, [_p, _p, _p, _p, _p, _p, _l, _f, _f, _i, _i, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p]),
    "aon_grender_bwd": (_i, [_p, _p, _p, _p, _l, _i, _i, _p, _p, _p, _p, _p, _p, _l, _p, _l, _p, _p]),
}


class MlpGeometryC(C.Structure):
    """aon_mlp_geometry (include/aon_hip.h): the arguments of NeRFMLP.__init__ (model.py:40-54)."""
    _fields_ = [(n, C.c_int32) for n in ("min_deg_point", "max_deg_point", "deg_view", "netdepth", "netwidth", "netdepth_condition",
                                         "netwidth_condition", "skip_layer", "input_ch", "input_ch_view", "num_rgb_channels",
                                         "num_density_channels")]


class RenderOptsC(C.Structure):
    """aon_render_opts (include/aon_hip.h)."""
    _fields_ = [("num_coarse_samples", C.c_int32), ("num_fine_samples", C.c_int32), ("lindisp", C.c_int32),
                ("inv_near", C.c_float), ("inv_far", C.c_float), ("noise_std", C.c_float),
                ("noise_c", C.c_void_p), ("noise_f", C.c_void_p),
                ("rgb_scale", C.c_float), ("rgb_shift", C.c_float), ("sigma_bias", C.c_float),
                ("min_deg_point", C.c_int32), ("max_deg_point", C.c_int32), ("deg_view", C.c_int32)]


for _name, (_res, _args) in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = the .so does not export what the header declares
    _fn.restype = _res
    _fn.argtypes = _args

ABI_VERSION = 5
if lib.aon_abi_version() != ABI_VERSION:
    raise ImportError(f"libaon_hip.so ABI {lib.aon_abi_version()} != binding ABI {ABI_VERSION}; rebuild")


class AonError(RuntimeError):
    pass


# measurements: AON_BOTTLENECK_FOLD=0 in the environment starts the process in the literal two-layer form (aon_set_bottleneck_fold(0)),
# so every tool / test can be A/B'd without a flag of its own
if os.environ.get("AON_BOTTLENECK_FOLD", "") == "0":
    lib.aon_set_bottleneck_fold(0)


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.aon_last_error().decode("utf-8", "replace")
        raise AonError(f"{what or 'libaon_hip'} failed (code {rc}): {msg}")


def exported_symbols():
    return sorted(_SIGS)
