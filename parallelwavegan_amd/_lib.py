"""ctypes binding of libpwgkernels.so (the C ABI declared in include/pwg_kernels.h).

The library is the ONLY compute path of this package: there is no CPU or
PyTorch-op fallback.  If it cannot be loaded, every op raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PWG_KERNEL_LIB: another build of the same library (tools/build_variant.py: compile-time kernel variants for same-box
# A/B runs); the ABI check below applies to it as well
LIB_PATH = os.environ.get("PWG_KERNEL_LIB") or os.path.join(_HERE, "libpwgkernels.so")

PWG_ACT_NONE, PWG_ACT_LEAKY_RELU, PWG_ACT_TANH, PWG_ACT_RELU = 0, 1, 2, 3
PWG_PAD_ZERO, PWG_PAD_REFLECT, PWG_PAD_REPLICATE = 0, 1, 2
ABI_VERSION = 11
SPECTRAL_NORM_SCRATCH_FLOATS = 257  # PWG_SPECTRAL_NORM_SCRATCH_FLOATS (include/pwg_kernels.h)


class ConvDesc(ctypes.Structure):
    """Mirror of ``pwg_conv1d_desc`` (include/pwg_kernels.h)."""

    _fields_ = [
        ("batch", ctypes.c_int32),
        ("c_in", ctypes.c_int32),
        ("c_out", ctypes.c_int32),
        ("t_in", ctypes.c_int32),
        ("t_out", ctypes.c_int32),
        ("width", ctypes.c_int32),
        ("kernel", ctypes.c_int32),
        ("stride", ctypes.c_int32),
        ("dilation", ctypes.c_int32),
        ("pad_left", ctypes.c_int32),
        ("groups", ctypes.c_int32),
        ("transposed", ctypes.c_int32),
        ("pad_mode", ctypes.c_int32),
        ("pre_act", ctypes.c_int32),
        ("pre_slope", ctypes.c_float),
        ("post_act", ctypes.c_int32),
        ("post_slope", ctypes.c_float),
        ("out_mul", ctypes.c_float),
        ("out_div", ctypes.c_float),
    ]


class BankItem(ctypes.Structure):
    """Mirror of ``pwg_bank_item`` (include/pwg_kernels.h)."""

    _fields_ = [("w", ctypes.c_void_p), ("g", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("fwd", ctypes.c_void_p),
                ("bwd", ctypes.c_void_p), ("desc", ConvDesc)]


class WaveNetDesc(ctypes.Structure):
    """Mirror of ``pwg_wavenet_desc`` (include/pwg_kernels.h)."""

    _fields_ = [
        ("batch", ctypes.c_int32),
        ("t", ctypes.c_int32),
        ("residual_channels", ctypes.c_int32),
        ("gate_channels", ctypes.c_int32),
        ("skip_channels", ctypes.c_int32),
        ("aux_channels", ctypes.c_int32),
        ("kernel", ctypes.c_int32),
        ("dilation", ctypes.c_int32),
        ("causal", ctypes.c_int32),
        ("out_mul", ctypes.c_float),
        ("skip_mul", ctypes.c_float),
    ]


class WaveNetParamGrad(ctypes.Structure):
    """Mirror of ``pwg_wavenet_param_grad`` (include/pwg_kernels.h)."""

    _fields_ = [("v", ctypes.c_void_p), ("g", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("dg", ctypes.c_void_p),
                ("db", ctypes.c_void_p)]


class ResUnitDesc(ctypes.Structure):
    """Mirror of ``pwg_resunit_desc`` (include/pwg_kernels.h)."""

    _fields_ = [
        ("batch", ctypes.c_int32),
        ("channels", ctypes.c_int32),
        ("t", ctypes.c_int32),
        ("kernel", ctypes.c_int32),
        ("dilation", ctypes.c_int32),
        ("has_conv2", ctypes.c_int32),
        ("slope1", ctypes.c_float),
        ("slope2", ctypes.c_float),
        ("out_div", ctypes.c_float),
    ]


class RedItem(ctypes.Structure):
    """Mirror of ``pwg_red_item`` (include/pwg_kernels.h)."""

    _fields_ = [
        ("a", ctypes.c_void_p),
        ("b", ctypes.c_void_p),
        ("da", ctypes.c_void_p),
        ("db", ctypes.c_void_p),
        ("n", ctypes.c_int64),
        ("mode", ctypes.c_int32),
        ("slot", ctypes.c_int32),
        ("scale", ctypes.c_float),
        ("c", ctypes.c_float),
    ]


RED_MAX_ITEMS = 64

_lib = None
_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_f32 = ctypes.c_float

# name -> (restype, argtypes); every symbol include/pwg_kernels.h declares
SIGNATURES = {
    "pwg_last_error": (ctypes.c_char_p, []),
    "pwg_abi_version": (ctypes.c_int, []),
    "pwg_target_arch": (ctypes.c_int, []),
    "pwg_debug_poison_lds": (ctypes.c_int, [ctypes.c_int]),
    "pwg_set_concurrency_hint": (ctypes.c_float, [ctypes.c_float]),
    "pwg_prof_enable": (ctypes.c_int, [ctypes.c_int]),
    "pwg_prof_reset": (ctypes.c_int, []),
    "pwg_prof_num_kernels": (ctypes.c_int, []),
    "pwg_prof_get": (ctypes.c_int, [_i32, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(ctypes.c_double)]),
    "pwg_conv1d_packed_weight_floats": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    "pwg_conv1d_pack_weight": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    "pwg_conv1d_forward_workspace_floats": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    "pwg_conv1d_forward": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t,
                                          _vp]),
    "pwg_conv1d_packed_weight_bwd_floats": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    "pwg_conv1d_pack_weight_bwd": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    "pwg_weight_bank_table_bytes": (ctypes.c_size_t, [_i32]),
    "pwg_weight_bank_build": (ctypes.c_int, [ctypes.POINTER(BankItem), _i32, _vp, ctypes.c_size_t,
                                             ctypes.POINTER(ctypes.c_int32)]),
    "pwg_weight_bank_prepare": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32), _i32, _vp]),
    "pwg_conv1d_backward_data_workspace_floats": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    "pwg_conv1d_backward_data": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t,
                                                _vp]),
    "pwg_conv1d_backward_weight_workspace_floats": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    "pwg_conv1d_backward_weight": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "pwg_conv1d_backward_weight_wn_workspace_floats": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    "pwg_conv1d_backward_weight_wn": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                     ctypes.c_size_t, _vp]),
    "pwg_conv1d_num_tile_configs": (ctypes.c_int, []),
    "pwg_conv1d_forward_cfg": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "pwg_conv1d_plan": (ctypes.c_int, [ctypes.POINTER(ConvDesc), _i32, ctypes.POINTER(_i32)]),
    "pwg_debug_conv_tile_of_workgroup": (ctypes.c_int, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, ctypes.POINTER(_i32)]),
    "pwg_resunit_supported": (ctypes.c_int, [ctypes.POINTER(ResUnitDesc)]),
    "pwg_resunit_profitable": (ctypes.c_int, [ctypes.POINTER(ResUnitDesc)]),
    "pwg_resunit_packed_weight_floats": (ctypes.c_size_t, [_i32, _i32]),
    "pwg_resunit_pack_weight": (ctypes.c_int, [_i32, _i32, _vp, _vp, _vp, _vp]),
    "pwg_resunit_forward": (ctypes.c_int, [ctypes.POINTER(ResUnitDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pwg_resstack_supported": (ctypes.c_int, [_i32, _i32, _i32]),
    "pwg_resstack_packed_weight_floats": (ctypes.c_size_t, [_i32]),
    "pwg_resstack_pack_weight": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pwg_resstack_forward": (ctypes.c_int, [_i32, _i32, _i32, _i32, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pwg_resstack_pack_weight_bwd": (ctypes.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pwg_resstack_backward_data": (ctypes.c_int, [_i32, _i32, _i32, _i32, ctypes.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pwg_wavenet_layer_supported": (ctypes.c_int, [ctypes.POINTER(WaveNetDesc)]),
    "pwg_wavenet_packed_weight_floats": (ctypes.c_size_t, [ctypes.POINTER(WaveNetDesc)]),
    "pwg_wavenet_pack_weights": (ctypes.c_int, [ctypes.POINTER(WaveNetDesc)] + [_vp] * 10),
    "pwg_wavenet_layer_forward": (ctypes.c_int, [ctypes.POINTER(WaveNetDesc)] + [_vp] * 12),
    "pwg_wavenet_packed_weight_bwd_floats": (ctypes.c_size_t, [ctypes.POINTER(WaveNetDesc)]),
    "pwg_wavenet_pack_weights_bwd": (ctypes.c_int, [ctypes.POINTER(WaveNetDesc)] + [_vp] * 10),
    "pwg_wavenet_gate_backward": (ctypes.c_int, [ctypes.POINTER(WaveNetDesc)] + [_vp] * 7),
    "pwg_wavenet_data_backward": (ctypes.c_int, [ctypes.POINTER(WaveNetDesc)] + [_vp] * 7),
    "pwg_wavenet_weight_backward_workspace_floats": (ctypes.c_size_t, [ctypes.POINTER(WaveNetDesc)]),
    "pwg_wavenet_weight_backward": (ctypes.c_int, [ctypes.POINTER(WaveNetDesc)] + [_vp] * 6
                                    + [ctypes.POINTER(WaveNetParamGrad), _vp, ctypes.c_size_t, _vp]),
    "pwg_act_backward": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _f32, _vp]),
    "pwg_add3_div": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _vp]),
    "pwg_wave_to_pcm16": (ctypes.c_int, [_vp, _vp, _i64, _vp]),
    "pwg_copy_channels": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i64, _i32, _vp]),
    "pwg_dropout": (ctypes.c_int, [_vp, _vp, _i64, _f32, ctypes.c_uint64, _vp, _vp]),
    "pwg_instance_norm_forward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "pwg_instance_norm_backward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "pwg_upsample_nearest_forward": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "pwg_upsample_nearest_backward": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "pwg_tade_modulate_forward": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pwg_tade_modulate_backward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pwg_softmax_gate_forward": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i64, _i32, _vp]),
    "pwg_softmax_gate_backward": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp]),
    "pwg_gather_crop": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32,
                                       _vp]),
    "pwg_normalize_transpose": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "pwg_weight_norm_backward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "pwg_spectral_norm_forward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "pwg_spectral_norm_forward_saved": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "pwg_spectral_norm_backward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "pwg_gate_forward": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i64, _vp]),
    "pwg_gate_backward": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i64, _vp]),
    "pwg_stretch_conv_forward": (ctypes.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp]),
    "pwg_stretch_conv_backward_workspace_floats": (ctypes.c_size_t, [_i32, _i32]),
    "pwg_stretch_conv_backward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp,
                                                 ctypes.c_size_t, _vp]),
    "pwg_pqmf_down": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _vp]),
    "pwg_pqmf_up": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i64, _i64, _i32, _i32, _i32, _vp]),
    "pwg_avg_pool1d_forward": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pwg_avg_pool1d_backward": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pwg_pad1d_forward": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "pwg_pad1d_backward": (ctypes.c_int, [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "pwg_frame_fold_forward": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pwg_frame_fold_backward": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pwg_stft_mag_forward": (ctypes.c_int, [_vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "pwg_stft_mag_backward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _vp]),
    "pwg_stft_loss_workspace_floats": (ctypes.c_size_t, [_i32, _i32, _i32]),
    "pwg_stft_loss_forward": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    "pwg_stft_loss_backward": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp,
                                              _vp]),
    "pwg_stft_fft_supported": (ctypes.c_int, [_i32, _i32, _i32]),
    "pwg_stft_fft_workspace_floats": (ctypes.c_size_t, [_i32, _i32, _i32]),
    "pwg_stft_fft_loss_forward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp]),
    "pwg_stft_fft_loss_backward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp,
                                                  _vp, _vp]),
    "pwg_mel_fft_loss_forward": (ctypes.c_int, [_vp] * 7 + [_i32] * 7 + [_f32, _f32, _vp, _vp, _vp]),
    "pwg_mel_fft_loss_backward": (ctypes.c_int, [_vp] * 7 + [_i32] * 7 + [_f32, _f32, _vp, _vp, _vp, _vp]),
    "pwg_mel_loss_workspace_floats": (ctypes.c_size_t, [_i32, _i32, _i32, _i32]),
    "pwg_mel_loss_forward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp,
                                            _vp, _vp, _vp, _vp]),
    "pwg_mel_loss_backward": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _f32, _f32,
                                             _vp, _vp, _vp]),
    "pwg_log_clamp_forward": (ctypes.c_int, [_vp, _vp, _i64, _f32, _f32, _vp]),
    "pwg_log_clamp_backward": (ctypes.c_int, [_vp, _vp, _vp, _i64, _f32, _f32, _vp]),
    "pwg_reduce_forward": (ctypes.c_int, [_vp, _vp, _f32, _i64, _i32, _f32, _vp, _vp, _vp]),
    "pwg_reduce_backward": (ctypes.c_int, [_vp, _vp, _f32, _i64, _i32, _f32, _vp, _vp, _vp, _vp]),
    "pwg_multi_reduce_workspace_floats": (ctypes.c_size_t, [ctypes.POINTER(RedItem), _i32]),
    "pwg_multi_reduce_forward": (ctypes.c_int, [ctypes.POINTER(RedItem), _i32, _i32, _vp, _i32, _vp, _vp]),
    "pwg_multi_reduce_backward": (ctypes.c_int, [ctypes.POINTER(RedItem), _i32, _i32, _vp, _vp]),
    "pwg_adam_step": (ctypes.c_int, [_vp, _i32, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp]),
    "pwg_radam_step": (ctypes.c_int, [_vp, _i32, _f32, _f32, _f32, _f32, _f32, _i32, _f32, _vp]),
    "pwg_adam_step_dev": (ctypes.c_int, [_vp, _i32, _vp, _vp]),
    "pwg_radam_step_dev": (ctypes.c_int, [_vp, _i32, _vp, _vp]),
    "pwg_clip_grad_norm": (ctypes.c_int, [_vp, _i32, _f32, _vp, _vp, _vp]),
    "pwg_weight_norm_scale": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
    "pwg_scale_rows": (ctypes.c_int, [_vp, _vp, _vp, _i32, _i32, _vp]),
}


def lib():
    """Load (once) and return the kernel library; raise loudly if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m parallelwavegan_amd.csrc.build` "
                "(hipcc --offload-arch=gfx950). This package has no fallback compute path."
            )
        # torch bundles its own HIP runtime; it must be the one already loaded when our
        # library's libamdhip64 dependency is resolved, or the process ends up with two
        # runtimes and ours sees "no ROCm-capable device".
        import torch  # noqa: F401

        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if l.pwg_abi_version() != ABI_VERSION:
            raise RuntimeError(
                f"libpwgkernels.so ABI {l.pwg_abi_version()} != expected {ABI_VERSION}; rebuild it"
            )
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().pwg_last_error().decode(errors="replace")
        raise RuntimeError(f"pwg kernel call failed ({what}, status {rc}): {msg}")
