"""ctypes binding of libscmgan_hip.so (the C ABI declared in include/scm_hip.h).

The product path has no CPU fallback: if the shared library is missing or a
call fails, this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libscmgan_hip.so")

PAD_ZERO, PAD_CIRCULAR = 0, 1
ACT_NONE, ACT_LRELU, ACT_SIGMOID = 0, 1, 2
PREC_F32, PREC_F16, PREC_BF16, PREC_BF16X3 = 0, 1, 2, 3

c_float_p = ctypes.c_void_p  # raw device pointers travel as void*


class ConvDesc(ctypes.Structure):
    """Mirror of scm_conv3x3_desc (include/scm_hip.h)."""
    _fields_ = [
        ("x1", ctypes.c_void_p), ("x2", ctypes.c_void_p), ("c1", ctypes.c_int32), ("c2", ctypes.c_int32),
        ("gate1", ctypes.c_void_p), ("gate2", ctypes.c_void_p), ("gate_act", ctypes.c_int32),
        ("wpack", ctypes.c_void_p),
        ("cout", ctypes.c_int32),
        ("scale", ctypes.c_void_p),
        ("bias", ctypes.c_void_p),
        ("sbias", ctypes.c_void_p),
        ("y1", ctypes.c_void_p), ("y2", ctypes.c_void_p), ("co1", ctypes.c_int32),
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
        ("pad_mode", ctypes.c_int32), ("act", ctypes.c_int32),
        ("gate_mask_out", ctypes.c_void_p), ("gate_mask_in", ctypes.c_void_p),
        ("accumulate_y1", ctypes.c_int32),
        ("gate_mask_y1", ctypes.c_void_p), ("gate_mask_y2", ctypes.c_void_p),
        ("x_half", ctypes.c_int32), ("y_half", ctypes.c_int32),
    ]


class WgradDesc(ctypes.Structure):
    """Mirror of scm_conv3x3_wgrad_desc (include/scm_hip.h)."""
    _fields_ = [
        ("x1", ctypes.c_void_p), ("x2", ctypes.c_void_p), ("c1", ctypes.c_int32), ("c2", ctypes.c_int32),
        ("dy", ctypes.c_void_p), ("gate", ctypes.c_void_p), ("gate_act", ctypes.c_int32), ("cout", ctypes.c_int32),
        ("dw", ctypes.c_void_p), ("db", ctypes.c_void_p), ("csum", ctypes.c_void_p),
        ("part_ws", ctypes.c_void_p), ("csum_ws", ctypes.c_void_p),
        ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("pad_mode", ctypes.c_int32),
        ("transposed_dst", ctypes.c_int32), ("accumulate", ctypes.c_int32),
        ("dw_cin_total", ctypes.c_int32), ("dw_ci0", ctypes.c_int32),
        ("dot_w", ctypes.c_void_p), ("dot_part", ctypes.c_void_p),
        ("gate_mask", ctypes.c_void_p),
        ("x_half", ctypes.c_int32),
    ]


class SnLayer(ctypes.Structure):
    """Mirror of scm_sn_layer."""
    _fields_ = [("w_bar", ctypes.c_void_p), ("u", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("sigma", ctypes.c_void_p), ("inv_sigma", ctypes.c_void_p),
                ("rows", ctypes.c_int32), ("cols", ctypes.c_int32)]


class Conv2dDesc(ctypes.Structure):
    """Mirror of scm_conv2d_desc."""
    _fields_ = [("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("scale", ctypes.c_void_p),
                ("y", ctypes.c_void_p), ("gate", ctypes.c_void_p), ("gate_act", ctypes.c_int32),
                ("B", ctypes.c_int32), ("cin", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("cout", ctypes.c_int32), ("K", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32),
                ("act", ctypes.c_int32)]


class DecoderFusedDesc(ctypes.Structure):
    """Mirror of scm_decoder_fused_desc."""
    _fields_ = [("z", ctypes.c_void_p), ("pack", ctypes.c_void_p), ("target", ctypes.c_void_p),
                ("logits", ctypes.c_void_p), ("loss", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
                ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("C", ctypes.c_int32),
                ("loss_kind", ctypes.c_int32), ("h_save", ctypes.c_void_p)]


class DecoderFusedBwdDesc(ctypes.Structure):
    """Mirror of scm_decoder_fused_bwd_desc."""
    _fields_ = [("z", ctypes.c_void_p), ("pack", ctypes.c_void_p), ("logits", ctypes.c_void_p), ("target", ctypes.c_void_p),
                ("g_loss", ctypes.c_void_p), ("dz", ctypes.c_void_p),
                ("d_conv1_weight", ctypes.c_void_p), ("d_conv1_bias", ctypes.c_void_p),
                ("d_conv2_weight", ctypes.c_void_p), ("d_conv2_bias", ctypes.c_void_p), ("workspace", ctypes.c_void_p),
                ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32), ("C", ctypes.c_int32),
                ("loss_kind", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("h_saved", ctypes.c_void_p)]


class CsrnDesc(ctypes.Structure):
    """Mirror of scm_csrn_desc."""
    _fields_ = [("x", ctypes.c_void_p), ("ctx", ctypes.c_void_p), ("hprev", ctypes.c_void_p),
                ("w_ih", ctypes.c_void_p), ("w_hh", ctypes.c_void_p), ("wc", ctypes.c_void_p), ("bc", ctypes.c_void_p),
                ("B", ctypes.c_int32), ("ch", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("along_rows", ctypes.c_int32), ("reverse", ctypes.c_int32)]


_lib = None


def lib():
    """Load (once) and return the ctypes handle; raise loudly when absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "scm_gan_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        # PyTorch ships its own libamdhip64; the library must bind to THAT runtime instance (same
        # devices, streams and allocations), so torch is loaded first.  Loading this library ahead of
        # torch leaves two HIP runtimes in the process and every launch fails with hipErrorNoDevice.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        L.scm_version.restype = ctypes.c_int
        L.scm_last_error.restype = ctypes.c_char_p
        L.scm_conv3x3_pack_floats.restype = ctypes.c_size_t
        vp, ci, cf, st = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
        u64 = ctypes.c_uint64
        sigs = {
            "scm_conv3x3_pack_floats": [ci, ci],
            "scm_conv3x3": [vp, vp],
            "scm_pack_conv3x3_weights": [vp, ci, ci, ci, ci, ci, vp, vp, vp],
            "scm_conv3x3_pack_h16_bytes": [ci, ci],
            "scm_pack_conv3x3_weights_h16": [vp, ci, ci, ci, ci, ci, vp, ci, vp, ci, vp],
            "scm_conv3x3_h16": [vp, ci, vp],
            "scm_conv3x3_wgrad_workspace": [ci, ci, ci, ci, ci, vp, vp],
            "scm_conv3x3_wgrad": [vp, vp],
            "scm_conv3x3_wgrad_h16": [vp, ci, vp],
            "scm_sn_power_iter": [ci, vp, ci, vp],
            "scm_sn_fold_grad": [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp, vp],
            "scm_conv2d_fwd": [vp, vp],
            "scm_conv2d_bwd": [vp, vp, vp, vp, vp, ci, vp, vp],
            "scm_conv2d_bwd_workspace": [ci, ci, ci, ci],
            "scm_bernoulli": [vp, vp, vp, st, ci, u64, u64, vp, vp],
            "scm_factor_sum": [vp, vp, ci, ci, ci, ci, vp],
            "scm_factor_bcast": [vp, vp, ci, ci, ci, ci, vp],
            "scm_pixel_loss": [vp, vp, ci, ci, ci, vp, vp, vp],
            "scm_pixel_loss_bwd": [vp, vp, vp, ci, ci, ci, vp, vp],
            "scm_reward_tail": [vp, ci, ci, ci, vp, vp, vp],
            "scm_reward_tail_bwd": [vp, vp, vp, ci, ci, ci, vp, vp],
            "scm_action_bias": [vp, vp, vp, ci, ci, ci, ci, ci, vp, vp],
            "scm_action_bias_bwd": [vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp],
            "scm_linear": [vp, vp, vp, ci, ci, ci, ci, vp, vp],
            "scm_linear_bwd": [vp, vp, vp, vp, ci, ci, ci, ci, vp, vp, vp, vp],
            "scm_coord_cat": [vp, vp, ci, ci, ci, ci, vp],
            "scm_step_loss": [vp, vp, ctypes.c_long, vp, ctypes.c_long, vp, vp, vp, ci, ci, cf, vp, vp, vp, vp, vp],
            "scm_step_loss_bwd": [vp, vp, ctypes.c_long, vp, vp, ci, ci, cf, vp, vp, vp, vp],
            "scm_onehot": [vp, ctypes.c_long, ci, vp, vp],
            "scm_csrn_sweep_fwd": [vp, vp],
            "scm_csrn_sweep_bwd": [vp, vp, vp, vp, vp, vp],
            "scm_adam_clip_multi": [ci, vp, vp, vp, vp, vp, cf, cf, cf, cf, ci, vp, cf, cf, vp],
            "scm_conv3x3_pack_x3_bytes": [ci, ci],
            "scm_pack_conv3x3_weights_x3": [vp, ci, ci, ci, ci, ci, vp, vp, vp],
            "scm_conv3x3_x3": [vp, vp],
            "scm_conv3x3_pack_wino_floats": [ci, ci],
            "scm_pack_conv3x3_weights_wino": [vp, ci, ci, ci, ci, ci, vp, vp, vp],
            "scm_conv3x3_wino": [vp, vp],
            "scm_conv3x3_wgrad_wino": [vp, vp],
            "scm_conv3x3_wino_serves": [ci, ci],
            "scm_conv3x3_wino16_serves": [ci, ci],
            "scm_decoder_fused_pack_floats": [],
            "scm_decoder_fused_pack": [vp, vp, vp, vp, ci, ci, vp, vp],
            "scm_decoder_fused_workspace": [ci, ci, ci, ci, vp, vp],
            "scm_decoder_fused_fwd": [vp, vp],
            "scm_decoder_fused_bwd": [vp, vp],
            "scm_reward_fused_pack_floats": [],
            "scm_reward_fused_pack": [vp, vp, vp, vp, ci, ci, vp, vp],
            "scm_reward_fused_workspace": [ci, ci, ci, ci, vp],
            "scm_reward_fused_fwd": [vp, vp, vp, vp, ci, ci, ci, ci, vp],
            "scm_reward_fused_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp],
        }
        for name, args in sigs.items():
            fn = getattr(L, name)      # raises AttributeError when the library lacks a declared symbol
            fn.argtypes = args
            fn.restype = ctypes.c_size_t if name in ("scm_conv3x3_pack_floats", "scm_conv2d_bwd_workspace",
                                                  "scm_conv3x3_pack_h16_bytes", "scm_conv3x3_pack_x3_bytes", "scm_decoder_fused_pack_floats",
                                                  "scm_conv3x3_pack_wino_floats",
                                                  "scm_reward_fused_pack_floats") \
                else ctypes.c_int
        L._scm_symbols = sorted(sigs) + ["scm_version", "scm_last_error"]
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, lib().scm_last_error().decode()))


def ptr(t):
    """Device pointer of a contiguous fp32 torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_contiguous(), "scm_gan_amd ops need contiguous tensors"
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
