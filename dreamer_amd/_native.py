"""ctypes binding of libdreamer_amd.so (include/dreamer_amd.h).

The library is the product path on a GPU: there is no other implementation behind these
wrappers, and every wrapper raises if the shared object is missing or a call fails.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DRM_LIB", os.path.join(_HERE, "libdreamer_amd.so"))
_lib = None

DRM_MAX_MLP_LAYERS = 4
_f32p = C.POINTER(C.c_float)


class NativeError(RuntimeError):
    pass


class drm_ring(C.Structure):
    _fields_ = [("observations", C.c_void_p), ("actions", C.c_void_p), ("rewards", C.c_void_p),
                ("continues", C.c_void_p), ("capacity", C.c_int64), ("frame_bytes", C.c_int64)]


class drm_linear_args(C.Structure):
    _fields_ = [("X", C.c_void_p), ("ldx", C.c_int64), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p),
                ("W", C.c_void_p), ("ldw", C.c_int64), ("bias", C.c_void_p), ("table", C.c_void_p),
                ("ldt", C.c_int64), ("zidx", C.c_void_p), ("ldz", C.c_int64), ("latents", C.c_int32),
                ("classes", C.c_int32), ("Y", C.c_void_p), ("ldy", C.c_int64), ("M", C.c_int32),
                ("N", C.c_int32), ("K", C.c_int32)]


class drm_mlp(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("in_dim", C.c_int32), ("hidden", C.c_int32), ("out_dim", C.c_int32),
                ("w", C.c_void_p * DRM_MAX_MLP_LAYERS), ("b", C.c_void_p * DRM_MAX_MLP_LAYERS),
                ("ln_g", C.c_void_p * DRM_MAX_MLP_LAYERS), ("ln_b", C.c_void_p * DRM_MAX_MLP_LAYERS),
                ("head_w", C.c_void_p), ("head_b", C.c_void_p)]


class drm_dims(C.Structure):
    _fields_ = [("D", C.c_int32), ("L", C.c_int32), ("K", C.c_int32), ("M", C.c_int32), ("A", C.c_int32),
                ("E", C.c_int32), ("NB", C.c_int32), ("unimix_eps", C.c_float)]


class drm_imagine_weights(C.Structure):
    _fields_ = [("dims", drm_dims), ("a_emb", C.c_void_p), ("lin_w", C.c_void_p), ("lin_b", C.c_void_p),
                ("gru_w_ih", C.c_void_p), ("gru_w_hh", C.c_void_p), ("gru_b_ih", C.c_void_p),
                ("gru_b_hh", C.c_void_p), ("dyn", drm_mlp), ("rew", drm_mlp), ("cont", drm_mlp),
                ("actor", drm_mlp), ("critic", drm_mlp), ("buckets", C.c_void_p), ("pack", C.c_void_p),
                ("pack_bytes", C.c_size_t)]


class drm_imagine_io(C.Structure):
    _fields_ = [("B", C.c_int32), ("H", C.c_int32), ("h0", C.c_void_p), ("zidx0", C.c_void_p),
                ("expo_a", C.c_void_p), ("expo_z", C.c_void_p), ("forced_actions", C.c_void_p),
                ("forced_zidx", C.c_void_p), ("h_out", C.c_void_p), ("zidx_out", C.c_void_p),
                ("zidx_sampled_out", C.c_void_p), ("z_out", C.c_void_p), ("actions_out", C.c_void_p),
                ("actions_sampled_out", C.c_void_p), ("actor_logp_out", C.c_void_p),
                ("prior_logits_out", C.c_void_p), ("rewards_out", C.c_void_p), ("conts_out", C.c_void_p),
                ("values_out", C.c_void_p), ("v_last_out", C.c_void_p), ("lam_returns_out", C.c_void_p),
                ("gamma", C.c_double), ("lam", C.c_double), ("workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t), ("use_graph", C.c_int32), ("precision", C.c_int32),
                ("rng_state", C.c_void_p), ("rng_seed", C.c_uint64), ("rng_counter", C.c_uint64)]


class drm_observe_weights(C.Structure):
    _fields_ = [("dims", drm_dims), ("a_emb", C.c_void_p), ("lin_w", C.c_void_p), ("lin_b", C.c_void_p),
                ("gru_w_ih", C.c_void_p), ("gru_w_hh", C.c_void_p), ("gru_b_ih", C.c_void_p), ("gru_b_hh", C.c_void_p),
                ("fc_w_h", C.c_void_p), ("ld_fc", C.c_int64), ("pack", C.c_void_p), ("pack_bytes", C.c_size_t)]


class drm_observe_io(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("h0", C.c_void_p), ("zidx0", C.c_void_p), ("a_prev", C.c_void_p),
                ("c_prev", C.c_void_p), ("expo_reset", C.c_void_p), ("expo_z", C.c_void_p), ("e_part", C.c_void_p),
                ("h_out", C.c_void_p), ("zidx_out", C.c_void_p), ("zidx_in_out", C.c_void_p), ("z_out", C.c_void_p),
                ("post_out", C.c_void_p), ("y_out", C.c_void_p), ("gates_out", C.c_void_p), ("x_out", C.c_void_p),
                ("h_in_out", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("use_graph", C.c_int32), ("precision", C.c_int32)]


class drm_observe_bwd_io(C.Structure):
    _fields_ = [("B", C.c_int32), ("T", C.c_int32)] + [(n, C.c_void_p) for n in (
        "c_prev", "y", "wmix", "psoft", "gates", "h_in", "g_h", "g_z", "g_post", "fc_w_hT", "gru_w_ihT", "gru_w_hhT",
        "lin_zT", "d_gi", "d_gh", "d_x", "d_post", "workspace")] + [("workspace_bytes", C.c_size_t), ("precision", C.c_int32), ("gru_w_ih", C.c_void_p)]


# name -> (restype, argtypes); must list every symbol include/dreamer_amd.h declares
SIGNATURES = {
    "drm_version": (C.c_int, []),
    "drm_last_error": (C.c_char_p, []),
    "drm_arch_ok": (C.c_int, []),
    "drm_prof_enable": (C.c_int, [C.c_int]),
    "drm_prof_class_name": (C.c_char_p, [C.c_int]),
    "drm_prof_summary": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double)]),
    "drm_lambda_return": (C.c_int, [C.c_void_p] * 5 + [C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_void_p]),
    "drm_ln_silu_forward": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_void_p]),
    "drm_ln_silu_slices": (C.c_int32, [C.c_int64, C.c_int32]),
    "drm_ln_silu_sums_floats": (C.c_size_t, [C.c_int64, C.c_int32]),
    "drm_ln_silu_backward": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p]),
    "drm_seq_gather": (C.c_int, [C.POINTER(drm_ring), C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 6),
    "drm_sample_categorical_st": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "drm_onehot_to_index": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "drm_kl_pair_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_void_p]),
    "drm_kl_pair_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float]
                             + [C.c_void_p] * 5),
    "drm_mse_rows_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "drm_mse_rows_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "drm_critic_loss_forward": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                          C.c_void_p]),
    "drm_critic_loss_backward": (C.c_int, [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                           C.c_void_p]),
    "drm_ema_multi": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_double,
                                C.c_void_p]),
    "drm_percentile_scale": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "drm_convt_k4s2_forward": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] + [C.c_int32] * 4 + [C.c_void_p]),
    "drm_convt_k4s2_input_grad": (C.c_int, [C.c_void_p] * 3 + [C.c_int64] + [C.c_int32] * 5 + [C.c_void_p]),
    "drm_conv_img_down": (C.c_int, [C.c_void_p] * 3 + [C.c_int64] + [C.c_int32] * 4 + [C.c_void_p]),
    "drm_conv_img_up": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] + [C.c_int32] * 4 + [C.c_void_p]),
    "drm_conv_img_wgrad_workspace_bytes": (C.c_size_t, [C.c_int64] + [C.c_int32] * 4),
    "drm_conv_img_wgrad": (C.c_int, [C.c_void_p] * 4 + [C.c_size_t, C.c_int64] + [C.c_int32] * 4 + [C.c_void_p]),
    "drm_conv4s2_down_prec": (C.c_int, [C.c_void_p] * 3 + [C.c_int64] + [C.c_int32] * 5 + [C.c_void_p]),
    "drm_conv4s2_up_prec": (C.c_int, [C.c_void_p] * 3 + [C.c_int64] + [C.c_int32] * 5 + [C.c_void_p]),
    "drm_conv4s2_wgrad_prec": (C.c_int, [C.c_void_p] * 3 + [C.c_int64] + [C.c_int32] * 4 + [C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "drm_conv4s2_down": (C.c_int, [C.c_void_p] * 3 + [C.c_int64] + [C.c_int32] * 4 + [C.c_void_p]),
    "drm_conv4s2_up": (C.c_int, [C.c_void_p] * 3 + [C.c_int64] + [C.c_int32] * 4 + [C.c_void_p]),
    "drm_conv4s2_wgrad_workspace_bytes": (C.c_size_t, [C.c_int64] + [C.c_int32] * 4),
    "drm_conv4s2_wgrad": (C.c_int, [C.c_void_p] * 3 + [C.c_int64] + [C.c_int32] * 4 + [C.c_void_p, C.c_size_t, C.c_void_p]),
    "drm_adam_workspace_bytes": (C.c_size_t, []),
    "drm_adam_flat_step": (C.c_int, [C.c_void_p] * 4 + [C.c_int64] + [C.c_double] * 5 + [C.c_void_p] * 4),
    "drm_copy_pair": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "drm_copy_pair_rng": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint64,
                                    C.c_void_p]),
    "drm_fused_linear_f32": (C.c_int, [C.POINTER(drm_linear_args), C.c_void_p]),
    "drm_fused_linear_bf16": (C.c_int, [C.POINTER(drm_linear_args), C.c_void_p]),
    "drm_fused_linear_split": (C.c_int, [C.POINTER(drm_linear_args), C.c_void_p]),
    "drm_gemm_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "drm_gemm_split_workspace_bytes": (C.c_size_t, [C.c_int32] * 4),
    "drm_gemm_split_f32": (C.c_int, [C.c_int32] * 5 + [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "drm_colsum_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "drm_gemm_f32": (C.c_int, [C.c_int32] * 5 + [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_void_p, C.c_size_t, C.c_void_p]),
    "drm_pack_imagine_bytes": (C.c_size_t, [C.POINTER(drm_dims)]),
    "drm_pack_imagine": (C.c_int, [C.POINTER(drm_imagine_weights), C.c_void_p, C.c_size_t, C.c_void_p]),
    "drm_imagine_workspace_bytes": (C.c_size_t, [C.POINTER(drm_dims), C.c_int32, C.c_int32]),
    "drm_imagine_block_f32": (C.c_int, [C.POINTER(drm_imagine_weights), C.POINTER(drm_imagine_io), C.c_void_p]),
    "drm_graph_cache_clear": (C.c_int, []),
    "drm_imagine_block_noalloc_f32": (C.c_int, [C.POINTER(drm_imagine_weights), C.POINTER(drm_imagine_io), C.c_void_p]),
    "drm_pack_observe_bytes": (C.c_size_t, [C.POINTER(drm_dims)]),
    "drm_pack_observe": (C.c_int, [C.POINTER(drm_observe_weights), C.c_void_p, C.c_size_t, C.c_void_p]),
    "drm_pack_observe_prec": (C.c_int, [C.POINTER(drm_observe_weights), C.c_void_p, C.c_size_t, C.c_int32, C.c_void_p]),
    "drm_observe_workspace_bytes": (C.c_size_t, [C.POINTER(drm_dims), C.c_int32, C.c_int32]),
    "drm_observe_forward_f32": (C.c_int, [C.POINTER(drm_observe_weights), C.POINTER(drm_observe_io), C.c_void_p]),
    "drm_observe_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(drm_dims), C.c_int32]),
    "drm_observe_backward_f32": (C.c_int, [C.POINTER(drm_dims), C.POINTER(drm_observe_bwd_io), C.c_void_p]),
}


def lib():
    """The loaded shared library; raises NativeError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                              f"g.build()'` (or `make -C dreamer_amd/csrc`). There is no fallback path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise NativeError(f"{what} failed with code {rc}: {lib().drm_last_error().decode()}")


def stream_ptr(device=None) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def require_cuda(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NativeError("this operation runs only on a HIP device tensor")


def _extent(t: torch.Tensor) -> int:
    """Bytes between the tensor's first element and the end of the storage it lives in."""
    return t.untyped_storage().nbytes() - t.storage_offset() * t.element_size()


def _require_bytes(t: Optional[torch.Tensor], nbytes: int, name: str) -> None:
    """The kernel behind a wrapper addresses ``nbytes`` from ``t``'s first element: refuse the call when the storage is shorter."""
    if t is not None and _extent(t) < nbytes:
        raise NativeError(f"{name}: the kernel addresses {nbytes} bytes, the tensor's storage holds {_extent(t)} from its first element")


def _dense(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if t.dtype != dtype or not t.is_contiguous():
        raise NativeError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    return t


# --------------------------------------------------------------------------- thin wrappers
def lambda_return(R, C_, V, V_last, gamma: float, lam: float) -> torch.Tensor:
    require_cuda(R, C_, V, V_last)
    B, H = R.shape
    for n, t in (("R", R), ("C", C_), ("V", V), ("V_last", V_last)):
        _dense(t, torch.float32, n)
    out = torch.empty_like(R)
    check(lib().drm_lambda_return(R.data_ptr(), C_.data_ptr(), V.data_ptr(), V_last.data_ptr(), out.data_ptr(),
                                  B, H, float(gamma), float(lam), stream_ptr(R.device)), "drm_lambda_return")
    return out


def seq_gather(obs, actions, rewards, continues, starts, T: int, want_f32: bool = False):
    require_cuda(obs, actions, rewards, continues, starts)
    cap = obs.shape[0]
    frame = obs[0].numel()
    _dense(obs, torch.uint8, "observations"); _dense(actions, torch.int64, "actions")
    _dense(rewards, torch.float32, "rewards"); _dense(continues, torch.bool, "continues")
    _dense(starts, torch.int64, "starts")
    B = starts.numel()
    dev = obs.device
    o = torch.empty((B, T, *obs.shape[1:]), dtype=torch.uint8, device=dev)
    a = torch.empty((B, T), dtype=torch.int64, device=dev)
    r = torch.empty((B, T), dtype=torch.float32, device=dev)
    c = torch.empty((B, T), dtype=torch.bool, device=dev)
    f = torch.empty((B, T, *obs.shape[1:]), dtype=torch.float32, device=dev) if want_f32 else None
    ring = drm_ring(obs.data_ptr(), actions.data_ptr(), rewards.data_ptr(), continues.data_ptr(), cap, frame)
    check(lib().drm_seq_gather(C.byref(ring), starts.data_ptr(), B, T, o.data_ptr(), a.data_ptr(), r.data_ptr(),
                               c.data_ptr(), ptr(f), stream_ptr(dev)), "drm_seq_gather")
    return o, a, r, c, f


def _is_channels_last(x: torch.Tensor) -> bool:
    return x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)


def _dense_or_cl(x: torch.Tensor, name: str) -> bool:
    """True for a channels-last (N, C, H, W) tensor, False for a plain contiguous one; anything else is rejected."""
    if _is_channels_last(x):
        if x.dtype != torch.float32:
            raise ValueError(f"{name} must be float32")
        return True
    _dense(x, torch.float32, name)
    return False


def ln_silu_forward(x, gamma, beta, eps: float, channel_bias=None):
    """y = SiLU(LayerNorm over each whole sample of x (+ channel_bias[c])); returns y (x's memory format), mean (N), rstd (N)."""
    require_cuda(x, gamma, beta)
    cl = _dense_or_cl(x, "x")
    _dense(gamma, torch.float32, "gamma"); _dense(beta, torch.float32, "beta")
    N, E = x.shape[0], gamma.numel()
    hw = 0
    if cl:
        hw = x.shape[2] * x.shape[3]
    if channel_bias is not None:
        _dense(channel_bias, torch.float32, "channel_bias")
        hw = E // channel_bias.numel()
    if x.numel() != N * E:
        raise NativeError(f"ln_silu_forward: x has {x.numel()} elements, expected {N} samples of {E}")
    if channel_bias is not None and (E % channel_bias.numel() or (cl and channel_bias.numel() != x.shape[1])):
        raise NativeError("ln_silu_forward: channel bias does not match the sample shape")
    _require_bytes(beta, 4 * E, "beta")
    y = torch.empty_like(x)
    mean = torch.empty(N, dtype=torch.float32, device=x.device)
    rstd = torch.empty(N, dtype=torch.float32, device=x.device)
    check(lib().drm_ln_silu_forward(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                    rstd.data_ptr(), N, E, float(eps), ptr(channel_bias), hw, int(cl), stream_ptr(x.device)),
          "drm_ln_silu_forward")
    return y, mean, rstd


def ln_silu_backward(x, dy, gamma, beta, mean, rstd, channel_bias=None, want_dx_sum=False):
    """dx and [d gamma; d beta] (2, E) of ln_silu_forward; with a channel bias a third result, d bias (C);
    with ``want_dx_sum`` a third result sum_n dx[n] (E).  ``dy`` must have x's memory format."""
    require_cuda(x, dy, gamma, beta, mean, rstd)
    cl = _dense_or_cl(x, "x")
    if _dense_or_cl(dy, "dy") != cl:
        raise ValueError("dy must have the memory format of x")
    N, E = x.shape[0], gamma.numel()
    if x.numel() != N * E or dy.numel() != N * E or beta.numel() != E or mean.numel() < N or rstd.numel() < N:
        raise NativeError(f"ln_silu_backward: operand sizes do not match {N} samples of {E}")
    if channel_bias is not None and (E % channel_bias.numel() or (cl and channel_bias.numel() != x.shape[1])):
        raise NativeError("ln_silu_backward: channel bias does not match the sample shape")
    slices = lib().drm_ln_silu_slices(N, E)
    dx = torch.empty_like(x)
    rows = 2 if (channel_bias is None and not want_dx_sum) else 3
    part = torch.empty((slices, rows, E), dtype=torch.float32, device=x.device)
    sums = None if rows == 2 else torch.empty(lib().drm_ln_silu_sums_floats(N, E), dtype=torch.float32, device=x.device)
    hw = x.shape[2] * x.shape[3] if cl else (0 if channel_bias is None else E // channel_bias.numel())
    check(lib().drm_ln_silu_backward(x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                     rstd.data_ptr(), dx.data_ptr(), part.data_ptr(), N, E, ptr(channel_bias), hw, int(cl),
                                     ptr(sums), stream_ptr(x.device)), "drm_ln_silu_backward")
    total = sum_rows(part.view(slices, rows * E)).view(rows, E)
    if rows == 2:
        return dx, total
    if channel_bias is None:
        return dx, total[:2], total[2]
    return dx, total[:2], total[2].view(channel_bias.numel(), E // channel_bias.numel()).sum(1)


def sample_categorical_st(logits, expo, unimix_eps: Optional[float], want_z=True, want_idx=True, want_y=False):
    require_cuda(logits, expo)
    _dense(logits, torch.float32, "logits"); _dense(expo, torch.float32, "expo")
    K = logits.shape[-1]
    rows = logits.numel() // K
    dev = logits.device
    z = torch.empty_like(logits) if want_z else None
    idx = torch.empty(logits.shape[:-1], dtype=torch.uint8, device=dev) if want_idx else None
    y = torch.empty_like(logits) if want_y else None
    eps = -1.0 if unimix_eps is None else float(unimix_eps)
    check(lib().drm_sample_categorical_st(logits.data_ptr(), expo.data_ptr(), rows, K, eps, ptr(z), ptr(idx), ptr(y),
                                          stream_ptr(dev)), "drm_sample_categorical_st")
    return z, idx, y


def onehot_to_index(z: torch.Tensor) -> torch.Tensor:
    require_cuda(z)
    _dense(z, torch.float32, "z")
    K = z.shape[-1]
    idx = torch.empty(z.shape[:-1], dtype=torch.uint8, device=z.device)
    check(lib().drm_onehot_to_index(z.data_ptr(), z.numel() // K, K, idx.data_ptr(), stream_ptr(z.device)),
          "drm_onehot_to_index")
    return idx


def convt_k4s2_forward(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """ConvTranspose2d(kernel 4, stride 2, padding 1) + bias for at most 4 output channels."""
    require_cuda(x, weight)
    _dense(x, torch.float32, "x"); _dense(weight, torch.float32, "weight")
    n, cin, h, w = x.shape
    cout = weight.shape[1]
    if tuple(weight.shape) != (cin, cout, 4, 4) or (bias is not None and (bias.numel() != cout or bias.dtype != torch.float32)):
        raise NativeError(f"convt_k4s2_forward: weight {tuple(weight.shape)} / bias do not match {cin} input channels")
    y = torch.empty(n, cout, 2 * h, 2 * w, dtype=torch.float32, device=x.device)
    check(lib().drm_convt_k4s2_forward(x.data_ptr(), weight.data_ptr(), ptr(bias), y.data_ptr(), n, cin, cout, h, w,
                                       stream_ptr(x.device)), "drm_convt_k4s2_forward")
    return y


def conv_img_ok(cs: int, cimg: int, up: bool = False) -> bool:
    """Whether the image-side kernels take a layer with `cimg` image channels and `cs` feature channels."""
    return 1 <= cimg <= 4 and cs % 32 == 0 and 32 <= cs <= 512


def _img_operands(small, big, weight, what):
    """Shapes of the image-side layer: small (N, Cs, H, W) logical in channels-last memory, big (N, Cimg, 2H, 2W) NCHW."""
    n, cs, h, w = small.shape
    cimg = big.shape[1]
    if tuple(big.shape) != (n, cimg, 2 * h, 2 * w) or (weight is not None and tuple(weight.shape) != (cs, cimg, 4, 4)):
        raise NativeError(f"{what}: feature map {tuple(small.shape)}, image {tuple(big.shape)}"
                          + (f" and weight {tuple(weight.shape)}" if weight is not None else "") + " do not fit one k4 s2 p1 layer")
    if not conv_img_ok(cs, cimg):
        raise NativeError(f"{what}: {cimg} image channels / {cs} feature channels are outside the kernels' range")
    return n, cs, cimg, h, w


def conv_img_down(big: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """conv2d(big, weight, stride 2, padding 1) without bias for an image with at most 4 channels: big (N, Cimg, 2H, 2W)
    contiguous, weight (Cs, Cimg, 4, 4); the result (N, Cs, H, W) is in channels-last memory."""
    require_cuda(big, weight)
    _dense(big, torch.float32, "big"); _dense(weight, torch.float32, "weight")
    n, cimg, h2, w2 = big.shape
    cs = weight.shape[0]
    if h2 % 2 or w2 % 2:
        raise NativeError(f"conv_img_down: image {tuple(big.shape)} needs even height and width")
    out = torch.empty((n, cs, h2 // 2, w2 // 2), dtype=torch.float32, device=big.device, memory_format=torch.channels_last)
    _img_operands(out, big, weight, "conv_img_down")
    check(lib().drm_conv_img_down(big.data_ptr(), weight.data_ptr(), out.data_ptr(), n, cs, cimg, h2 // 2, w2 // 2,
                                  stream_ptr(big.device)), "drm_conv_img_down")
    return out


def conv_img_up(small: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """conv_transpose2d(small, weight, stride 2, padding 1) + bias for at most 4 output channels: small (N, Cs, H, W) in
    channels-last memory, weight (Cs, Cimg, 4, 4); the result (N, Cimg, 2H, 2W) is contiguous."""
    require_cuda(small, weight)
    small = _cl4(small, "small")
    _dense(weight, torch.float32, "weight")
    n, cs, h, w = small.shape
    cimg = weight.shape[1]
    if bias is not None:
        _dense(bias, torch.float32, "bias")
        _require_bytes(bias, 4 * cimg, "bias")
    out = torch.empty((n, cimg, 2 * h, 2 * w), dtype=torch.float32, device=small.device)
    _img_operands(small, out, weight, "conv_img_up")
    if not conv_img_ok(cs, cimg, up=True):
        raise NativeError(f"conv_img_up: {cs} feature channels are outside the kernel's range")
    check(lib().drm_conv_img_up(small.data_ptr(), weight.data_ptr(), ptr(bias), out.data_ptr(), n, cs, cimg, h, w,
                                stream_ptr(small.device)), "drm_conv_img_up")
    return out


def conv_img_wgrad(small: torch.Tensor, big: torch.Tensor) -> torch.Tensor:
    """Weight gradient (Cs, Cimg, 4, 4) of the image-side layer: sum over images and positions of small[n, cs, i, j] *
    big[n, cimg, 2i - 1 + ky, 2j - 1 + kx]; small in channels-last memory, big contiguous."""
    require_cuda(small, big)
    small = _cl4(small, "small")
    _dense(big, torch.float32, "big")
    n, cs, cimg, h, w = _img_operands(small, big, None, "conv_img_wgrad")
    need = lib().drm_conv_img_wgrad_workspace_bytes(n, cs, cimg, h, w)
    ws = torch.empty(need // 4 + 4, dtype=torch.float32, device=small.device)
    gw = torch.empty((cs, cimg, 4, 4), dtype=torch.float32, device=small.device)
    check(lib().drm_conv_img_wgrad(small.data_ptr(), big.data_ptr(), gw.data_ptr(), ws.data_ptr(), ws.numel() * 4, n, cs, cimg,
                                   h, w, stream_ptr(small.device)), "drm_conv_img_wgrad")
    return gw


def convt_k4s2_input_grad(grad_y: torch.Tensor, weight: torch.Tensor, channels_last: bool) -> torch.Tensor:
    """Gradient wrt the input of that layer from the gradient wrt its output (N, Cout, 2H, 2W); the result has the
    logical shape (N, Cin, H, W), in channels-last memory when asked for (Cin a multiple of 4)."""
    require_cuda(grad_y, weight)
    _dense(grad_y, torch.float32, "grad_y"); _dense(weight, torch.float32, "weight")
    n, cout, ho, wo = grad_y.shape
    cin = weight.shape[0]
    if tuple(weight.shape) != (cin, cout, 4, 4) or ho % 2 or wo % 2:
        raise NativeError(f"convt_k4s2_input_grad: weight {tuple(weight.shape)} does not match a gradient of shape {tuple(grad_y.shape)}")
    channels_last = bool(channels_last) and cin % 4 == 0
    gx = torch.empty((n, cin, ho // 2, wo // 2), dtype=torch.float32, device=grad_y.device,
                     memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    check(lib().drm_convt_k4s2_input_grad(grad_y.data_ptr(), weight.data_ptr(), gx.data_ptr(), n, cin, cout, ho // 2, wo // 2,
                                          1 if channels_last else 0, stream_ptr(grad_y.device)), "drm_convt_k4s2_input_grad")
    return gx


def copy_pair(a_dst: torch.Tensor, a_src: torch.Tensor, b_dst: torch.Tensor, b_src: torch.Tensor,
              rng_state: Optional[torch.Tensor] = None, rng_counter: int = 0) -> None:
    """a_dst <- a_src and b_dst <- b_src (dense tensors of equal byte size each) in one launch; with ``rng_state`` (one
    int64 on the device) the same launch stores ``rng_counter`` there."""
    require_cuda(a_dst)
    for dst, src in ((a_dst, a_src), (b_dst, b_src)):
        if not (dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.numel() == src.numel()):
            raise ValueError("copy_pair needs dense tensors of the same dtype and size")
    if rng_state is not None and not (rng_state.is_cuda and rng_state.dtype == torch.int64 and rng_state.numel() == 1):
        raise ValueError("rng_state must be one int64 on the device")
    check(lib().drm_copy_pair_rng(a_src.data_ptr(), a_dst.data_ptr(), a_dst.numel() * a_dst.element_size(),
                                  b_src.data_ptr(), b_dst.data_ptr(), b_dst.numel() * b_dst.element_size(),
                                  ptr(rng_state), int(rng_counter) & 0xFFFFFFFFFFFFFFFF, stream_ptr(a_dst.device)),
          "drm_copy_pair_rng")


def fused_linear(x, w, bias=None, ln=None, table=None, zidx=None, classes: int = 32, w_cols=None, bf16: bool = False,
                 split: bool = False):
    """Y = f(x) @ w[:, :K]^T + bias (+ one-hot table rows).  ``w`` may be wider than K (``w_cols`` = K).
    ``bf16``: bf16 matrix inputs with fp32 accumulation (no LayerNorm prologue).
    ``split``: fp32 accuracy from an exact three-way bf16 split of both operands (no LayerNorm prologue)."""
    require_cuda(x, w)
    M, K = x.shape
    N = w.shape[0]
    if w_cols is not None:
        assert w_cols == K
    if x.dim() != 2 or w.dim() != 2 or w.shape[1] < K or x.stride(1) != 1 or w.stride(1) != 1 or x.dtype != torch.float32 \
            or w.dtype != torch.float32:
        raise NativeError("fused_linear: x (M, K) and w (N, >= K) must be fp32 matrices with contiguous rows")
    _require_bytes(bias, 4 * N, "bias")
    if ln is not None:
        _require_bytes(ln[0], 4 * K, "ln gamma"); _require_bytes(ln[1], 4 * K, "ln beta")
    if table is not None:
        if zidx is None or zidx.dtype != torch.uint8 or zidx.dim() != 2 or zidx.shape[0] != M or zidx.stride(1) != 1:
            raise NativeError("fused_linear: a gather table needs uint8 class indices (M, latents)")
        if table.dim() != 2 or table.shape[0] < zidx.shape[1] * classes or table.shape[1] < N or table.stride(1) != 1:
            raise NativeError("fused_linear: the gather table must be (latents * classes, >= N)")
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    a = drm_linear_args()
    a.X, a.ldx = x.data_ptr(), x.stride(0)
    if ln is not None:
        a.ln_gamma, a.ln_beta = ln[0].data_ptr(), ln[1].data_ptr()
    a.W, a.ldw = w.data_ptr(), w.stride(0)
    a.bias = ptr(bias)
    if table is not None:
        a.table, a.ldt = table.data_ptr(), table.stride(0)
        a.zidx, a.ldz = zidx.data_ptr(), zidx.stride(0)
        a.latents, a.classes = zidx.shape[1], classes
    a.Y, a.ldy = y.data_ptr(), y.stride(0)
    a.M, a.N, a.K = M, N, K
    if split:
        check(lib().drm_fused_linear_split(C.byref(a), stream_ptr(x.device)), "drm_fused_linear_split")
    elif bf16:
        check(lib().drm_fused_linear_bf16(C.byref(a), stream_ptr(x.device)), "drm_fused_linear_bf16")
    else:
        check(lib().drm_fused_linear_f32(C.byref(a), stream_ptr(x.device)), "drm_fused_linear_f32")
    return y


def ema_multi(srcs, dsts, decay: float) -> None:
    """dst <- dst * decay + src * (1 - decay) for every (src, dst) pair of dense fp32 device tensors, one launch."""
    n = len(dsts)
    if n == 0:
        return
    require_cuda(*srcs, *dsts)
    for s, d in zip(srcs, dsts):
        _dense(s, torch.float32, "src"); _dense(d, torch.float32, "dst")
        if s.numel() != d.numel():
            raise ValueError("ema_multi: tensors of a pair differ in size")
    sp = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
    dp = (C.c_void_p * n)(*[t.data_ptr() for t in dsts])
    ne = (C.c_int64 * n)(*[t.numel() for t in dsts])
    check(lib().drm_ema_multi(n, sp, dp, ne, float(decay), stream_ptr(dsts[0].device)), "drm_ema_multi")


def _gemm_operand_ok(t: torch.Tensor) -> bool:
    return (t.dim() == 2 and t.dtype == torch.float32 and t.is_cuda and t.stride(1) == 1 and t.stride(0) % 4 == 0
            and t.stride(0) >= t.shape[1] and t.data_ptr() % 16 == 0)


def gemm_ok(a: torch.Tensor, b: torch.Tensor, trans_a: bool, trans_b: bool) -> bool:
    """Whether ``gemm`` takes these operands (dense-row fp32 device matrices, 16-byte aligned rows)."""
    if not (_gemm_operand_ok(a) and _gemm_operand_ok(b)):
        return False
    k = a.shape[0] if trans_a else a.shape[1]
    return (trans_a and trans_b) or k % 4 == 0


_gemm_ws = {}


def gemm(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False, bias: Optional[torch.Tensor] = None):
    """op(a) @ op(b) (+ bias) on the library's fp32 MFMA kernel.
    trans_a=False: a is (M, K); True: a is (K, M).  trans_b=False: b is (N, K) (a Linear weight); True: b is (K, N)."""
    K, M = (a.shape[0], a.shape[1]) if trans_a else (a.shape[1], a.shape[0])
    Kb, N = (b.shape[0], b.shape[1]) if trans_b else (b.shape[1], b.shape[0])
    if K != Kb:
        raise ValueError(f"gemm: reduction extents differ ({K} vs {Kb})")
    if not gemm_ok(a, b, trans_a, trans_b):
        raise NativeError("gemm: operands must be fp32 device matrices with contiguous, 16-byte aligned rows")
    if bias is not None:
        _dense(bias, torch.float32, "bias")
        _require_bytes(bias, 4 * N, "bias")
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    need = lib().drm_gemm_workspace_bytes(M, N, K)
    ws = None
    if need:
        # one scratch buffer per device, grown on demand; launches on one stream use it one after another
        held = _gemm_ws.setdefault(a.device, [])
        if not held or held[-1].numel() * 4 < need:
            # earlier, smaller buffers stay alive: a recorded graph may still point at them
            held.append(torch.empty(need // 4 + 4, dtype=torch.float32, device=a.device))
        ws = held[-1]
    check(lib().drm_gemm_f32(int(trans_a), int(trans_b), M, N, K, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), ptr(bias),
                             c.data_ptr(), c.stride(0), ptr(ws), 0 if ws is None else ws.numel() * 4, stream_ptr(a.device)),
          "drm_gemm_f32")
    return c


def gemm_split_ok(a: torch.Tensor, b: torch.Tensor, trans_a: bool) -> bool:
    """Whether ``gemm_split`` takes (a, b): fp32 device matrices with contiguous 16-byte aligned rows, b (K, N) with N % 4 == 0,
    a (K, M) with M % 4 == 0 (trans_a) or (M, K) with K % 32 == 0."""
    for t in (a, b):
        if not (t.is_cuda and t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(0) % 4 == 0
                and t.stride(0) >= t.shape[1] and t.data_ptr() % 16 == 0):
            return False
    if b.shape[1] % 4:
        return False
    if trans_a:
        return a.shape[0] == b.shape[0] and a.shape[1] % 4 == 0
    return a.shape[1] == b.shape[0] and a.shape[1] % 32 == 0


def gemm_split(a: torch.Tensor, b: torch.Tensor, trans_a: bool, bf16: bool = False) -> torch.Tensor:
    """a.T @ b (trans_a: a is (K, M)) or a @ b (a is (M, K)) with b (K, N): fp32 results on the bf16 matrix pipe (both operands
    split exactly into three bf16 parts; ``bf16``: rounded to bf16 instead), no transposed copies -- the weight and input gradients
    of a large Linear layer."""
    if not gemm_split_ok(a, b, trans_a):
        raise NativeError("gemm_split: operands must be fp32 device matrices with contiguous, 16-byte aligned rows and matching extents")
    M = a.shape[1] if trans_a else a.shape[0]
    K, Nn = b.shape
    c = torch.empty((M, Nn), dtype=torch.float32, device=a.device)
    need = lib().drm_gemm_split_workspace_bytes(int(trans_a), M, Nn, K)
    ws = None
    if need:
        held = _gemm_ws.setdefault(a.device, [])
        if not held or held[-1].numel() * 4 < need:
            held.append(torch.empty(need // 4 + 4, dtype=torch.float32, device=a.device))
        ws = held[-1]
    check(lib().drm_gemm_split_f32(int(trans_a), int(bool(bf16)), M, Nn, K, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0),
                                   ptr(ws), 0 if ws is None else ws.numel() * 4, stream_ptr(a.device)), "drm_gemm_split_f32")
    return c


def colsum_ok(x: torch.Tensor) -> bool:
    """Whether ``colsum`` takes x: a 2-D fp32 device matrix with contiguous 16-byte aligned rows, columns a multiple of 4."""
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= 1 and x.shape[1] >= 4 and x.shape[1] % 4 == 0
            and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.stride(0) >= x.shape[1] and x.data_ptr() % 16 == 0)


def colsum(x: torch.Tensor) -> torch.Tensor:
    """x.sum(0) of a (rows, cols) matrix on the library's kernel (rows added in a fixed order)."""
    if not colsum_ok(x):
        raise NativeError("colsum: operand must be an fp32 device matrix with contiguous, 16-byte aligned rows and cols % 4 == 0")
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    check(lib().drm_colsum_f32(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], out.data_ptr(), stream_ptr(x.device)), "drm_colsum_f32")
    return out


def sum_rows(x: torch.Tensor) -> torch.Tensor:
    """x.sum(0) for a 2-D tensor: the library's kernel where it applies, torch's otherwise."""
    return colsum(x) if colsum_ok(x) else x.sum(0)


def _cl4(x: torch.Tensor, name: str) -> torch.Tensor:
    """A 4-D fp32 device tensor whose memory is channels-last (N, H, W, C) dense."""
    if x.dim() != 4 or x.dtype != torch.float32 or not x.is_cuda:
        raise NativeError(f"{name}: expected a 4-D fp32 device tensor")
    return x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)


def conv4s2_ok(cs: int, cb: int, mode: str) -> bool:
    """Whether the implicit-product kernels take a layer with `cs` channels on its small-resolution side and `cb` on
    its big-resolution side (mode: "down", "up" or "wgrad")."""
    if cs % 4 or cb % 4:
        return False
    return {"down": cb % 32 == 0, "up": cs % 32 == 0, "wgrad": True}[mode]


def conv4s2_down(big: torch.Tensor, kernel: torch.Tensor, bf16: bool = False) -> torch.Tensor:
    """conv2d(big, kernel, stride 2, padding 1) for kernel (cs, cb, 4, 4); channels-last in and out."""
    big, kernel = _cl4(big, "big"), _cl4(kernel, "kernel")
    ni, cb, hb, wb = big.shape
    cs = kernel.shape[0]
    if tuple(kernel.shape) != (cs, cb, 4, 4) or hb % 2 or wb % 2:
        raise NativeError(f"conv4s2_down: kernel {tuple(kernel.shape)} does not fit an input of shape {tuple(big.shape)} (even height and width)")
    out = torch.empty((ni, cs, hb // 2, wb // 2), dtype=torch.float32, device=big.device, memory_format=torch.channels_last)
    check(lib().drm_conv4s2_down_prec(big.data_ptr(), kernel.data_ptr(), out.data_ptr(), ni, hb // 2, wb // 2, cs, cb, int(bool(bf16)),
                                      stream_ptr(big.device)), "drm_conv4s2_down")
    return out


def conv4s2_up(small: torch.Tensor, kernel: torch.Tensor, bf16: bool = False) -> torch.Tensor:
    """conv_transpose2d(small, kernel, stride 2, padding 1) for kernel (cs, cb, 4, 4); channels-last in and out."""
    small, kernel = _cl4(small, "small"), _cl4(kernel, "kernel")
    ni, cs, hs, ws = small.shape
    cb = kernel.shape[1]
    if tuple(kernel.shape) != (cs, cb, 4, 4):
        raise NativeError(f"conv4s2_up: kernel {tuple(kernel.shape)} does not fit an input of shape {tuple(small.shape)}")
    out = torch.empty((ni, cb, 2 * hs, 2 * ws), dtype=torch.float32, device=small.device, memory_format=torch.channels_last)
    check(lib().drm_conv4s2_up_prec(small.data_ptr(), kernel.data_ptr(), out.data_ptr(), ni, hs, ws, cs, cb, int(bool(bf16)),
                                    stream_ptr(small.device)), "drm_conv4s2_up")
    return out


def conv4s2_wgrad(small: torch.Tensor, big: torch.Tensor, bf16: bool = False) -> torch.Tensor:
    """Kernel gradient (cs, cb, 4, 4) (channels-last memory) of a stride-2 4x4 layer from its small- and big-resolution
    tensors: for a convolution (big -> small) pass (grad_output, input), for a transposed one (input, grad_output)."""
    small, big = _cl4(small, "small"), _cl4(big, "big")
    ni, cs, hs, ws = small.shape
    cb = big.shape[1]
    if tuple(big.shape) != (ni, cb, 2 * hs, 2 * ws):
        raise NativeError(f"conv4s2_wgrad: shapes {tuple(small.shape)} and {tuple(big.shape)} are not the two sides of one stride-2 layer")
    out = torch.empty((cs, cb, 4, 4), dtype=torch.float32, device=small.device, memory_format=torch.channels_last)
    need = lib().drm_conv4s2_wgrad_workspace_bytes(ni, hs, ws, cs, cb)
    ws_t = None
    if need:
        held = _gemm_ws.setdefault(small.device, [])
        if not held or held[-1].numel() * 4 < need:
            held.append(torch.empty(need // 4 + 4, dtype=torch.float32, device=small.device))
        ws_t = held[-1]
    check(lib().drm_conv4s2_wgrad_prec(small.data_ptr(), big.data_ptr(), out.data_ptr(), ni, hs, ws, cs, cb, ptr(ws_t),
                                       0 if ws_t is None else ws_t.numel() * 4, int(bool(bf16)), stream_ptr(small.device)),
          "drm_conv4s2_wgrad")
    return out


PERCENTILE_MAX = 16384


def percentile_scale(x, p05, p95, decay: float, limit: float, training: bool) -> torch.Tensor:
    """S = max(EMA(q95) - EMA(q05), limit) with the EMA buffers updated in place when training; one launch."""
    require_cuda(x, p05, p95)
    _dense(x, torch.float32, "x")
    out = torch.empty((), dtype=torch.float32, device=x.device)
    check(lib().drm_percentile_scale(x.data_ptr(), x.numel(), 0.05, 0.95, float(decay), float(limit), int(training),
                                     p05.data_ptr(), p95.data_ptr(), out.data_ptr(), stream_ptr(x.device)), "drm_percentile_scale")
    return out


PROF_CLASSES = 12


def prof_enable(on: bool) -> None:
    check(lib().drm_prof_enable(1 if on else 0), "drm_prof_enable")


def prof_summary():
    """{kernel class name: dict(ms, launches, flops, bytes)} for the launches recorded since prof_enable(True)."""
    out = {}
    for cls in range(PROF_CLASSES):
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        check(lib().drm_prof_summary(cls, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "drm_prof_summary")
        if n.value:
            out[lib().drm_prof_class_name(cls).decode()] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
    return out
