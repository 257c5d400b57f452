"""Thin tensor-level wrappers over the C ABI (include/asr_hip.h).

Every function takes contiguous float32 ROCm tensors, checks shapes on the host (a kernel that
indexes out of bounds can take the whole node down), launches on torch's current stream and
returns device tensors.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import AsrError, call, ptr, stream_ptr

f32 = torch.float32


def _dev(t):
    return t.device


def to_device(a, dtype=f32, device=None):
    """Host array -> device tensor on the CURRENT stream.  Small parameter arrays (transform vectors, step sizes) go
    through pinned memory with a non-blocking copy: a pageable copy makes the host wait for everything already queued
    on the stream (e.g. a whole forward pass), which would serialise the lanes of the pipelined hot path."""
    device = device or _lib.require_gpu()
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    t = torch.as_tensor(np.ascontiguousarray(a), dtype=dtype)
    if t.numel() * t.element_size() <= (1 << 20):
        return t.pin_memory().to(device, non_blocking=True)      # the caching host allocator keeps the staging buffer alive
    return t.to(device)


# ---------------------------------------------------------------------------------------------
# warps
# ---------------------------------------------------------------------------------------------
def warp_affine(src, transforms, out_hw=None, n=None, interpolation="bilinear"):
    """src [N,H,W,C] or [H,W,C] (shared), transforms [N,8] or [8] (shared) -> [N,Ho,Wo,C]; interpolation
    "bilinear" or "nearest" (ImageProjectiveTransformV3, zero fill)."""
    if interpolation not in ("bilinear", "nearest"):
        raise AsrError("warp_affine: interpolation must be 'bilinear' or 'nearest'")
    src_b = src.dim() == 4
    tf_b = transforms.dim() == 2
    if src_b:
        n_src, h, w, c = src.shape
    else:
        h, w, c = src.shape
        n_src = None
    n_tf = transforms.shape[0] if tf_b else None
    n = n or n_src or n_tf or 1
    if (n_src is not None and n_src != n) or (n_tf is not None and n_tf != n):
        raise AsrError(f"warp_affine: batch mismatch src={n_src} transforms={n_tf} n={n}")
    if transforms.shape[-1] != 8:
        raise AsrError("warp_affine: transforms must have 8 coefficients")
    ho, wo = out_hw or (h, w)
    dst = torch.empty((n, ho, wo, c), dtype=f32, device=src.device)
    call("asr_warp_affine_f32" if interpolation == "bilinear" else "asr_warp_affine_nearest_f32", ptr(src), ptr(dst),
         ptr(transforms), n, int(src_b), int(tf_b), h, w, ho, wo, c, stream_ptr())
    return dst


def augment_copies(image, rot_tf, trans_tf, out=None):
    """image [H,W,C] -> [N,H,W,C] = translate(rotate(tile(image))); out: an [N,H,W,C] tensor to write into (e.g. the input
    buffer of a forward plan)."""
    h, w, c = image.shape
    n = rot_tf.shape[0]
    if rot_tf.shape != (n, 8) or trans_tf.shape != (n, 8):
        raise AsrError("augment_copies: transforms must be [N,8]")
    if out is None:
        out = torch.empty((n, h, w, c), dtype=f32, device=image.device)
    elif tuple(out.shape) != (n, h, w, c):
        raise AsrError(f"augment_copies: out must be [{n},{h},{w},{c}], got {tuple(out.shape)}")
    call("asr_augment_copies_f32", ptr(image), ptr(out), ptr(rot_tf), ptr(trans_tf), n, h, w, c, stream_ptr())
    return out


# ---------------------------------------------------------------------------------------------
# SR
# ---------------------------------------------------------------------------------------------
def _sr_dims(x, y):
    if x.dim() != 3 or y.dim() != 4 or x.shape[0] != y.shape[0]:
        raise AsrError(f"SR tensors: x [B,H,W] and y [B,N,h,w] expected, got {tuple(x.shape)} {tuple(y.shape)}")
    b, H, W = x.shape
    _, n, h, w = y.shape
    return b, n, H, W, h, w


def _check_tf(t, b, n, name):
    if tuple(t.shape) != (b, n, 8):
        raise AsrError(f"{name} must be [{b},{n},8], got {tuple(t.shape)}")


def sr_init_target(y, out_hw):
    b, n, h, w = y.shape
    x = torch.empty((b, out_hw[0], out_hw[1]), dtype=f32, device=y.device)
    call("asr_sr_init_target_f32", ptr(y), ptr(x), b, n, out_hw[0], out_hw[1], h, w, stream_ptr())
    return x


def sr_forward_residual(x, y, rot_tf, trans_tf):
    b, n, H, W, h, w = _sr_dims(x, y)
    _check_tf(rot_tf, b, n, "rot_tf")
    _check_tf(trans_tf, b, n, "trans_tf")
    resid = torch.empty_like(y)
    call("asr_sr_forward_residual_f32", ptr(x), ptr(y), ptr(rot_tf), ptr(trans_tf), ptr(resid), b, n, H, W, h, w,
         stream_ptr())
    return resid


def sr_config(optimizer=_lib.OPT_ADAM, flag=False, c0=0.0, c1=0.0, c2=0.0, use_btv=False, btv_alpha=0.6, btv_shift=2,
              plane_chunk=0):
    """asr_sr_config for the *_cfg entry points (meaning of c0..c2 per optimizer: include/asr_hip.h).  plane_chunk: copies
    whose gradient planes the solver keeps alive at once (0 = library default: all of them up to 1 GiB of planes; results
    do not depend on it)."""
    return _lib.SrConfig(int(optimizer), int(bool(flag)), float(c0), float(c1), float(c2),
                         _lib.PRIOR_BTV if use_btv else _lib.PRIOR_TV, float(btv_alpha), int(btv_shift), int(plane_chunk))


def _adam_config(one_minus_beta1, one_minus_beta2, epsilon, amsgrad):
    return sr_config(_lib.OPT_ADAM, amsgrad, one_minus_beta1, one_minus_beta2, epsilon)


def sr_backward(x, resid, inv_rot_tf, inv_trans_tf, lambdas, cfg, state=None, want_grad=False):
    """One step with the update rule / prior of cfg.  state = dict(m, v, vhat, alphas[B]) (slots the optimizer
    does not use may be None) or None for the gradient only.  Returns (x_new | None, grad | None)."""
    b, n, H, W, h, w = _sr_dims(x, resid)
    _check_tf(inv_rot_tf, b, n, "inv_rot_tf")
    _check_tf(inv_trans_tf, b, n, "inv_trans_tf")
    grad = torch.empty_like(x) if (want_grad or state is None) else None
    x_new = torch.empty_like(x) if state is not None else None
    st = state or {}
    if state is not None:
        for k in ("m", "v", "vhat"):
            if st.get(k) is not None and st[k].shape != x.shape:
                raise AsrError(f"state['{k}'] shape mismatch")
        if st["alphas"].numel() != b:
            raise AsrError("state['alphas'] must hold one value per image")
    call("asr_sr_backward_cfg_f32", ptr(x), ptr(x_new, allow_none=True), ptr(resid), ptr(inv_rot_tf),
         ptr(inv_trans_tf), ptr(st.get("m"), allow_none=True), ptr(st.get("v"), allow_none=True),
         ptr(st.get("vhat"), allow_none=True), ptr(st.get("alphas"), allow_none=True), ptr(grad, allow_none=True),
         b, n, H, W, h, w, float(lambdas[0]), float(lambdas[1]), float(lambdas[2]), float(lambdas[3]), C.byref(cfg),
         stream_ptr())
    return x_new, grad


def sr_backward_adam(x, resid, inv_rot_tf, inv_trans_tf, lambdas, adam=None, want_grad=False):
    """One Adam / AMSGrad step with the TV prior.  adam = dict(m, v, vhat, alphas[B], one_minus_beta1,
    one_minus_beta2, epsilon, amsgrad) or None for gradient only.  Returns (x_new | None, grad | None)."""
    b, n, H, W, h, w = _sr_dims(x, resid)
    _check_tf(inv_rot_tf, b, n, "inv_rot_tf")
    _check_tf(inv_trans_tf, b, n, "inv_trans_tf")
    grad = torch.empty_like(x) if (want_grad or adam is None) else None
    x_new = torch.empty_like(x) if adam is not None else None
    if adam is not None:
        for k in ("m", "v"):
            if adam[k].shape != x.shape:
                raise AsrError(f"adam['{k}'] shape mismatch")
        if adam["alphas"].numel() != b:
            raise AsrError("adam['alphas'] must hold one value per image")
    a = adam or {}
    call("asr_sr_backward_adam_f32", ptr(x), ptr(x_new, allow_none=True), ptr(resid), ptr(inv_rot_tf),
         ptr(inv_trans_tf), ptr(a.get("m"), allow_none=True), ptr(a.get("v"), allow_none=True),
         ptr(a.get("vhat"), allow_none=True), ptr(a.get("alphas"), allow_none=True), ptr(grad, allow_none=True),
         b, n, H, W, h, w, float(lambdas[0]), float(lambdas[1]), float(lambdas[2]), float(lambdas[3]),
         float(a.get("one_minus_beta1", 0.0)), float(a.get("one_minus_beta2", 0.0)), float(a.get("epsilon", 0.0)),
         int(bool(a.get("amsgrad", False))), stream_ptr())
    return x_new, grad


def sr_loss_terms(x, resid, cfg=None):
    """[B,4] float64 {sum resid^2, TV or bilateral TV (cfg), sum x^2, sum |x|}."""
    b, n, H, W, h, w = _sr_dims(x, resid)
    terms = torch.empty((b, 4), dtype=torch.float64, device=x.device)
    if cfg is None:
        call("asr_sr_loss_terms_f64", ptr(x), ptr(resid), ptr(terms, torch.float64), b, n, H, W, h, w, stream_ptr())
    else:
        call("asr_sr_loss_terms_cfg_f64", ptr(x), ptr(resid), ptr(terms, torch.float64), b, n, H, W, h, w, C.byref(cfg),
             stream_ptr())
    return terms


def sr_solve(x, y, rot_tf, trans_tf, inv_rot_tf, inv_trans_tf, alphas, lambdas, one_minus_beta1=None, one_minus_beta2=None,
             epsilon=None, amsgrad=False, want_loss=True, cfg=None, slot_init=None, state=None):
    """Runs alphas.shape[0] iterations in place on x.  alphas [num_iter, B] (device).  Either the Adam
    hyper-parameters (asr_sr_solve_f32) or cfg (+ slot_init = {"m"/"v"/"vhat": initial value}) for
    asr_sr_solve_cfg_f32.  state: a dict that carries the optimiser slots and the workspace from one call to the next --
    a solve cut into several calls (the verbose loss print-outs) then performs exactly the updates of a single call."""
    b, n, H, W, h, w = _sr_dims(x, y)
    for t, name in ((rot_tf, "rot_tf"), (trans_tf, "trans_tf"), (inv_rot_tf, "inv_rot_tf"), (inv_trans_tf, "inv_trans_tf")):
        _check_tf(t, b, n, name)
    if alphas.dim() != 2 or alphas.shape[1] != b:
        raise AsrError(f"alphas must be [num_iter,{b}]")
    num_iter = alphas.shape[0]
    lib = _lib.load()
    ws_bytes = (lib.asr_sr_solve_workspace_bytes(b, n, H, W, h, w) if cfg is None else
                lib.asr_sr_solve_workspace_bytes_cfg(b, n, H, W, h, w, C.byref(cfg)))
    if state is not None and "ws" in state:
        ws, m, v, vhat = state["ws"], state["m"], state["v"], state["vhat"]
        if m.shape != x.shape:
            raise AsrError(f"sr_solve: state holds optimiser slots of shape {tuple(m.shape)}, x is {tuple(x.shape)}")
        if ws.numel() * 4 < ws_bytes:       # another plane_chunk / copy count than the call that sized it
            ws = state["ws"] = torch.empty((ws_bytes + 3) // 4, dtype=f32, device=x.device)
    else:
        ws = torch.empty((ws_bytes + 3) // 4, dtype=f32, device=x.device)
        init = slot_init or {}
        m = torch.full_like(x, float(init.get("m", 0.0)))
        v = torch.full_like(x, float(init.get("v", 0.0)))
        vhat = torch.full_like(x, float(init.get("vhat", 0.0)))
        if state is not None:
            state.update(ws=ws, m=m, v=v, vhat=vhat)
    terms = torch.zeros((b, 4), dtype=torch.float64, device=x.device) if want_loss else None
    if cfg is None:
        call("asr_sr_solve_f32", ptr(x), ptr(y), ptr(rot_tf), ptr(trans_tf), ptr(inv_rot_tf), ptr(inv_trans_tf), ptr(m),
             ptr(v), ptr(vhat), ptr(alphas), num_iter, ptr(terms, torch.float64, allow_none=True), ptr(ws),
             C.c_size_t(ws_bytes), b, n, H, W, h, w, float(lambdas[0]), float(lambdas[1]), float(lambdas[2]),
             float(lambdas[3]), float(one_minus_beta1), float(one_minus_beta2), float(epsilon), int(bool(amsgrad)),
             stream_ptr())
    else:
        call("asr_sr_solve_cfg_f32", ptr(x), ptr(y), ptr(rot_tf), ptr(trans_tf), ptr(inv_rot_tf), ptr(inv_trans_tf), ptr(m),
             ptr(v), ptr(vhat), ptr(alphas), num_iter, ptr(terms, torch.float64, allow_none=True), ptr(ws),
             C.c_size_t(ws_bytes), b, n, H, W, h, w, float(lambdas[0]), float(lambdas[1]), float(lambdas[2]),
             float(lambdas[3]), C.byref(cfg), stream_ptr())
    return x, terms


def class_counts(truth, pred, segments=1):
    """int32 label tensors -> int64 [segments, 3, 256]: per label |truth|, |pred|, |truth & pred| (Mean_IOU)."""
    per = truth.numel() // segments
    if per * segments != truth.numel() or pred.numel() != truth.numel():
        raise AsrError("class_counts: truth / pred sizes do not split into equal segments")
    out = torch.empty((segments, 3, 256), dtype=torch.int64, device=truth.device)
    call("asr_class_counts_i32", ptr(truth, torch.int32), ptr(pred, torch.int32), ptr(out, torch.int64), per, segments,
         stream_ptr())
    return out


def realign(y, trans_tf, rot_tf, out_hw, mode):
    """mode "max" | "mean" -> [B,H,W]; "both" -> (max, mean) from one pass over the copies."""
    if y.dim() != 4:
        raise AsrError("realign: y must be [B,N,h,w]")
    b, n, h, w = y.shape
    _check_tf(trans_tf, b, n, "trans_tf")
    _check_tf(rot_tf, b, n, "rot_tf")
    out = torch.empty((b, out_hw[0], out_hw[1]), dtype=f32, device=y.device)
    if mode == "both":
        out_mean = torch.empty_like(out)
        call("asr_realign_max_mean_f32", ptr(y), ptr(out), ptr(out_mean), ptr(trans_tf), ptr(rot_tf), b, n, out_hw[0],
             out_hw[1], h, w, stream_ptr())
        return out, out_mean
    fn = {"max": "asr_realign_max_f32", "mean": "asr_realign_mean_f32"}[mode]
    call(fn, ptr(y), ptr(out), ptr(trans_tf), ptr(rot_tf), b, n, out_hw[0], out_hw[1], h, w, stream_ptr())
    return out


# ---------------------------------------------------------------------------------------------
# OPM / threshold / IoU
# ---------------------------------------------------------------------------------------------
def minmax(x, segments=1):
    per = x.numel() // segments
    if per * segments != x.numel() or per == 0:
        raise AsrError("minmax: tensor does not split into equal non-empty segments")
    out = torch.empty((segments, 2), dtype=f32, device=x.device)
    call("asr_minmax_f32", ptr(x), ptr(out), per, segments, stream_ptr())
    return out


def minmax_normalize(x, segments=1, new_min=0.0, new_max=1.0):
    """min_max_normalization of each of ``segments`` equal parts of x with its own global extrema (device, one call)."""
    per = x.numel() // segments
    if per * segments != x.numel() or per == 0:
        raise AsrError("minmax_normalize: tensor does not split into equal non-empty segments")
    out = torch.empty_like(x)
    ws = torch.empty((segments, 2), dtype=f32, device=x.device)
    call("asr_minmax_normalize_f32", ptr(x), ptr(out), ptr(ws), per, segments, float(new_min), float(new_max), stream_ptr())
    return out


def standard_mask(logits0, out_hw, class_id, out=None):
    """logits0 [h,w,C] of the un-augmented image -> int32 mask [H,W] in {0, class_id}: bilinear upsample + argmax + class
    filter in one kernel (generate_standard_output.py:52-65)."""
    h, w, c = logits0.shape
    if out is None:
        out = torch.empty(tuple(out_hw), dtype=torch.int32, device=logits0.device)
    call("asr_standard_mask_i32", ptr(logits0), ptr(out, torch.int32), h, w, c, int(out_hw[0]), int(out_hw[1]), int(class_id),
         stream_ptr())
    return out


def class_activation(logits, kind):
    """softmax / sigmoid over the last (class) axis, in a new tensor."""
    classes = logits.shape[-1]
    out = torch.empty_like(logits)
    call("asr_class_activation_f32", ptr(logits), ptr(out), logits.numel() // classes, classes,
         {"softmax": 1, "sigmoid": 2}[kind], stream_ptr())
    return out


def argmax(logits):
    classes = logits.shape[-1]
    pixels = logits.numel() // classes
    out = torch.empty(logits.shape[:-1], dtype=torch.int32, device=logits.device)
    call("asr_argmax_i32", ptr(logits), ptr(out, torch.int32), pixels, classes, stream_ptr())
    return out


def _opm_out(out, logits, name):
    """out: optional contiguous [N,h,w] float32 destination (a slice of a per-image stack), else a new tensor."""
    if out is None:
        return torch.empty(logits.shape[:-1], dtype=f32, device=logits.device)
    if tuple(out.shape) != tuple(logits.shape[:-1]) or not out.is_contiguous():
        raise AsrError(f"{name}: out must be a contiguous {tuple(logits.shape[:-1])} tensor, got {tuple(out.shape)}")
    return out


def opm_argmax(logits, class_id, out=None):
    classes = logits.shape[-1]
    pixels = logits.numel() // classes
    out = _opm_out(out, logits, "opm_argmax")
    call("asr_opm_argmax_f32", ptr(logits), ptr(out), pixels, classes, class_id, stream_ptr())
    return out


def opm_slice_max(logits, class_id, out=None, out_max=None):
    classes = logits.shape[-1]
    pixels = logits.numel() // classes
    cls = _opm_out(out, logits, "opm_slice_max")
    mx = _opm_out(out_max, logits, "opm_slice_max")
    call("asr_opm_slice_max_f32", ptr(logits), ptr(cls), ptr(mx), pixels, classes, class_id, stream_ptr())
    return cls, mx


def opm_slice(logits, class_id, new_min=0.0, new_max=1.0, out=None):
    """logits [N,h,w,C]: per-copy global min/max normalisation of the class slice."""
    n = logits.shape[0]
    classes = logits.shape[-1]
    per_copy = logits.numel() // (n * classes)
    out = _opm_out(out, logits, "opm_slice")
    ws = torch.empty((n, 2), dtype=f32, device=logits.device)
    call("asr_opm_slice_f32", ptr(logits), ptr(out), ptr(ws), n, per_copy, classes, class_id, float(new_min),
         float(new_max), stream_ptr())
    return out


def threshold(image, th_value, th_factor=0.15, th_mask=None, segments=1, out=None):
    per = image.numel() // segments
    if out is None:
        out = torch.empty(image.shape, dtype=torch.int32, device=image.device)
    elif out.numel() != image.numel():
        raise AsrError("threshold: out size mismatch")
    ws = torch.empty((segments, 2), dtype=f32, device=image.device)
    if th_mask is not None and th_mask.shape != image.shape:
        raise AsrError("threshold: th_mask shape mismatch")
    call("asr_threshold_f32", ptr(image), ptr(th_mask, allow_none=True), ptr(ws), ptr(out, torch.int32), per, segments,
         float(np.float32(th_factor)), int(th_value), stream_ptr())
    return out


def iou_counts_shared_truth(truth, preds, class_id, include_bg=False):
    """preds [K, ...] int32 masks against ONE int32 label map of the same pixel count -> int64 [K, 4]."""
    k = preds.shape[0]
    per = preds.numel() // k
    if truth.numel() != per:
        raise AsrError("iou_counts_shared_truth: size mismatch")
    counts = torch.empty((k, 4), dtype=torch.int64, device=truth.device)
    call("asr_iou_counts_shared_truth_i32", ptr(truth, torch.int32), ptr(preds, torch.int32), ptr(counts, torch.int64), per, k,
         int(class_id), int(bool(include_bg)), stream_ptr())
    return counts


def iou_counts(truth, pred, class_id, include_bg=False, segments=1):
    if truth.numel() != pred.numel():
        raise AsrError("iou_counts: size mismatch")
    per = truth.numel() // segments
    counts = torch.empty((segments, 4), dtype=torch.int64, device=truth.device)
    call("asr_iou_counts_i32", ptr(truth, torch.int32), ptr(pred, torch.int32), ptr(counts, torch.int64), per, segments,
         int(class_id), int(bool(include_bg)), stream_ptr())
    return counts


# ---------------------------------------------------------------------------------------------
# model layers
# ---------------------------------------------------------------------------------------------
def pack_pw_weights(w_kn):
    k, n = w_kn.shape
    lib = _lib.load()
    out = torch.empty(lib.asr_pwconv_packed_floats(k, n), dtype=f32, device=w_kn.device)
    call("asr_pwconv_pack_weights_f32", ptr(w_kn), ptr(out), k, n, stream_ptr())
    return out


def pack_pw_weights_f16x3(w_kn):
    k, n = w_kn.shape
    lib = _lib.load()
    out = torch.empty(lib.asr_pwconv_packed_floats_f16x3(k, n), dtype=f32, device=w_kn.device)
    call("asr_pwconv_pack_weights_f16x3", ptr(w_kn), ptr(out), k, n, stream_ptr())
    return out


def pwconv(x, w_packed, bias, k, n, out=None, residual=None, relu=False, ldx=None, ldy=None, ldres=None, m=None,
           sub_stride=1, h_in=0, w_in=0, f16x3=False):
    """Rows of x ([..., ldx] with the first k columns used) times packed W [k,n].  f16x3: w_packed comes from
    pack_pw_weights_f16x3 and the split-f16 kernel is used."""
    ldx = ldx or x.shape[-1]
    m = m if m is not None else x.numel() // ldx
    if out is None:
        out = torch.empty((m, n), dtype=f32, device=x.device)
        ldy = n
    ldy = ldy or out.shape[-1]
    ldres = ldres or (residual.shape[-1] if residual is not None else 0)
    call("asr_pwconv_mfma_f16x3" if f16x3 else "asr_pwconv_mfma_f32", ptr(x), ptr(w_packed), ptr(bias, allow_none=True),
         ptr(residual, allow_none=True),
         ptr(out), m, k, n, ldx, ldy, ldres, int(relu), sub_stride, h_in, w_in, stream_ptr())
    return out


def conv3x3_mfma(x, w_packed, bias, cout, stride=1, pad=1, dil=1, relu=False, f16x3=False):
    """Dense 3x3 as an implicit GEMM; w_packed from pack_pw_weights (f32 MFMA) or pack_pw_weights_f16x3 (f16x3=True)
    on the [9 * cin, cout] matrix."""
    b, h, w, cin = x.shape
    ho = (h + 2 * pad - (2 * dil + 1)) // stride + 1
    wo = (w + 2 * pad - (2 * dil + 1)) // stride + 1
    y = torch.empty((b, ho, wo, cout), dtype=f32, device=x.device)
    call("asr_conv3x3_mfma_f16x3" if f16x3 else "asr_conv3x3_mfma_f32", ptr(x), ptr(w_packed), ptr(bias, allow_none=True), ptr(y), b, h, w, cin, cout, stride,
         pad, dil, ho, wo, cin, cout, int(relu), stream_ptr())
    return y


def conv3x3_direct(x, w_hwio, bias, stride, pad_top, pad_left, out_hw, relu=False, f16x3=False):
    """Dense 3x3 for tiny cin (the stem).  f16x3=True: the split-f16 MFMA form (cin = 3, cout = 32 only)."""
    b, h, w, cin = x.shape
    cout = w_hwio.shape[-1]
    y = torch.empty((b, out_hw[0], out_hw[1], cout), dtype=f32, device=x.device)
    call("asr_conv3x3_stem_f16x3" if f16x3 else "asr_conv3x3_direct_f32", ptr(x), ptr(w_hwio), ptr(bias), ptr(y), b, h, w, cin,
         cout, stride, pad_top, pad_left, out_hw[0], out_hw[1], cin, cout, int(relu), stream_ptr())
    return y


def entry_stem_fused(x, w1_hwio, b1, w2_packed16, b2):
    """relu(conv3x3(relu(conv3x3_s2(x, w1) + b1), w2) + b2), 3 -> 32 -> 64 channels, in one kernel (even input sizes);
    w2_packed16 = pack_pw_weights_f16x3 of the [288, 64] matrix."""
    b, h, w, c = x.shape
    if c != 3 or tuple(w1_hwio.shape) != (3, 3, 3, 32):
        raise AsrError("entry_stem_fused: x [B,H,W,3] and w1 [3,3,3,32] expected")
    y = torch.empty((b, h // 2, w // 2, 64), dtype=f32, device=x.device)
    call("asr_entry_stem_f16x3", ptr(x), ptr(w1_hwio), ptr(b1), ptr(w2_packed16), ptr(b2), ptr(y), b, h, w, 3, 64, stream_ptr())
    return y


def sepconv_fused(x, w_33c, bias_dw, w_packed16, bias_pw, cout, pre_relu=False, dw_relu=False, out_relu=False):
    """A whole separable conv (depthwise 3x3 stride 1 + pointwise) in one kernel: cin in {64, 128} -> 128."""
    b, h, w, c = x.shape
    y = torch.empty((b, h, w, cout), dtype=f32, device=x.device)
    call("asr_sepconv_fused_f16x3", ptr(x), ptr(w_33c), ptr(bias_dw), ptr(w_packed16), ptr(bias_pw), ptr(y), b, h, w, c, cout, c,
         cout, int(pre_relu), int(dw_relu), int(out_relu), stream_ptr())
    return y


def dwconv3x3_split(x, w_33c, bias, stride=1, rate=1, pre_relu=False, post_relu=0):
    """Depthwise 3x3 ('same', or the explicit symmetric pad of the stride-2 sepconvs) whose output is written as
    split-f16 chunks for pwconv_presplit.  Returns (buffer [B*Ho*Wo, chunks, 32] float32-typed storage, (B, Ho, Wo), chunks)."""
    b, h, w, c = x.shape
    pad = rate
    ho, wo = (h, w) if stride == 1 else ((h + 2 * pad - (2 * rate + 1)) // stride + 1, (w + 2 * pad - (2 * rate + 1)) // stride + 1)
    chunks = (c + 31) // 32
    y = torch.empty((b * ho * wo, chunks, 32), dtype=f32, device=x.device)
    call("asr_dwconv3x3_nhwc_split_f16", ptr(x), ptr(w_33c), ptr(bias), ptr(y), b, h, w, c, stride, rate, pad, pad, ho, wo, c,
         chunks, int(pre_relu), int(post_relu), stream_ptr())
    return y, (b, ho, wo), chunks


def pwconv_presplit(x_split, w_packed16, bias, k, n, chunks, out=None, residual=None, relu=0):
    """Pointwise conv on a split-f16 operand (dwconv3x3_split); w_packed16 from pack_pw_weights_f16x3."""
    m = x_split.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=f32, device=x_split.device)
    call("asr_pwconv_mfma_f16x3_presplit", ptr(x_split), ptr(w_packed16), ptr(bias, allow_none=True),
         ptr(residual, allow_none=True), ptr(out), m, k, n, chunks, out.shape[-1], residual.shape[-1] if residual is not None else 0,
         int(relu), stream_ptr())
    return out


def aspp_dwconv3(x, w3, bias3, rates=(6, 12, 18), pre_relu=False, post_relu=True):
    """Fused three-rate ASPP depthwise: x [B,H,W,C], w3 [3,3,3,C], bias3 [3,C] -> three [B,H,W,C]."""
    b, h, w, c = x.shape
    outs = [torch.empty((b, h, w, c), dtype=f32, device=x.device) for _ in range(3)]
    call("asr_aspp_dwconv3_nhwc_f32", ptr(x), ptr(w3), ptr(bias3), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), b, h, w, c,
         int(rates[0]), int(rates[1]), int(rates[2]), c, c, int(pre_relu), int(post_relu), stream_ptr())
    return outs


def aspp_dwconv3_split(x, w3, bias3, rates=(6, 12, 18), pre_relu=False, post_relu=True):
    """aspp_dwconv3 with its three outputs as split-f16 GEMM operands (c % 32 == 0): three buffers
    [B*H*W, c / 32, 32] of float32-typed storage for pwconv_presplit."""
    b, h, w, c = x.shape
    if c % 32:
        raise AsrError("aspp_dwconv3_split: channels must be a multiple of 32")
    outs = [torch.empty((b * h * w, c // 32, 32), dtype=f32, device=x.device) for _ in range(3)]
    call("asr_aspp_dwconv3_nhwc_split_f16", ptr(x), ptr(w3), ptr(bias3), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), b, h, w, c,
         int(rates[0]), int(rates[1]), int(rates[2]), c, c // 32, int(pre_relu), int(post_relu), stream_ptr())
    return outs


def dwconv3x3(x, w_33c, bias, stride=1, rate=1, pad_top=None, pad_left=None, out_hw=None, pre_relu=False,
              post_relu=False, force_direct=0, out=None, ldy=None):
    """force_direct is the kernel mode: 0 auto, 1 direct, 2 streaming register window."""
    b, h, w, c = x.shape
    if pad_top is None:
        pad_top = pad_left = rate            # stride-1 'same'
    if out_hw is None:
        out_hw = (h, w)
    if out is None:
        out = torch.empty((b, out_hw[0], out_hw[1], c), dtype=f32, device=x.device)
    ldy = ldy or out.shape[-1]
    call("asr_dwconv3x3_nhwc_f32", ptr(x), ptr(w_33c), ptr(bias), ptr(out), b, h, w, c, stride, rate, pad_top, pad_left,
         out_hw[0], out_hw[1], c, ldy, int(pre_relu), int(post_relu), int(force_direct), stream_ptr())
    return out


def gap(x):
    b, h, w, c = x.shape
    y = torch.empty((b, c), dtype=f32, device=x.device)
    call("asr_gap_f32", ptr(x), ptr(y), b, h * w, c, c, stream_ptr())
    return y


def resize_bilinear(x, out_hw, out=None, ldy=None):
    b, h, w, c = x.shape
    if out is None:
        out = torch.empty((b, out_hw[0], out_hw[1], c), dtype=f32, device=x.device)
    ldy = ldy or out.shape[-1]
    call("asr_resize_bilinear_f32", ptr(x), ptr(out), b, h, w, c, out_hw[0], out_hw[1], c, ldy, stream_ptr())
    return out
