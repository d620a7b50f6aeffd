"""ctypes binding of libssdhip.so (C ABI: include/ssdhip.h) + torch-side buffer plumbing.

PyTorch is used for device memory and streams only: tensors' `data_ptr()` and the current
HIP stream cross the boundary as plain pointers.  There is NO fallback: if the library
is missing, or a tensor is not on a GPU, these functions raise.
"""
from __future__ import annotations

import ctypes
import os
import threading

import numpy as np

# SSDHIP_LIB lets tools/ load the instrumented build of the same sources (tools/prof_build.sh); never a CPU fallback.
_LIB_PATH = os.environ.get("SSDHIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libssdhip.so")
_lib = None
_lock = threading.Lock()

F32, F64 = 0, 1
COORDS = {"centroids": 0, "corners": 1, "minmax": 2}
BORDER = {"half": 0, "include": 1, "exclude": 2}
SEM_NUMPY, SEM_KERAS, SEM_DEBUG = 0, 1, 2
ABI_VERSION = 1


class SsdHipError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def load():
    """Load libssdhip.so (once).  Raises if it has not been built -- there is no CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise SsdHipError("libssdhip.so not found at %s: build it with `python -m ssd_keras_amd.build` "
                              "(hipcc, gfx950).  ssd_keras_amd has no CPU fallback." % _LIB_PATH)
        lib = ctypes.CDLL(_LIB_PATH)
        c_int, c_dbl, c_vp, c_sz, c_flt = ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_float
        lib.ssdhip_abi_version.restype = c_int
        lib.ssdhip_abi_version.argtypes = []
        lib.ssdhip_strerror.restype = ctypes.c_char_p
        lib.ssdhip_strerror.argtypes = [c_int]
        lib.ssdhip_decode_workspace_bytes.restype = c_sz
        lib.ssdhip_decode_workspace_bytes.argtypes = [c_int] * 7
        lib.ssdhip_decode_detections.restype = c_int
        lib.ssdhip_decode_detections.argtypes = ([c_vp, c_int, c_int, c_int, c_int, c_dbl, c_dbl, c_int, c_int, c_int, c_int,
                                                  c_int, c_int, c_dbl, c_dbl, c_int,
                                                  c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_sz, c_vp])
        lib.ssdhip_decode_stages.restype = c_int
        lib.ssdhip_decode_stages.argtypes = [c_int] + lib.ssdhip_decode_detections.argtypes
        if hasattr(lib, "ssdhip_encode"):
            lib.ssdhip_encode_workspace_bytes.restype = c_sz
            lib.ssdhip_encode_workspace_bytes.argtypes = [c_int] * 4
            lib.ssdhip_encode.restype = c_int
            lib.ssdhip_encode.argtypes = ([c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_dbl, c_dbl,
                                           c_int, c_dbl, c_dbl, c_int, c_int, c_int, c_int,
                                           c_vp, c_vp, c_vp, c_vp, c_sz, c_vp])
        if hasattr(lib, "ssdhip_loss_forward"):
            lib.ssdhip_loss_workspace_bytes.restype = c_sz
            lib.ssdhip_loss_workspace_bytes.argtypes = [c_int] * 3
            lib.ssdhip_loss_forward.restype = c_int
            lib.ssdhip_loss_forward.argtypes = [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_flt,
                                                c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]
            lib.ssdhip_loss_backward.restype = c_int
            lib.ssdhip_loss_backward.argtypes = [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_flt, c_vp, c_vp]
        if lib.ssdhip_abi_version() != ABI_VERSION:
            raise SsdHipError("libssdhip.so ABI %d != expected %d" % (lib.ssdhip_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise SsdHipError("%s failed: %s (rc=%d)" % (what, load().ssdhip_strerror(rc).decode(), rc))


def _torch():
    import torch
    return torch


def require_cuda(t, name):
    if not t.is_cuda:
        raise SsdHipError("%s must live on a GPU (got device %s): ssd_keras_amd has no CPU path" % (name, t.device))
    if not t.is_contiguous():
        raise SsdHipError("%s must be contiguous" % name)


def current_stream_ptr(device):
    torch = _torch()
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _Workspaces:
    """One growable scratch buffer per (device, purpose, stream).  Kernels on one stream serialise, so a buffer is reused across
    calls on that stream; work enqueued on another stream gets its own buffer (two streams decoding at once must not share
    candidate lists), allocated while that stream is current so the caching allocator orders its reuse on the right stream."""

    def __init__(self):
        self._bufs = {}

    def get(self, device, purpose, nbytes, zeroed=False):
        """`zeroed`: the buffer is zero-filled when it is (re)allocated -- for kernels that keep counters in it and leave them zero."""
        torch = _torch()
        key = (str(device), purpose, int(torch.cuda.current_stream(device).cuda_stream))
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = (torch.zeros if zeroed else torch.empty)(int(nbytes), dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        return buf


workspaces = _Workspaces()


def decode(y_pred, conf_thresh, iou_thresh, top_k, nms_cap, class_agnostic, semantics, coords, normalize_coords,
           img_height, img_width, border_pixels, out_dtype, out_rows, want_anchor_idx=False, workspace=None,
           stages=7, outputs=None):
    """Enqueue ssdhip_decode_detections on the current stream.
    y_pred: CUDA float32 or float64 (B, N, C+12) -- float64 predictions take the reference's all-float64 flow
    (csrc/ssdhip_decode64.hip).  Returns (out (B,out_rows,6), count (B,) int32, anchor_idx or None)."""
    torch = _torch()
    lib = load()
    require_cuda(y_pred, "y_pred")
    if y_pred.dtype not in (torch.float32, torch.float64):
        raise SsdHipError("y_pred must be float32 or float64")
    in_dt = F32 if y_pred.dtype == torch.float32 else F64
    if in_dt == F64 and semantics == SEM_KERAS:
        raise SsdHipError("the DecodeDetections layers are float32 graphs: float64 predictions are not accepted there")
    B, N, L = y_pred.shape
    C = L - 12
    dev = y_pred.device
    k = int(top_k) if top_k else 0
    cap = int(nms_cap) if nms_cap else 0
    need = lib.ssdhip_decode_workspace_bytes(B, N, C, k, cap, int(bool(class_agnostic)), in_dt)
    if need == 0:
        raise SsdHipError("unsupported decode shape B=%d N=%d C=%d" % (B, N, C))
    ws = workspace if workspace is not None else workspaces.get(dev, "decode", need)
    if outputs is not None:
        out, count, aidx = outputs
    else:
        out = torch.empty((B, out_rows, 6), dtype=torch.float64 if out_dtype == F64 else torch.float32, device=dev)
        count = torch.empty((B,), dtype=torch.int32, device=dev)
        aidx = torch.empty((B, out_rows), dtype=torch.int32, device=dev) if want_anchor_idx else None
    with torch.cuda.device(dev):
        rc = lib.ssdhip_decode_stages(
            int(stages), ctypes.c_void_p(y_pred.data_ptr()), in_dt, B, N, C, float(conf_thresh), float(iou_thresh), k, cap,
            int(bool(class_agnostic)), int(semantics), COORDS[coords], int(bool(normalize_coords)),
            float(img_height if img_height is not None else 1.0), float(img_width if img_width is not None else 1.0),
            BORDER[border_pixels], ctypes.c_void_p(out.data_ptr()), out_dtype, int(out_rows),
            ctypes.c_void_p(count.data_ptr()), ctypes.c_void_p(aidx.data_ptr()) if aidx is not None else None,
            ctypes.c_void_p(ws.data_ptr()), ws.numel(), current_stream_ptr(dev))
    check(rc, "ssdhip_decode_detections")
    return out, count, aidx


def to_device(a, device=None, dtype=None):
    """NumPy array or torch tensor -> contiguous CUDA tensor (host buffers cross PCIe here)."""
    torch = _torch()
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a))
    if device is None:
        device = a.device if a.is_cuda else torch.device("cuda", torch.cuda.current_device())
    if dtype is not None and a.dtype != dtype:
        a = a.to(dtype)
    return a.to(device, non_blocking=False).contiguous()


# ------------------------------------------------------------------------------------------------
# graph glue (csrc/ssdhip_layers.hip): bf16 NHWC activations
# ------------------------------------------------------------------------------------------------
def _bind_layers(lib):
    if getattr(lib, "_layers_bound", False):
        return
    c_int, c_vp, c_ll = ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong
    lib.ssdhip_bias_act_nhwc_bf16.restype = c_int
    lib.ssdhip_bias_act_nhwc_bf16.argtypes = [c_vp, c_vp, c_vp, c_ll, c_int, c_int, c_vp]
    lib.ssdhip_bias_act_maxpool_nhwc_bf16.restype = c_int
    lib.ssdhip_bias_act_maxpool_nhwc_bf16.argtypes = [c_vp, c_vp, c_vp] + [c_int] * 10 + [c_vp]
    lib.ssdhip_l2_normalize_nhwc_bf16.restype = c_int
    lib.ssdhip_l2_normalize_nhwc_bf16.argtypes = [c_vp, c_vp, c_vp, c_ll, c_int, c_vp]
    lib.ssdhip_preprocess_nhwc_f32_to_bf16.restype = c_int
    lib.ssdhip_preprocess_nhwc_f32_to_bf16.argtypes = [c_vp, c_vp, c_ll, c_int, c_vp, c_vp, c_vp, c_vp]
    lib.ssdhip_assemble_predictions_bf16.restype = c_int
    lib.ssdhip_assemble_predictions_bf16.argtypes = [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp]
    lib.ssdhip_assemble_predictions_strided_bf16.restype = c_int
    lib.ssdhip_assemble_predictions_strided_bf16.argtypes = [c_int] + [c_vp] * 9 + [c_int, c_int, c_int, c_vp, c_vp]
    lib._layers_bound = True


def _layers_lib():
    lib = load()
    _bind_layers(lib)
    return lib


def _nhwc_bf16(t, name):
    """(B, C, H, W) bf16 CUDA tensor whose memory is NHWC; returns it (made so if needed) and (B, H, W, C)."""
    torch = _torch()
    if not t.is_cuda or t.dtype != torch.bfloat16 or t.dim() != 4:
        raise SsdHipError("%s must be a 4-D bfloat16 CUDA tensor" % name)
    if not t.permute(0, 2, 3, 1).is_contiguous():
        t = t.contiguous(memory_format=torch.channels_last)
        if not t.permute(0, 2, 3, 1).is_contiguous():          # size-1 dims can leave odd strides behind
            t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    b, c, h, w = t.shape
    if c % 8:
        raise SsdHipError("%s: channel count %d is not a multiple of 8" % (name, c))
    return t, (b, h, w, c)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def bias_act(x, bias, relu=True, inplace=True):
    """act(x + bias[c]) on an NHWC-memory bf16 feature map (B, C, H, W); in place by default."""
    torch = _torch()
    lib = _layers_lib()
    x, (b, h, w, c) = _nhwc_bf16(x, "x")
    y = x if inplace else torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_bias_act_nhwc_bf16(_ptr(x), _ptr(bias), _ptr(y), b * h * w, c, int(bool(relu)), current_stream_ptr(x.device))
    check(rc, "ssdhip_bias_act_nhwc_bf16")
    return y


def pool_out_size(n, k, s, p, ceil_mode):
    if ceil_mode:
        o = -(-(n + 2 * p - k) // s) + 1
        if (o - 1) * s >= n + p:
            o -= 1
        return o
    return (n + 2 * p - k) // s + 1


def bias_act_maxpool(x, bias, kernel, stride, pad=0, ceil_mode=False, relu=True):
    """max_pool2d(act(x + bias)) in one pass; x (B, C, H, W) bf16 with NHWC memory -> (B, C, Ho, Wo) likewise."""
    torch = _torch()
    lib = _layers_lib()
    x, (b, h, w, c) = _nhwc_bf16(x, "x")
    ho, wo = pool_out_size(h, kernel, stride, pad, ceil_mode), pool_out_size(w, kernel, stride, pad, ceil_mode)
    y = torch.empty((b, ho, wo, c), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_bias_act_maxpool_nhwc_bf16(_ptr(x), _ptr(bias), _ptr(y), b, h, w, c, int(kernel), int(stride), int(pad),
                                                   ho, wo, int(bool(relu)), current_stream_ptr(x.device))
    check(rc, "ssdhip_bias_act_maxpool_nhwc_bf16")
    return y


def l2_normalize(x, gamma):
    torch = _torch()
    lib = _layers_lib()
    x, (b, h, w, c) = _nhwc_bf16(x, "x")
    g = gamma.detach().float().contiguous()
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_l2_normalize_nhwc_bf16(_ptr(x), _ptr(g), _ptr(y), b * h * w, c, current_stream_ptr(x.device))
    check(rc, "ssdhip_l2_normalize_nhwc_bf16")
    return y


def pool2_l2_normalize(x, gamma):
    """MaxPooling2D(2, 2, 'same') AND L2Normalization of the same (B, 512, H, W) bf16 channels_last map in one pass
    (ssdhip_pool2_l2_normalize_nhwc_bf16): returns (pooled (B, 512, ceil(H/2), ceil(W/2)), normalised (B, 512, H, W))."""
    torch = _torch()
    lib = _layers_lib()
    if not getattr(lib, "_p2l2_bound", False):
        lib.ssdhip_pool2_l2_normalize_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_pool2_l2_normalize_nhwc_bf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
        lib._p2l2_bound = True
    x, (b, h, w, c) = _nhwc_bf16(x, "x")
    g = gamma.detach().float().contiguous()
    pooled = torch.empty((b, (h + 1) // 2, (w + 1) // 2, c), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    normed = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_pool2_l2_normalize_nhwc_bf16(_ptr(x), _ptr(g), _ptr(pooled), _ptr(normed), b, h, w, c, current_stream_ptr(x.device))
    check(rc, "ssdhip_pool2_l2_normalize_nhwc_bf16")
    return pooled, normed


def _l2_bind(lib):
    if not getattr(lib, "_l2_bound", False):
        c_int, c_vp, c_ll = ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong
        lib.ssdhip_l2_normalize_bwd_waves.restype = c_int
        lib.ssdhip_l2_normalize_bwd_waves.argtypes = [c_ll, c_int, c_int]
        lib.ssdhip_l2_normalize_fwd.restype = c_int
        lib.ssdhip_l2_normalize_fwd.argtypes = [c_vp, c_vp, c_vp, c_vp, c_ll, c_int, c_int, c_vp]
        lib.ssdhip_l2_normalize_bwd.restype = c_int
        lib.ssdhip_l2_normalize_bwd.argtypes = [c_vp] * 6 + [c_int, c_ll, c_int, c_int, c_vp]
        lib._l2_bound = True
    return lib


def _l2_view(t, name):
    """(B, C, H, W) float32 or bf16 CUDA tensor in channels_last memory -> (tensor, n_pixels, C, is_bf16)."""
    torch = _torch()
    if not t.is_cuda or t.dim() != 4 or t.dtype not in (torch.float32, torch.bfloat16):
        raise SsdHipError("%s must be a 4-D float32 / bfloat16 CUDA tensor" % name)
    if not t.permute(0, 2, 3, 1).is_contiguous():
        t = t.contiguous(memory_format=torch.channels_last)
        if not t.permute(0, 2, 3, 1).is_contiguous():
            t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    b, c, h, w = t.shape
    return t, b * h * w, c, int(t.dtype == torch.bfloat16)


def l2_normalize_supported(x):
    torch = _torch()
    if not x.is_cuda or x.dim() != 4 or x.dtype not in (torch.float32, torch.bfloat16):
        return False
    b, c, h, w = x.shape
    is_bf16 = int(x.dtype == torch.bfloat16)
    return _l2_bind(load()).ssdhip_l2_normalize_bwd_waves(b * h * w, c, is_bf16) > 0


def l2_normalize_fwd(x, gamma, want_inv=True):
    """L2Normalization forward for float32 / bf16 maps (ssdhip_l2_normalize_fwd): returns (y, inv_norm | None)."""
    torch = _torch()
    lib = _l2_bind(load())
    x, n_px, c, is_bf16 = _l2_view(x, "x")
    g = gamma.detach().float().contiguous()
    y = torch.empty_like(x)
    inv = torch.empty((n_px,), dtype=torch.float32, device=x.device) if want_inv else None
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_l2_normalize_fwd(_ptr(x), _ptr(g), _ptr(y), _ptr(inv), n_px, c, is_bf16, current_stream_ptr(x.device))
    check(rc, "ssdhip_l2_normalize_fwd")
    return y, inv


def l2_normalize_bwd(x, dy, gamma, inv):
    """Gradients of L2Normalization (ssdhip_l2_normalize_bwd): returns (dx like x, dgamma float32 (C,))."""
    torch = _torch()
    lib = _l2_bind(load())
    x, n_px, c, is_bf16 = _l2_view(x, "x")
    dy, _, _, _ = _l2_view(dy.to(x.dtype), "dy")
    g = gamma.detach().float().contiguous()
    n_waves = lib.ssdhip_l2_normalize_bwd_waves(n_px, c, is_bf16)
    if n_waves <= 0:
        raise SsdHipError("ssdhip_l2_normalize_bwd: unsupported shape")
    dx = torch.empty_like(x)
    part = torch.empty((n_waves, c), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_l2_normalize_bwd(_ptr(x), _ptr(dy), _ptr(g), _ptr(inv), _ptr(dx), _ptr(part), n_waves, n_px, c, is_bf16,
                                         current_stream_ptr(x.device))
    check(rc, "ssdhip_l2_normalize_bwd")
    return dx, row_sums(part)


def preprocess(images, mean=None, divide=None, swap=None):
    """(B, H, W, C<=4) float32 CUDA images -> (B, C, H, W) bf16 with NHWC memory: (img[..., swap] - mean[swap]) / divide[swap]."""
    torch = _torch()
    lib = _layers_lib()
    require_cuda(images, "images")
    if images.dtype != torch.float32 or images.dim() != 4 or images.shape[3] > 4:
        raise SsdHipError("images must be float32 (B, H, W, C<=4)")
    b, h, w, c = images.shape
    out = torch.empty((b, h, w, c), dtype=torch.bfloat16, device=images.device)
    def fa(v):
        if v is None:
            return None
        v = [float(v)] * c if np.isscalar(v) else [float(t) for t in v]
        if len(v) != c:
            raise SsdHipError("per-channel constant has %d entries for %d channels" % (len(v), c))
        return (ctypes.c_float * c)(*v)
    ia = (ctypes.c_int * c)(*[int(t) for t in swap]) if swap else None
    with torch.cuda.device(images.device):
        rc = lib.ssdhip_preprocess_nhwc_f32_to_bf16(_ptr(images), _ptr(out), b * h * w, c, fa(mean), fa(divide), ia,
                                                    current_stream_ptr(images.device))
    check(rc, "ssdhip_preprocess_nhwc_f32_to_bf16")
    return out.permute(0, 3, 1, 2)


def _head_sources(confs, locs, conf_biases, loc_biases, n_boxes, anchors_var, n_classes):
    """Validate the per-layer head outputs and pack them for the C ABI.  Layer i is given either as two dense head outputs
    confs[i] (B, n_boxes*C, h, w), locs[i] (B, n_boxes*4, h, w), or -- locs[i] is None -- as ONE wider output confs[i]
    (B, >= n_boxes*(C+4), h, w) whose channels are [conf | loc | padding].  Returns (ctypes argument tuple, keep-alive list, B, N)."""
    torch = _torch()
    nl = len(confs)
    keep = []
    cp, lp, cbp, lbp, na, cs, ls = [], [], [], [], [], [], []
    nhwc = lambda t: t if t.permute(0, 2, 3, 1).is_contiguous() else t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    f32 = confs[0].dtype == torch.float32                    # the reference-precision path: float32 packed heads, bias already added
    esz = 4 if f32 else 2
    for i in range(nl):
        cf = nhwc(confs[i])
        if cf.dtype != (torch.float32 if f32 else torch.bfloat16) or not cf.is_cuda:
            raise SsdHipError("the predictor head outputs must be CUDA tensors, all bfloat16 or all float32")
        if f32 and (locs[i] is not None or conf_biases[i] is not None or loc_biases[i] is not None):
            raise SsdHipError("float32 head outputs are packed [conf | loc | padding] maps that carry their bias already")
        b, ch, h, w = cf.shape
        nc, nloc = n_boxes[i] * n_classes, n_boxes[i] * 4
        if locs[i] is None:                                  # packed heads
            if ch < nc + nloc:
                raise SsdHipError("packed head %d has %d channels, expected >= %d" % (i, ch, nc + nloc))
            keep.append(cf)
            cp.append(cf.data_ptr()); lp.append(cf.data_ptr() + esz * nc); cs.append(ch); ls.append(ch)
        else:
            lc = nhwc(locs[i])
            if lc.dtype != torch.bfloat16 or ch != nc or lc.shape[1] != nloc:
                raise SsdHipError("head %d has %d / %d channels, expected %d / %d" % (i, ch, lc.shape[1], nc, nloc))
            keep += [cf, lc]
            cp.append(cf.data_ptr()); lp.append(lc.data_ptr()); cs.append(nc); ls.append(nloc)
        na.append(h * w * n_boxes[i])
        cbp.append(conf_biases[i].data_ptr() if conf_biases[i] is not None else 0)
        lbp.append(loc_biases[i].data_ptr() if loc_biases[i] is not None else 0)
    B = confs[0].shape[0]
    N = int(sum(na))
    if anchors_var.shape != (N, 8) or anchors_var.dtype != torch.float32:
        raise SsdHipError("anchors_var must be float32 (%d, 8)" % N)
    arr = lambda v: (ctypes.c_void_p * nl)(*v)
    iarr = lambda v: (ctypes.c_int * nl)(*[int(t) for t in v])
    if f32:
        args = (nl, arr(cp), arr(lp), iarr(na), iarr(n_boxes), iarr(cs), iarr(ls), _ptr(anchors_var))
    else:
        args = (nl, arr(cp), arr(lp), arr(cbp), arr(lbp), iarr(na), iarr(n_boxes), iarr(cs), iarr(ls), _ptr(anchors_var))
    return args, keep, int(B), N


ASSEMBLE_BACKWARD_MAX_LDS = 160 * 1024 - 64          # include/ssdhip.h: SSDHIP_ASSEMBLE_BACKWARD_MAX_LDS


def assemble_backward_supported(n_classes, n_boxes, strides):
    """Whether `assemble_predictions_backward` runs for source maps with these boxes per pixel and packed channel strides: the kernel's
    own LDS formula (ssdhip_assemble_backward_lds_bytes), so the model's gate and the launch cannot disagree."""
    lib = _layers_lib()
    if not hasattr(lib, "ssdhip_assemble_backward_lds_bytes"):
        return False
    lib.ssdhip_assemble_backward_lds_bytes.restype = ctypes.c_size_t
    lib.ssdhip_assemble_backward_lds_bytes.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    nl = len(n_boxes)
    nbx = (ctypes.c_int * nl)(*[int(v) for v in n_boxes])
    st = (ctypes.c_int * nl)(*[int(v) for v in strides])
    need = int(lib.ssdhip_assemble_backward_lds_bytes(nl, nbx, st, int(n_classes)))
    return 0 < need <= ASSEMBLE_BACKWARD_MAX_LDS


def assemble_predictions_backward(grad_pred, y_pred, packed_shapes, n_boxes, n_classes):
    """Backward of `assemble_predictions` for PACKED bf16 heads (the training step): grad_pred, y_pred (B, N, C+12) float32 -> one
    (B, Cp, h, w) bf16 gradient with NHWC memory per source map, channels [conf | loc | zero padding] (csrc/ssdhip_layers.hip,
    head_grad_kernel: softmax backward, pass-through of the offsets, one rounding).  packed_shapes: the (B, Cp, h, w) of every map."""
    torch = _torch()
    lib = _layers_lib()
    if not getattr(lib, "_apb_bound", False):
        c_int, c_vp = ctypes.c_int, ctypes.c_void_p
        lib.ssdhip_assemble_predictions_backward_bf16.restype = c_int
        lib.ssdhip_assemble_predictions_backward_bf16.argtypes = [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]
        lib._apb_bound = True
    require_cuda(grad_pred, "grad_pred")
    if grad_pred.dtype != torch.float32 or y_pred.dtype != torch.float32 or grad_pred.shape != y_pred.shape or grad_pred.dim() != 3:
        raise SsdHipError("grad_pred and y_pred must be float32 (B, N, C+12) tensors of one shape")
    grad_pred, y_pred = grad_pred.contiguous(), y_pred.contiguous()
    B, N, L = y_pred.shape
    if L != n_classes + 12:
        raise SsdHipError("y_pred has %d columns, expected %d" % (L, n_classes + 12))
    nl = len(packed_shapes)
    outs = [torch.empty((b, h, w, cp), dtype=torch.bfloat16, device=y_pred.device) for (b, cp, h, w) in packed_shapes]
    ptrs = (ctypes.c_void_p * nl)(*[o.data_ptr() for o in outs])
    na = (ctypes.c_int * nl)(*[int(h * w * nb) for (b, cp, h, w), nb in zip(packed_shapes, n_boxes)])
    nbx = (ctypes.c_int * nl)(*[int(v) for v in n_boxes])
    st = (ctypes.c_int * nl)(*[int(cp) for (b, cp, h, w) in packed_shapes])
    with torch.cuda.device(y_pred.device):
        rc = lib.ssdhip_assemble_predictions_backward_bf16(nl, ptrs, na, nbx, st, _ptr(y_pred), _ptr(grad_pred), B, N, int(n_classes),
                                                           current_stream_ptr(y_pred.device))
    check(rc, "ssdhip_assemble_predictions_backward_bf16")
    return [o.permute(0, 3, 1, 2) for o in outs]


def assemble_predictions(confs, locs, conf_biases, loc_biases, n_boxes, anchors_var, n_classes):
    """Per-layer NHWC conv outputs -> y_pred (B, N, C+12) float32 in one pass (softmax, biases, anchors, concatenation);
    see `_head_sources` for the accepted layer formats."""
    torch = _torch()
    lib = _layers_lib()
    args, keep, B, N = _head_sources(confs, locs, conf_biases, loc_biases, n_boxes, anchors_var, n_classes)
    y = torch.empty((B, N, n_classes + 12), dtype=torch.float32, device=confs[0].device)
    if confs[0].dtype == torch.float32:
        if not getattr(lib, "_apf32_bound", False):
            c_int, c_vp = ctypes.c_int, ctypes.c_void_p
            lib.ssdhip_assemble_predictions_strided_f32.restype = c_int
            lib.ssdhip_assemble_predictions_strided_f32.argtypes = [c_int] + [c_vp] * 7 + [c_int, c_int, c_int, c_vp, c_vp]
            lib._apf32_bound = True
        with torch.cuda.device(y.device):
            rc = lib.ssdhip_assemble_predictions_strided_f32(*args, B, N, int(n_classes), _ptr(y), current_stream_ptr(y.device))
        check(rc, "ssdhip_assemble_predictions_strided_f32")
        return y
    with torch.cuda.device(y.device):
        rc = lib.ssdhip_assemble_predictions_strided_bf16(*args, B, N, int(n_classes), _ptr(y), current_stream_ptr(y.device))
    check(rc, "ssdhip_assemble_predictions_strided_bf16")
    return y


def decode_from_heads(confs, locs, conf_biases, loc_biases, n_boxes, anchors_var, n_classes, conf_thresh, iou_thresh, top_k,
                      nms_cap, class_agnostic, semantics, coords, normalize_coords, img_height, img_width, border_pixels, out_dtype,
                      out_rows, want_anchor_idx=False):
    """DecodeDetections straight from the predictor heads' outputs (no y_pred in HBM): `ssdhip_decode_from_heads`.
    Returns (out (B,out_rows,6), count (B,) int32, anchor_idx or None) exactly as `decode` does for the assembled tensor."""
    torch = _torch()
    lib = _layers_lib()
    if not getattr(lib, "_dfh_bound", False):
        c_int, c_vp, c_dbl, c_sz = ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_size_t
        lib.ssdhip_decode_from_heads.restype = c_int
        lib.ssdhip_decode_from_heads.argtypes = ([c_int] + [c_vp] * 9 + [c_int, c_int, c_int, c_dbl, c_dbl, c_int, c_int, c_int, c_int,
                                                                       c_int, c_int, c_dbl, c_dbl, c_int, c_vp, c_int, c_int, c_vp, c_vp,
                                                                       c_vp, c_sz, c_vp])
        lib.ssdhip_decode_from_heads_f32.restype = c_int
        lib.ssdhip_decode_from_heads_f32.argtypes = ([c_int] + [c_vp] * 7 + [c_int, c_int, c_int, c_dbl, c_dbl, c_int, c_int, c_int, c_int,
                                                                           c_int, c_int, c_dbl, c_dbl, c_int, c_vp, c_int, c_int, c_vp, c_vp,
                                                                           c_vp, c_sz, c_vp])
        lib._dfh_bound = True
    args, keep, B, N = _head_sources(confs, locs, conf_biases, loc_biases, n_boxes, anchors_var, n_classes)
    fn = lib.ssdhip_decode_from_heads_f32 if confs[0].dtype == torch.float32 else lib.ssdhip_decode_from_heads
    dev = confs[0].device
    k = int(top_k) if top_k else 0
    cap = int(nms_cap) if nms_cap else 0
    need = lib.ssdhip_decode_workspace_bytes(B, N, int(n_classes), k, cap, int(bool(class_agnostic)), F32)
    if need == 0:
        raise SsdHipError("unsupported decode shape B=%d N=%d C=%d" % (B, N, n_classes))
    ws = workspaces.get(dev, "decode", need)
    out = torch.empty((B, out_rows, 6), dtype=torch.float64 if out_dtype == F64 else torch.float32, device=dev)
    count = torch.empty((B,), dtype=torch.int32, device=dev)
    aidx = torch.empty((B, out_rows), dtype=torch.int32, device=dev) if want_anchor_idx else None
    with torch.cuda.device(dev):
        rc = fn(*args, B, N, int(n_classes), float(conf_thresh), float(iou_thresh), k, cap,
                                          int(bool(class_agnostic)), int(semantics), COORDS[coords], int(bool(normalize_coords)),
                                          float(img_height if img_height is not None else 1.0),
                                          float(img_width if img_width is not None else 1.0), BORDER[border_pixels], _ptr(out),
                                          out_dtype, int(out_rows), _ptr(count), _ptr(aidx), _ptr(ws), ws.numel(),
                                          current_stream_ptr(dev))
    check(rc, "ssdhip_decode_from_heads")
    return out, count, aidx


def _train_lib():
    lib = load()
    if not getattr(lib, "_train_bound", False):
        c_int, c_vp, c_ll = ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong
        lib.ssdhip_relu_bwd_bias_blocks.restype = c_int
        lib.ssdhip_relu_bwd_bias_blocks.argtypes = [c_ll, c_int]
        lib.ssdhip_relu_bwd_bias_nhwc_bf16.restype = c_int
        lib.ssdhip_relu_bwd_bias_nhwc_bf16.argtypes = [c_vp, c_vp, c_vp, c_vp, c_ll, c_int, c_int, c_vp]
        lib.ssdhip_maxpool_bwd_nhwc_bf16.restype = c_int
        lib.ssdhip_maxpool_bwd_nhwc_bf16.argtypes = [c_vp, c_vp, c_vp] + [c_int] * 9 + [c_vp]
        lib.ssdhip_maxpool2_relu_bwd_bias_nhwc_bf16.restype = c_int
        lib.ssdhip_maxpool2_relu_bwd_bias_nhwc_bf16.argtypes = [c_vp, c_vp, c_vp, c_vp] + [c_int] * 5 + [c_vp]
        lib.ssdhip_channel_sums_nhwc_bf16.restype = c_int
        lib.ssdhip_channel_sums_nhwc_bf16.argtypes = [c_vp, c_vp, c_ll, c_int, c_int, c_vp]
        lib.ssdhip_conv1_1_bwd_blocks.restype = c_int
        lib.ssdhip_conv1_1_bwd_blocks.argtypes = [c_int, c_int, c_int]
        lib.ssdhip_conv1_1_bwd_nhwc_bf16.restype = c_int
        lib.ssdhip_conv1_1_bwd_nhwc_bf16.argtypes = [c_vp] * 5 + [c_int] * 4 + [c_vp]
        lib._train_bound = True
    return lib


def channel_sums_partial(gy):
    """Per-workgroup channel sums of a bf16 NHWC map, float32 [n_blocks, C] (the bias gradient of a layer without activation is their
    sum over axis 0: conv3x3_wgrad(..., bias_partial=...) adds them in its reduction launch); None: channel count not supported."""
    torch = _torch()
    lib = _train_lib()
    gy, (b, h, w, c) = _nhwc_bf16(gy, "gy")
    nb = lib.ssdhip_relu_bwd_bias_blocks(b * h * w, c)
    if nb == 0:
        return None
    partial = torch.empty((nb, c), dtype=torch.float32, device=gy.device)
    with torch.cuda.device(gy.device):
        check(lib.ssdhip_channel_sums_nhwc_bf16(_ptr(gy), _ptr(partial), b * h * w, c, nb, current_stream_ptr(gy.device)), "ssdhip_channel_sums_nhwc_bf16")
    return partial


def relu_bwd_bias(gy, y, reduce=True):
    """Backward of `y = relu(conv + bias)` up to the convolution: returns (gy masked by y > 0, bias gradient float32 [C]) in one
    pass (csrc/ssdhip_train.hip), or None when the channel count is not supported.  gy, y: (B, C, H, W) bf16 with NHWC memory.
    reduce=False: the second element is the [n_blocks, C] per-workgroup partial sums (the bias gradient is their sum over axis 0)."""
    torch = _torch()
    lib = _train_lib()
    gy, (b, h, w, c) = _nhwc_bf16(gy, "gy")
    y, _ = _nhwc_bf16(y, "y")
    nb = lib.ssdhip_relu_bwd_bias_blocks(b * h * w, c)
    if nb == 0:
        return None
    out = torch.empty_like(gy)
    partial = torch.empty((nb, c), dtype=torch.float32, device=gy.device)
    with torch.cuda.device(gy.device):
        rc = lib.ssdhip_relu_bwd_bias_nhwc_bf16(_ptr(gy), _ptr(y), _ptr(out), _ptr(partial), b * h * w, c, nb, current_stream_ptr(gy.device))
    check(rc, "ssdhip_relu_bwd_bias_nhwc_bf16")
    return out, (row_sums(partial) if reduce else partial)


def conv1_1_backward(gy, y, x):
    """Backward of the FIRST layer, `y = relu(conv3x3(x) + bias)` with 3 input and 64 output channels and no data gradient, in one pass
    (csrc/ssdhip_train.hip, conv1_1_bwd_kernel): returns (dL/dW float32 (64, 3, 3, 3), dL/db float32 (64,)).  gy, y: (B, 64, H, W) bf16
    with NHWC memory (the gradient of the post-ReLU output, that output), x: (B, 3, H, W) bf16 with NHWC memory."""
    torch = _torch()
    lib = _train_lib()
    gy, (b, h, w, c) = _nhwc_bf16(gy, "gy")
    y, shp = _nhwc_bf16(y, "y")
    if not x.is_cuda or x.dtype != torch.bfloat16 or x.dim() != 4:
        raise SsdHipError("x must be a 4-D bfloat16 CUDA tensor")
    if not x.permute(0, 2, 3, 1).is_contiguous():
        x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    bx, cx, hx, wx = x.shape
    if c != 64 or shp != (b, h, w, c) or (bx, hx, wx, cx) != (b, h, w, 3):
        raise SsdHipError("conv1_1_backward needs (B, 64, H, W) gradients / activations and a (B, 3, H, W) input")
    nb = lib.ssdhip_conv1_1_bwd_blocks(b, h, w)
    wpart = torch.empty((nb, 64, 27), dtype=torch.float32, device=gy.device)
    bpart = torch.empty((nb, 64), dtype=torch.float32, device=gy.device)
    with torch.cuda.device(gy.device):
        rc = lib.ssdhip_conv1_1_bwd_nhwc_bf16(_ptr(gy), _ptr(y), _ptr(x), _ptr(wpart), _ptr(bpart), b, h, w, nb, current_stream_ptr(gy.device))
    check(rc, "ssdhip_conv1_1_bwd_nhwc_bf16")
    gw = row_sums(wpart).view(64, 3, 3, 3).permute(0, 3, 1, 2)            # k = (kh 3 + kw) 3 + ci  ->  (co, ci, kh, kw)
    return gw, row_sums(bpart)


def maxpool2_relu_bwd_bias(y, gp, reduce=True):
    """Backward of `p = max_pool2d(y, 2, 2, ceil_mode=True)`, `y = relu(conv + bias)` up to the convolution in ONE pass: returns (the
    full-resolution gradient masked by y > 0, bias gradient float32 [C]), or None when the channel count is not supported.  y (B, C, H,
    W), gp (B, C, ceil(H/2), ceil(W/2)) bf16 with NHWC memory.  Bit-identical to maxpool_bwd followed by relu_bwd_bias."""
    torch = _torch()
    lib = _train_lib()
    y, (b, h, w, c) = _nhwc_bf16(y, "y")
    gp, (b2, ho, wo, c2) = _nhwc_bf16(gp, "gp")
    if (b2, ho, wo, c2) != (b, (h + 1) // 2, (w + 1) // 2, c):
        raise SsdHipError("gp must be the gradient of the 2x2 / stride-2 pooled map of y")
    nb = lib.ssdhip_relu_bwd_bias_blocks(b * h * w, c)
    if nb == 0:
        return None
    out = torch.empty_like(y)
    partial = torch.empty((nb, c), dtype=torch.float32, device=y.device)
    with torch.cuda.device(y.device):
        rc = lib.ssdhip_maxpool2_relu_bwd_bias_nhwc_bf16(_ptr(y), _ptr(gp), _ptr(out), _ptr(partial), b, h, w, c, nb, current_stream_ptr(y.device))
    check(rc, "ssdhip_maxpool2_relu_bwd_bias_nhwc_bf16")
    return out, (row_sums(partial) if reduce else partial)


# ---- the parameter side of the training step (csrc/ssdhip_optim.hip) ------------------------------------------------------------
SHADOW_DESC = [("src", "<u8"), ("cl", "<u8"), ("tr", "<u8"), ("O", "<i4"), ("I", "<i4"), ("KK", "<i4"), ("tr_ostride", "<i4"),
               ("tr_ooff", "<i4"), ("tile0", "<i4"), ("src_channels_last", "<i4"), ("reserved", "<i4")]   # struct ssdhip_shadow_desc (include/ssdhip.h): 56 bytes


def _optim_lib():
    lib = load()
    if not getattr(lib, "_optim_bound", False):
        c_int, c_vp, c_d = ctypes.c_int, ctypes.c_void_p, ctypes.c_double
        lib.ssdhip_shadow_refresh.restype = c_int
        lib.ssdhip_shadow_refresh.argtypes = [c_vp, c_int, c_int, c_int, c_int, c_vp]
        lib.ssdhip_sgd_momentum_step.restype = c_int
        lib.ssdhip_sgd_momentum_step.argtypes = [c_int, c_vp, c_vp, c_vp, c_vp, c_d, c_d, c_d, c_vp]
        lib._optim_bound = True
    return lib


def shadow_table(weights, vectors, device):
    """The device table of ssdhip_shadow_refresh.  `weights`: (master float32 (O, I, kh, kw) contiguous, cl bf16 tensor or None,
    tr bf16 tensor or None, tr_ostride, tr_ooff); `vectors`: (master float32 (n,), bf16 (n,)).  Returns (table tensor, n_weights,
    n_tiles, n_vectors, n_vector_blocks); the caller keeps every tensor alive -- the table holds raw pointers."""
    import numpy as np
    torch = _torch()
    tab = np.zeros((len(weights) + len(vectors),), dtype=np.dtype(SHADOW_DESC))
    tiles = 0
    for k, (w, cl, tr, ostride, ooff) in enumerate(weights):
        o, i, kh, kw = w.shape
        src_cl = 0 if w.is_contiguous() else 1
        if w.dtype != torch.float32 or not (w.is_contiguous() or w.permute(0, 2, 3, 1).is_contiguous()) or kh * kw > 16:
            raise SsdHipError("shadow_table: master filters must be float32 (O, I, kh, kw), contiguous or channels_last, at most 16 taps")
        if cl is not None and (cl.dtype != torch.bfloat16 or tuple(cl.shape) != tuple(w.shape) or not cl.permute(0, 2, 3, 1).is_contiguous()):
            raise SsdHipError("shadow_table: `cl` must be a bf16 tensor of the filters' shape in channels_last memory")
        tab[k] = (w.data_ptr(), cl.data_ptr() if cl is not None else 0, tr.data_ptr() if tr is not None else 0, o, i, kh * kw,
                  ostride, ooff, tiles, src_cl, 0)
        tiles += -(-o // 32) * -(-i // 32)
    vblocks = 0
    for k, (v, dst) in enumerate(vectors):
        if v.dtype != torch.float32 or dst.dtype != torch.bfloat16 or v.dim() != 1 or not v.is_contiguous() or not dst.is_contiguous():
            raise SsdHipError("shadow_table: vectors are contiguous float32 (n,) -> bf16 (n,)")
        tab[len(weights) + k] = (v.data_ptr(), dst.data_ptr(), 0, v.numel(), 0, 0, 0, 0, vblocks, 0, 0)
        vblocks += -(-v.numel() // 256)
    dev_tab = torch.from_numpy(tab.view(np.uint8).copy()).to(device)
    return dev_tab, len(weights), tiles, len(vectors), vblocks


def shadow_refresh(table):
    """bf16 copies (channels_last, and transposed with flipped taps) of every tensor of `table` (shadow_table's result): ONE launch."""
    dev_tab, n_w, n_tiles, n_v, n_vb = table
    lib = _optim_lib()
    with _torch().cuda.device(dev_tab.device):
        check(lib.ssdhip_shadow_refresh(_ptr(dev_tab), n_w, n_tiles, n_v, n_vb, current_stream_ptr(dev_tab.device)), "ssdhip_shadow_refresh")


def sgd_table(params, grads, bufs, device):
    """The HOST table of ssdhip_sgd_momentum_step over dense float32 tensors (parameter, gradient, momentum buffer of one memory
    layout each): ctypes arrays of device pointers and element counts.  It travels in the kernel arguments -- nothing is uploaded."""
    torch = _torch()
    n = len(params)
    for p, g, m in zip(params, grads, bufs):
        for t in (p, g, m):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != p.numel() or t.data_ptr() % 16 or t.device != device:
                raise SsdHipError("sgd_table: parameter, gradient and momentum buffer must be dense float32 of one size on one device")
    vp, ll = ctypes.c_void_p * n, ctypes.c_longlong * n
    return (device, n, vp(*[p.data_ptr() for p in params]), vp(*[g.data_ptr() for g in grads]), vp(*[m.data_ptr() for m in bufs]),
            ll(*[p.numel() for p in params]))


def sgd_momentum_step(table, lr, momentum, weight_decay=0.0):
    device, n, pp, gp, mp, nn = table
    lib = _optim_lib()
    with _torch().cuda.device(device):
        check(lib.ssdhip_sgd_momentum_step(n, pp, gp, mp, nn, float(lr), float(momentum), float(weight_decay), current_stream_ptr(device)),
              "ssdhip_sgd_momentum_step")


def maxpool_bwd(x, gy, kernel, stride, pad=0):
    """Gradient of max_pool2d (windows clipped to the map) with respect to its input: x (B, C, H, W), gy (B, C, Ho, Wo), bf16 NHWC."""
    torch = _torch()
    lib = _train_lib()
    x, (b, h, w, c) = _nhwc_bf16(x, "x")
    gy, (_, ho, wo, _) = _nhwc_bf16(gy, "gy")
    gx = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_maxpool_bwd_nhwc_bf16(_ptr(x), _ptr(gy), _ptr(gx), b, h, w, c, int(kernel), int(stride), int(pad), ho, wo,
                                              current_stream_ptr(x.device))
    check(rc, "ssdhip_maxpool_bwd_nhwc_bf16")
    return gx


def conv2d_same(x, weight, bias, dilation=1, relu=True, variant=None):
    """'same' convolution (kernel 1 or 3, stride 1) + bias + ReLU in ONE libssdhip MFMA kernel (csrc/ssdhip_conv.hip).
    x (B, Cin, H, W) bf16 with NHWC memory; weight (Cout, Cin, k, k) bf16 with channels_last memory; Cin, Cout % 64 == 0."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_conv_bound", False):
        lib.ssdhip_conv2d_same_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv2d_same_nhwc_bf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 8 + [ctypes.c_void_p]
        lib.ssdhip_conv2d_same_nhwc_bf16_variant.restype = ctypes.c_int
        lib.ssdhip_conv2d_same_nhwc_bf16_variant.argtypes = [ctypes.c_int] + lib.ssdhip_conv2d_same_nhwc_bf16.argtypes
        lib._conv_bound = True
    x, (b, h, w, cin) = _nhwc_bf16(x, "x")
    cout, cin_w, kh, kw = weight.shape
    if weight.dtype != torch.bfloat16 or cin_w != cin or kh != kw:
        raise SsdHipError("weight must be bfloat16 (Cout, %d, k, k)" % cin)
    wt = weight if weight.permute(0, 2, 3, 1).is_contiguous() else weight.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    y = torch.empty((b, h, w, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    with torch.cuda.device(x.device):
        args = (_ptr(x), _ptr(wt), _ptr(bias), _ptr(y), b, h, w, cin, cout, int(kh), int(dilation), int(bool(relu)),
                current_stream_ptr(x.device))
        rc = lib.ssdhip_conv2d_same_nhwc_bf16(*args) if variant is None else lib.ssdhip_conv2d_same_nhwc_bf16_variant(int(variant), *args)
    check(rc, "ssdhip_conv2d_same_nhwc_bf16")
    return y


def conv3x3_halo_plan(b, h, w, pool, cout=128):
    """(geometry, position tiles, stacked-batch row pitch, rows of tiles) the slab entries pick for a batch of h x w maps with `cout`
    output channels: geometry 0 = padded position grid, 4 = 16 x 16 pixel tiles, 5 = 8 x 32; pitch 0 = tiles per image
    (csrc/ssdhip_convh.hip, convh_pick_2d / convh_plan_unpooled).  Host arithmetic only: works without a GPU (256 CUs assumed)."""
    lib = load()
    lib.ssdhip_conv3x3_halo_plan.restype = ctypes.c_int
    lib.ssdhip_conv3x3_halo_plan.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int)]
    plan = (ctypes.c_int * 4)()
    check(lib.ssdhip_conv3x3_halo_plan(int(b), int(h), int(w), int(cout), int(bool(pool)), plan), "ssdhip_conv3x3_halo_plan")
    return tuple(plan)


def conv3x3_halo(x, weight, bias, relu=True, pool=False):
    """3x3 'same' convolution + bias + ReLU [+ 2x2 / stride-2 'same' max-pool] through the slab kernel (csrc/ssdhip_convh.hip):
    Cin % 128 == 0, Cout % 128 == 0.  Layouts as conv2d_same; bit-identical to conv2d_same / conv2d_same_pool2."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_halo_bound", False):
        lib.ssdhip_conv3x3_halo_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv3x3_halo_nhwc_bf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
        lib._halo_bound = True
    x, (b, h, w, cin) = _nhwc_bf16(x, "x")
    cout, cin_w, kh, kw = weight.shape
    if weight.dtype != torch.bfloat16 or cin_w != cin or kh != 3 or kw != 3:
        raise SsdHipError("weight must be bfloat16 (Cout, %d, 3, 3)" % cin)
    wt = weight if weight.permute(0, 2, 3, 1).is_contiguous() else weight.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if pool else (h, w)
    y = torch.empty((b, ho, wo, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_conv3x3_halo_nhwc_bf16(_ptr(x), _ptr(wt), _ptr(bias), _ptr(y), b, h, w, cin, cout, int(bool(relu)), int(bool(pool)),
                                               current_stream_ptr(x.device))
    check(rc, "ssdhip_conv3x3_halo_nhwc_bf16")
    return y


def conv3x3_halo_masked(x, weight, mask, sums=False):
    """The 3x3 'same' convolution of x with weight (no bias, no activation) through the slab kernel, zeroed where `mask` (the output's
    shape) is <= 0: a layer's data gradient with the threshold_backward of the ReLU layer below in the epilogue
    (csrc/ssdhip_convh.hip, MSK).  None when the geometry is not the slab kernel's (Cin, Cout % 128).
    sums=True: returns (y, partial) -- partial [rows, Cout] float32 whose column sums are the channel sums of y (the bias gradient of the
    layer below; conv3x3_wgrad(..., bias_partial=partial) adds the rows in its reduction launch)."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_halo_masked_bound", False):
        lib.ssdhip_conv3x3_halo_masked_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv3x3_halo_masked_nhwc_bf16.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
        lib.ssdhip_conv3x3_halo_masked_bias_rows.restype = ctypes.c_int
        lib.ssdhip_conv3x3_halo_masked_bias_rows.argtypes = [ctypes.c_int] * 4
        lib._halo_masked_bound = True
    cout, cin_w, kh, kw = weight.shape
    if (not x.is_cuda or x.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16 or mask.dtype != torch.bfloat16 or kh != 3 or kw != 3
            or cin_w % 128 or cout % 128 or x.shape[1] != cin_w):
        return None
    x, (b, h, w, cin) = _nhwc_bf16(x, "x")
    mask, mshape = _nhwc_bf16(mask, "mask")
    if mshape != (b, h, w, cout):
        raise SsdHipError("mask must have the output's shape %s, got %s" % ((b, h, w, cout), mshape))
    wt = weight if weight.permute(0, 2, 3, 1).is_contiguous() else weight.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    y = torch.empty((b, h, w, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    partial, rows = None, 0
    with torch.cuda.device(x.device):
        if sums:
            rows = lib.ssdhip_conv3x3_halo_masked_bias_rows(b, h, w, cout)
            if rows <= 0:
                return None
            partial = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
        rc = lib.ssdhip_conv3x3_halo_masked_nhwc_bf16(_ptr(x), _ptr(wt), _ptr(mask), _ptr(y), _ptr(partial), rows, b, h, w, cin, cout,
                                                      current_stream_ptr(x.device))
    if rc == -1:                                          # SSDHIP_E_BADARG: not the slab kernel's geometry
        return None
    check(rc, "ssdhip_conv3x3_halo_masked_nhwc_bf16")
    return (y, partial) if sums else y


def conv3x3_image_supported(x, weight, dilation=1):
    """Geometry of conv3x3_image: 3x3 filters, H * W <= 384 pixels, Cin % 64 == 0, Cout % 64 == 0, 1 <= dilation <= 16."""
    b, cin, h, w = x.shape
    cout, cin_w, kh, kw = weight.shape
    return kh == 3 and kw == 3 and cin_w == cin and h * w <= 384 and cin % 64 == 0 and cout % 64 == 0 and 1 <= int(dilation) <= 16


def conv3x3_image(x, weight, bias, dilation=1, relu=True):
    """3x3 'same' convolution with any dilation on a small map, one image per tile (csrc/ssdhip_convimg.hip: fc6).  Layouts as
    conv2d_same; bit-identical to it."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_convimg_bound", False):
        lib.ssdhip_conv3x3_image_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv3x3_image_nhwc_bf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
        lib._convimg_bound = True
    x, (b, h, w, cin) = _nhwc_bf16(x, "x")
    cout, cin_w, kh, kw = weight.shape
    if weight.dtype != torch.bfloat16 or cin_w != cin or kh != 3 or kw != 3:
        raise SsdHipError("weight must be bfloat16 (Cout, %d, 3, 3)" % cin)
    wt = weight if weight.permute(0, 2, 3, 1).is_contiguous() else weight.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    y = torch.empty((b, h, w, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_conv3x3_image_nhwc_bf16(_ptr(x), _ptr(wt), _ptr(bias), _ptr(y), b, h, w, cin, cout, int(dilation), int(bool(relu)),
                                                current_stream_ptr(x.device))
    check(rc, "ssdhip_conv3x3_image_nhwc_bf16")
    return y


def conv2d_image_supported(x, weight, stride=1, padding=0, dilation=1):
    """Geometry of conv2d_image (csrc/ssdhip_convimg.hip, ssdhip_conv2d_image_nhwc_bf16): k x k filters with k in {1, 3}, at most 384
    input and 384 output pixels per image, Cin % 64 == 0, Cout % 64 == 0, 1 <= stride <= 4, 1 <= dilation <= 16,
    0 <= padding <= dilation (k // 2)."""
    b, cin, h, w = x.shape
    cout, cin_w, kh, kw = weight.shape
    k, s, p, d = int(kh), int(stride), int(padding), int(dilation)
    if kh != kw or k not in (1, 3) or cin_w != cin or cin % 64 or cout % 64 or h * w > 384:
        return False
    if not (1 <= s <= 4 and 1 <= d <= 16 and 0 <= p <= d * (k // 2)) or h + 2 * p < d * (k - 1) + 1 or w + 2 * p < d * (k - 1) + 1:
        return False
    ho, wo = (h + 2 * p - d * (k - 1) - 1) // s + 1, (w + 2 * p - d * (k - 1) - 1) // s + 1
    return 1 <= ho * wo <= 384


def conv2d_image(x, weight, bias, stride=1, padding=0, dilation=1, relu=True):
    """k x k convolution (k in {1, 3}) with stride / zero padding / dilation on a small map, one image per tile with its 64-channel
    slices resident in LDS (csrc/ssdhip_convimg.hip; round 6: fc7, conv6_1, conv6_2).  Layouts as conv2d; bit-identical to it."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_convimg2_bound", False):
        lib.ssdhip_conv2d_image_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv2d_image_nhwc_bf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 10 + [ctypes.c_void_p]
        lib._convimg2_bound = True
    x, (b, h, w, cin) = _nhwc_bf16(x, "x")
    cout, cin_w, kh, kw = weight.shape
    if weight.dtype != torch.bfloat16 or cin_w != cin or kh != kw or kh not in (1, 3):
        raise SsdHipError("weight must be bfloat16 (Cout, %d, k, k) with k in (1, 3)" % cin)
    k, s, p, d = int(kh), int(stride), int(padding), int(dilation)
    wt = weight if weight.permute(0, 2, 3, 1).is_contiguous() else weight.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    ho, wo = (h + 2 * p - d * (k - 1) - 1) // s + 1, (w + 2 * p - d * (k - 1) - 1) // s + 1
    if ho < 1 or wo < 1:
        raise SsdHipError("the filter does not fit the padded map")
    y = torch.empty((b, ho, wo, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_conv2d_image_nhwc_bf16(_ptr(x), _ptr(wt), _ptr(bias), _ptr(y), b, h, w, cin, cout, k, s, p, d, int(bool(relu)),
                                               current_stream_ptr(x.device))
    check(rc, "ssdhip_conv2d_image_nhwc_bf16")
    return y


def conv_chain_pack(weight, out=None):
    """[Cout, Cin, k, k] bfloat16 (channels_last) filters in the fragment order `conv_chain` streams; None if the geometry is not supported.
    `out`: an earlier result for the same geometry, re-packed IN PLACE (a captured HIP graph keeps reading that storage)."""
    torch = _torch()
    lib = load()
    _bind_chain(lib)
    cout, cin, kh, kw = weight.shape
    if weight.dtype != torch.bfloat16 or kh != kw:
        raise SsdHipError("weight must be bfloat16 (Cout, Cin, k, k)")
    n = int(lib.ssdhip_conv_chain_packed_bytes(kh, cin, cout))
    if n == 0:
        return None
    wt = weight if weight.permute(0, 2, 3, 1).is_contiguous() else weight.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    if out is not None and (out.numel() != n or out.dtype != torch.uint8 or out.device != weight.device):
        raise SsdHipError("conv_chain_pack: `out` does not match this filter's packed size")
    packed = out if out is not None else torch.empty((n,), dtype=torch.uint8, device=weight.device)
    with torch.cuda.device(weight.device):
        check(lib.ssdhip_conv_chain_pack_weight(_ptr(wt), _ptr(packed), kh, cin, cout, current_stream_ptr(weight.device)), "ssdhip_conv_chain_pack_weight")
    return packed


def _bind_chain(lib):
    if not getattr(lib, "_chain_bound", False):
        vp, ci = ctypes.c_void_p, ctypes.c_int
        lib.ssdhip_conv_chain_packed_bytes.restype = ctypes.c_size_t
        lib.ssdhip_conv_chain_packed_bytes.argtypes = [ci, ci, ci]
        lib.ssdhip_conv_chain_pack_weight.restype = ci
        lib.ssdhip_conv_chain_pack_weight.argtypes = [vp, vp, ci, ci, ci, vp]
        lib.ssdhip_conv_chain_nhwc_bf16.restype = ci
        lib.ssdhip_conv_chain_nhwc_bf16.argtypes = [vp, ci, ci, ci, ci, ci] + [vp] * 8 + [vp]
        lib._chain_bound = True


def conv_chain(x, layers):
    """A chain of small convolutions in one launch (csrc/ssdhip_chain.hip).  x (B, C0, H, W) bfloat16 channels_last; `layers`: a list of
    dicts {packed, bias (bfloat16 or None), k, stride, pad, cout, relu, keep}; returns the list of the kept layers' outputs
    (B, Cout, Ho, Wo) channels_last, or None when the chain does not fit (the caller runs the layers one by one)."""
    torch = _torch()
    lib = load()
    _bind_chain(lib)
    x, (b, h, w, c0) = _nhwc_bf16(x, "x")
    n = len(layers)
    outs, ys = [], []
    hh, ww = h, w
    for l in layers:
        hh = (hh + 2 * l["pad"] - l["k"]) // l["stride"] + 1
        ww = (ww + 2 * l["pad"] - l["k"]) // l["stride"] + 1
        if hh < 1 or ww < 1:
            return None
        if l.get("keep", False):
            y = torch.empty((b, hh, ww, l["cout"]), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
            outs.append(y)
            ys.append(y)
        else:
            ys.append(None)
    parr = lambda ts: (ctypes.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in ts])
    iarr = lambda key: (ctypes.c_int * n)(*[int(l[key]) for l in layers])
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_conv_chain_nhwc_bf16(_ptr(x), b, h, w, c0, n, parr([l["packed"] for l in layers]), parr([l.get("bias") for l in layers]),
                                             parr(ys), iarr("k"), iarr("stride"), iarr("pad"), iarr("cout"), iarr("relu"),
                                             current_stream_ptr(x.device))
    if rc == -1:                                          # SSDHIP_E_BADARG: the chain does not fit this kernel
        return None
    check(rc, "ssdhip_conv_chain_nhwc_bf16")
    return outs


def _bind_chain_x3(lib):
    if not getattr(lib, "_chainx3_bound", False):
        vp, ci = ctypes.c_void_p, ctypes.c_int
        lib.ssdhip_conv_chain_x3_packed_bytes.restype = ctypes.c_size_t
        lib.ssdhip_conv_chain_x3_packed_bytes.argtypes = [ci, ci, ci]
        lib.ssdhip_conv_chain_x3_pack_weight.restype = ci
        lib.ssdhip_conv_chain_x3_pack_weight.argtypes = [vp, vp, ci, ci, ci, vp]
        lib.ssdhip_conv_chain_x3_nhwc_f16.restype = ci
        lib.ssdhip_conv_chain_x3_nhwc_f16.argtypes = [vp, ci, ci, ci, ci, ci] + [vp] * 9 + [vp]
        lib._chainx3_bound = True


def conv_chain_x3_pack(packed_weight, out=None):
    """x3_pack_weight's (Cout, 3 Cin, k, k) float16 channels_last filters in the fragment order `conv_chain_x3` streams; None if the
    geometry is not supported.  `out`: an earlier result for the same geometry, re-packed in place."""
    torch = _torch()
    lib = load()
    _bind_chain_x3(lib)
    cout, c3, kh, kw = packed_weight.shape
    if packed_weight.dtype != torch.float16 or kh != kw or c3 % 3 or not packed_weight.permute(0, 2, 3, 1).is_contiguous():
        raise SsdHipError("conv_chain_x3_pack takes x3_pack_weight's (Cout, 3 Cin, k, k) float16 channels_last filters")
    n = int(lib.ssdhip_conv_chain_x3_packed_bytes(kh, c3 // 3, cout))
    if n == 0:
        return None
    if out is not None and (out.numel() != n or out.dtype != torch.uint8 or out.device != packed_weight.device):
        raise SsdHipError("conv_chain_x3_pack: `out` does not match this filter's packed size")
    packed = out if out is not None else torch.empty((n,), dtype=torch.uint8, device=packed_weight.device)
    with torch.cuda.device(packed_weight.device):
        check(lib.ssdhip_conv_chain_x3_pack_weight(_ptr(packed_weight), _ptr(packed), kh, c3 // 3, cout, current_stream_ptr(packed_weight.device)),
              "ssdhip_conv_chain_x3_pack_weight")
    return packed


def conv_chain_x3(x2, layers):
    """The chain of small convolutions at the reference's precision in one launch (csrc/ssdhip_chain.hip, conv_chain_x3_kernel).  x2
    (B, 2 C0, H, W) float16 channels_last pair map; `layers`: dicts {packed (conv_chain_x3_pack), bias (float32, divided by the layer's
    output divisor, or None), k, stride, pad, cout, relu, mul (oscale * input divisor / output divisor), keep}; returns the kept
    layers' pair maps (B, 2 Cout, Ho, Wo), or None when the chain does not fit (the caller runs the layers one by one)."""
    torch = _torch()
    lib = load()
    _bind_chain_x3(lib)
    if not (x2.is_cuda and x2.dtype == torch.float16 and x2.dim() == 4 and x2.shape[1] % 2 == 0 and _nhwc_ok(x2)):
        raise SsdHipError("conv_chain_x3 takes a float16 (B, 2 C, H, W) channels_last pair map")
    b, c2, h, w = x2.shape
    n = len(layers)
    outs, ys = [], []
    hh, ww = h, w
    for l in layers:
        hh = (hh + 2 * l["pad"] - l["k"]) // l["stride"] + 1
        ww = (ww + 2 * l["pad"] - l["k"]) // l["stride"] + 1
        if hh < 1 or ww < 1:
            return None
        if l.get("bias") is not None and (l["bias"].dtype != torch.float32 or not l["bias"].is_contiguous()):
            raise SsdHipError("conv_chain_x3: bias must be contiguous float32")
        if l.get("keep", False):
            y = torch.empty((b, hh, ww, 2 * l["cout"]), dtype=torch.float16, device=x2.device).permute(0, 3, 1, 2)
            outs.append(y)
            ys.append(y)
        else:
            ys.append(None)
    parr = lambda ts: (ctypes.c_void_p * n)(*[(t.data_ptr() if t is not None else None) for t in ts])
    iarr = lambda key: (ctypes.c_int * n)(*[int(l[key]) for l in layers])
    farr = (ctypes.c_float * n)(*[float(l["mul"]) for l in layers])
    with torch.cuda.device(x2.device):
        rc = lib.ssdhip_conv_chain_x3_nhwc_f16(_ptr(x2), b, h, w, c2 // 2, n, parr([l["packed"] for l in layers]),
                                               parr([l.get("bias") for l in layers]), parr(ys), iarr("k"), iarr("stride"), iarr("pad"),
                                               iarr("cout"), iarr("relu"), farr, current_stream_ptr(x2.device))
    if rc == -1:                                          # SSDHIP_E_BADARG: the chain does not fit this kernel
        return None
    check(rc, "ssdhip_conv_chain_x3_nhwc_f16")
    return outs


def conv3x3_wgrad(x, dy, bias_partial=None):
    """Weight gradient of a 3x3 'same' stride-1 convolution (csrc/ssdhip_wgrad.hip): x (B, Cin, H, W) and dy (B, Cout, H, W) bfloat16
    channels_last -> float32 (Cout, Cin, 3, 3) in channels_last memory format ([Cout, 3, 3, Cin] physical), or None when the
    geometry is not supported (the caller falls back to the framework's convolution_backward).  bias_partial: float32 [rows, Cout]
    per-workgroup channel sums of dy (relu_bwd_bias(..., reduce=False) and friends): the result is then (dw, db), db float32 [Cout] =
    their sum over axis 0, added by extra workgroups of the reduction launch."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_wgrad_bound", False):
        lib.ssdhip_conv3x3_wgrad_workspace_bytes.restype = ctypes.c_size_t
        lib.ssdhip_conv3x3_wgrad_workspace_bytes.argtypes = [ctypes.c_int] * 5
        lib.ssdhip_conv3x3_wgrad_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv3x3_wgrad_nhwc_bf16.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        lib.ssdhip_conv3x3_wgrad_bias_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv3x3_wgrad_bias_nhwc_bf16.argtypes = ([ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 5 +
                                                            [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p])
        lib._wgrad_bound = True
    x, (b, h, w, cin) = _nhwc_bf16(x, "x")
    dy, (b2, h2, w2, cout) = _nhwc_bf16(dy, "dy")
    if (b2, h2, w2) != (b, h, w):
        raise SsdHipError("x and dy must cover the same pixels")
    need = int(lib.ssdhip_conv3x3_wgrad_workspace_bytes(b, h, w, cin, cout))
    if need == 0:
        return None
    ws = workspaces.get(x.device, "wgrad", need)
    dw = torch.empty((cout, 3, 3, cin), dtype=torch.float32, device=x.device)
    if bias_partial is not None:
        if (bias_partial.dtype != torch.float32 or bias_partial.dim() != 2 or bias_partial.shape[1] != cout or not bias_partial.is_contiguous()):
            raise SsdHipError("bias_partial must be a contiguous float32 [rows, Cout] tensor")
        db = torch.empty((cout,), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.ssdhip_conv3x3_wgrad_bias_nhwc_bf16(_ptr(x), _ptr(dy), _ptr(dw), _ptr(bias_partial), int(bias_partial.shape[0]), _ptr(db),
                                                         b, h, w, cin, cout, _ptr(ws), need, current_stream_ptr(x.device))
        check(rc, "ssdhip_conv3x3_wgrad_bias_nhwc_bf16")
        return dw.permute(0, 3, 1, 2), db
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_conv3x3_wgrad_nhwc_bf16(_ptr(x), _ptr(dy), _ptr(dw), b, h, w, cin, cout, _ptr(ws), need, current_stream_ptr(x.device))
    check(rc, "ssdhip_conv3x3_wgrad_nhwc_bf16")
    return dw.permute(0, 3, 1, 2)


def conv1x1_wgrad(x, dy, bias_partial=None):
    """Weight gradient of a 1x1 stride-1 convolution as a pixel-contraction GEMM (csrc/ssdhip_wgrad.hip, conv1x1_wgrad_kernel):
    x (B, Cin, H, W), dy (B, Cout, H, W) bfloat16 channels_last -> float32 (Cout, Cin, 1, 1), or None when the channel counts are not
    multiples of 128 (the caller falls back to the framework).  bias_partial as in `conv3x3_wgrad`: the result is then (dw, db)."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_wgrad1_bound", False):
        lib.ssdhip_conv1x1_wgrad_workspace_bytes.restype = ctypes.c_size_t
        lib.ssdhip_conv1x1_wgrad_workspace_bytes.argtypes = [ctypes.c_longlong, ctypes.c_int, ctypes.c_int]
        lib.ssdhip_conv1x1_wgrad_bias_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv1x1_wgrad_bias_nhwc_bf16.argtypes = ([ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                                                             ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p])
        lib._wgrad1_bound = True
    x, (b, h, w, cin) = _nhwc_bf16(x, "x")
    dy, (b2, h2, w2, cout) = _nhwc_bf16(dy, "dy")
    if (b2, h2, w2) != (b, h, w):
        raise SsdHipError("x and dy must cover the same pixels")
    need = int(lib.ssdhip_conv1x1_wgrad_workspace_bytes(b * h * w, cin, cout))
    if need == 0:
        return None
    ws = workspaces.get(x.device, "wgrad", need)
    dw = torch.empty((cout, 1, 1, cin), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)     # channels_last strides, like the 3 x 3 form
    db = None
    rows = 0
    if bias_partial is not None:
        if (bias_partial.dtype != torch.float32 or bias_partial.dim() != 2 or bias_partial.shape[1] != cout or not bias_partial.is_contiguous()):
            raise SsdHipError("bias_partial must be a contiguous float32 [rows, Cout] tensor")
        db = torch.empty((cout,), dtype=torch.float32, device=x.device)
        rows = int(bias_partial.shape[0])
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_conv1x1_wgrad_bias_nhwc_bf16(_ptr(x), _ptr(dy), _ptr(dw), _ptr(bias_partial), rows, _ptr(db), b * h * w, cin, cout,
                                                     _ptr(ws), need, current_stream_ptr(x.device))
    check(rc, "ssdhip_conv1x1_wgrad_bias_nhwc_bf16")
    return (dw, db) if bias_partial is not None else dw


def row_sums(partial):
    """partial [rows, C] float32 (C % 4 == 0) -> [C]: the rows added in a fixed order by a libssdhip launch (csrc/ssdhip_wgrad.hip,
    ssdhip_row_sums_f32).  `partial.sum(0)` in the framework zeroes its semaphores with a memset node, which a replayed HIP graph
    of the training step does not honour on this runtime."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_rowsums_bound", False):
        lib.ssdhip_row_sums_f32.restype = ctypes.c_int
        lib.ssdhip_row_sums_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        lib._rowsums_bound = True
    if not partial.is_cuda or partial.dtype != torch.float32 or partial.dim() < 2 or not partial.is_contiguous():
        raise SsdHipError("row_sums needs a contiguous float32 CUDA tensor [rows, ...]")
    rows = int(partial.shape[0])
    c = partial.numel() // max(rows, 1)
    if rows == 0 or c % 4:
        return partial.sum(dim=0)
    out = torch.empty(partial.shape[1:], dtype=torch.float32, device=partial.device)
    with torch.cuda.device(partial.device):
        rc = lib.ssdhip_row_sums_f32(_ptr(partial), rows, c, _ptr(out), current_stream_ptr(partial.device))
    check(rc, "ssdhip_row_sums_f32")
    return out


def conv3x3_taps_wgrad(x, dy, stride=1, padding=1, dilation=1, bias_partial=None):
    """Weight gradient of a 3x3 convolution of any stride / padding / dilation (csrc/ssdhip_wgrad.hip, conv_taps_wgrad_kernel: fc6's
    dilation 6, the stride-2 conv6_2 / conv7_2, the 'valid' conv8_2 / conv9_2): x (B, Cin, H, W), dy (B, Cout, Ho, Wo) bfloat16
    channels_last -> float32 (Cout, Cin, 3, 3) in channels_last memory, or None when the channel counts are not multiples of 128 or dy's
    size is not the convolution's output size (the caller falls back to the framework).  bias_partial as in `conv3x3_wgrad`: the result is
    then (dw, db)."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_wgradt_bound", False):
        lib.ssdhip_conv3x3_taps_wgrad_workspace_bytes.restype = ctypes.c_size_t
        lib.ssdhip_conv3x3_taps_wgrad_workspace_bytes.argtypes = [ctypes.c_int] * 10
        lib.ssdhip_conv3x3_taps_wgrad_bias_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv3x3_taps_wgrad_bias_nhwc_bf16.argtypes = ([ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 10
                                                                  + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p])
        lib._wgradt_bound = True
    x, (b, h, w, cin) = _nhwc_bf16(x, "x")
    dy, (b2, ho, wo, cout) = _nhwc_bf16(dy, "dy")
    if b2 != b:
        raise SsdHipError("x and dy must have the same batch size")
    geom = (b, h, w, cin, ho, wo, cout, int(stride), int(padding), int(dilation))
    need = int(lib.ssdhip_conv3x3_taps_wgrad_workspace_bytes(*geom))
    if need == 0:
        return None
    ws = workspaces.get(x.device, "wgrad", need)
    dw = torch.empty((cout, 3, 3, cin), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)
    db = None
    rows = 0
    if bias_partial is not None:
        if (bias_partial.dtype != torch.float32 or bias_partial.dim() != 2 or bias_partial.shape[1] != cout or not bias_partial.is_contiguous()):
            raise SsdHipError("bias_partial must be a contiguous float32 [rows, Cout] tensor")
        db = torch.empty((cout,), dtype=torch.float32, device=x.device)
        rows = int(bias_partial.shape[0])
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_conv3x3_taps_wgrad_bias_nhwc_bf16(_ptr(x), _ptr(dy), _ptr(dw), _ptr(bias_partial), rows, _ptr(db), *geom, _ptr(ws), need,
                                                          current_stream_ptr(x.device))
    check(rc, "ssdhip_conv3x3_taps_wgrad_bias_nhwc_bf16")
    return (dw, db) if bias_partial is not None else dw


def embed_strided(gy, h, w, stride, offset):
    """gy (B, C, Ho, Wo) bfloat16 channels_last -> z (B, C, h, w): zeros with gy at (offset + stride i, offset + stride j)
    (csrc/ssdhip_train.hip, embed_strided_kernel).  The 3x3 'same' convolution of z with the transposed, tap-flipped filters is the data
    gradient of the 3x3 convolution with that stride and padding 1 - offset."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_embed_bound", False):
        lib.ssdhip_embed_strided_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_embed_strided_nhwc_bf16.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 8 + [ctypes.c_void_p]
        lib._embed_bound = True
    gy, (b, ho, wo, c) = _nhwc_bf16(gy, "gy")
    z = torch.empty((b, int(h), int(w), c), dtype=torch.bfloat16, device=gy.device).permute(0, 3, 1, 2)
    with torch.cuda.device(gy.device):
        rc = lib.ssdhip_embed_strided_nhwc_bf16(_ptr(gy), _ptr(z), b, ho, wo, c, int(h), int(w), int(stride), int(offset), current_stream_ptr(gy.device))
    check(rc, "ssdhip_embed_strided_nhwc_bf16")
    return z


def conv3x3_halo_group(xs, weights, biases=None, relu=False, max_workgroups=0):
    """Several independent 3x3 'same' convolutions through the slab kernel in ONE launch (persistent workgroups, deepest problem
    first): the packed predictor heads.  xs[i] (B, Cin_i, H_i, W_i) bf16 NHWC memory, weights[i] (Cout_i, Cin_i, 3, 3) bf16
    channels_last; Cin_i % 128 == 0, Cout_i % 128 == 0, W_i <= 62 -> list of outputs (bit-identical to conv2d_same)."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_halogroup_bound", False):
        lib.ssdhip_conv3x3_halo_group_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv3x3_halo_group_nhwc_bf16.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        lib._halogroup_bound = True
    n = len(xs)
    keep, xp, wp, bp, yp, dims, ys = [], [], [], [], [], [], []
    for i in range(n):
        x, (b, h, w, cin) = _nhwc_bf16(xs[i], "x")
        wt = weights[i]
        cout, cin_w, kh, kw = wt.shape
        if wt.dtype != torch.bfloat16 or cin_w != cin or kh != 3 or kw != 3:
            raise SsdHipError("weight %d must be bfloat16 (Cout, %d, 3, 3)" % (i, cin))
        if not wt.permute(0, 2, 3, 1).is_contiguous():
            wt = wt.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        y = torch.empty((b, h, w, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
        keep += [x, wt]
        ys.append(y)
        xp.append(x.data_ptr()); wp.append(wt.data_ptr()); yp.append(y.data_ptr())
        bp.append(biases[i].data_ptr() if (biases is not None and biases[i] is not None) else 0)
        dims.append((b, h, w, cin, cout))
    parr = lambda v: (ctypes.c_void_p * n)(*v)
    iarr = lambda k: (ctypes.c_int * n)(*[d[k] for d in dims])
    dev = ys[0].device
    with torch.cuda.device(dev):
        rc = lib.ssdhip_conv3x3_halo_group_nhwc_bf16(n, parr(xp), parr(wp), parr(bp), parr(yp), iarr(0), iarr(1), iarr(2), iarr(3),
                                                     iarr(4), int(bool(relu)), int(max_workgroups), current_stream_ptr(dev))
    check(rc, "ssdhip_conv3x3_halo_group_nhwc_bf16")
    return ys


def conv2d(x, weight, bias, stride=1, padding=0, dilation=1, relu=True, variant=None):
    """Convolution (kernel 1 or 3, stride 1..4, zero padding <= (k//2)*dilation: torch.nn.Conv2d semantics) + bias + ReLU in
    ONE libssdhip MFMA kernel -- the strided / 'valid' extra layers of the SSD trunk.  Layouts as conv2d_same."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_convgen_bound", False):
        lib.ssdhip_conv2d_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv2d_nhwc_bf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 10 + [ctypes.c_void_p]
        lib.ssdhip_conv2d_nhwc_bf16_variant.restype = ctypes.c_int
        lib.ssdhip_conv2d_nhwc_bf16_variant.argtypes = [ctypes.c_int] + lib.ssdhip_conv2d_nhwc_bf16.argtypes
        lib._convgen_bound = True
    x, (b, h, w, cin) = _nhwc_bf16(x, "x")
    cout, cin_w, kh, kw = weight.shape
    if weight.dtype != torch.bfloat16 or cin_w != cin or kh != kw:
        raise SsdHipError("weight must be bfloat16 (Cout, %d, k, k)" % cin)
    wt = weight if weight.permute(0, 2, 3, 1).is_contiguous() else weight.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    span = int(dilation) * (int(kh) - 1) + 1
    ho, wo = (h + 2 * int(padding) - span) // int(stride) + 1, (w + 2 * int(padding) - span) // int(stride) + 1
    if ho < 1 or wo < 1:
        raise SsdHipError("convolution output would be empty")
    y = torch.empty((b, ho, wo, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    with torch.cuda.device(x.device):
        args = (_ptr(x), _ptr(wt), _ptr(bias), _ptr(y), b, h, w, cin, cout, int(kh), int(stride), int(padding), int(dilation),
                int(bool(relu)), current_stream_ptr(x.device))
        if variant == 7:                                     # the slab kernel's strided / cropped form (3x3, dilation 1 only)
            if not getattr(lib, "_halostr_bound", False):
                lib.ssdhip_conv3x3_halo_strided_nhwc_bf16.restype = ctypes.c_int
                lib.ssdhip_conv3x3_halo_strided_nhwc_bf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 8 + [ctypes.c_void_p]
                lib._halostr_bound = True
            if int(kh) != 3 or int(dilation) != 1:
                raise SsdHipError("variant 7 is a 3x3, dilation-1 kernel")
            rc = lib.ssdhip_conv3x3_halo_strided_nhwc_bf16(_ptr(x), _ptr(wt), _ptr(bias), _ptr(y), b, h, w, cin, cout, int(stride),
                                                           int(padding), int(bool(relu)), current_stream_ptr(x.device))
        elif variant == 8:                                   # split-K: K ranges side by side, float32 partial tiles, ordered reduction
            if not getattr(lib, "_splitk_bound", False):
                lib.ssdhip_conv2d_splitk_workspace_bytes.restype = ctypes.c_size_t
                lib.ssdhip_conv2d_splitk_workspace_bytes.argtypes = [ctypes.c_int] * 10
                lib.ssdhip_conv2d_splitk_nhwc_bf16.restype = ctypes.c_int
                lib.ssdhip_conv2d_splitk_nhwc_bf16.argtypes = ([ctypes.c_void_p] * 4 + [ctypes.c_int] * 11 +
                                                              [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p])
                lib._splitk_bound = True
            geo = (b, h, w, cin, cout, int(kh), int(stride), int(padding), int(dilation))
            need = lib.ssdhip_conv2d_splitk_workspace_bytes(*geo, 0)
            if need == 0:
                raise SsdHipError("ssdhip_conv2d_splitk_nhwc_bf16: unsupported geometry")
            ws = workspaces.get(x.device, "conv_splitk", need)
            rc = lib.ssdhip_conv2d_splitk_nhwc_bf16(_ptr(x), _ptr(wt), _ptr(bias), _ptr(y), *geo, int(bool(relu)), 0, _ptr(ws),
                                                    ws.numel(), current_stream_ptr(x.device))
        else:
            rc = lib.ssdhip_conv2d_nhwc_bf16(*args) if variant is None else lib.ssdhip_conv2d_nhwc_bf16_variant(int(variant), *args)
    check(rc, "ssdhip_conv2d_nhwc_bf16")
    return y


def _x3_glue(lib):
    if not getattr(lib, "_x3g_bound", False):
        c_int, c_vp, c_ll = ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong
        lib.ssdhip_x3_split_nhwc.restype = c_int
        lib.ssdhip_x3_split_nhwc.argtypes = [c_vp, c_vp, c_ll, c_int, c_vp]
        lib.ssdhip_x3_merge_nhwc.restype = c_int
        lib.ssdhip_x3_merge_nhwc.argtypes = [c_vp, c_vp, c_ll, c_int, c_vp]
        lib.ssdhip_conv1_1_x3_nhwc.restype = c_int
        lib.ssdhip_conv1_1_x3_nhwc.argtypes = [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]
        lib._x3g_bound = True
    return lib


def _nhwc_ok(t):
    return t.dim() == 4 and t.permute(0, 2, 3, 1).is_contiguous()


def x3_merge(y2):
    """float16 (B, 2C, H, W) channels_last [hi | lo] -> float32 (B, C, H, W) channels_last: hi + lo (exact), one pass."""
    torch = _torch()
    b, c2, h, w = y2.shape
    c = c2 // 2
    if not (y2.is_cuda and y2.dtype == torch.float16 and c % 8 == 0 and _nhwc_ok(y2)):
        return y2[:, :c].float() + y2[:, c:].float()
    lib = _x3_glue(load())
    out = torch.empty((b, h, w, c), dtype=torch.float32, device=y2.device).permute(0, 3, 1, 2)
    with torch.cuda.device(y2.device):
        rc = lib.ssdhip_x3_merge_nhwc(_ptr(y2), _ptr(out), b * h * w, c, current_stream_ptr(y2.device))
    check(rc, "ssdhip_x3_merge_nhwc")
    return out


def x3_maxpool(y2, kernel, stride, padding=0, ceil_mode=False):
    """MaxPooling2D on a float16 (B, 2C, H, W) channels_last pair map -> the pair map of the windows' largest values
    (ssdhip_x3_maxpool_nhwc; torch.nn.functional.max_pool2d's output size and clipped windows)."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_x3pool_bound", False):
        lib.ssdhip_x3_maxpool_nhwc.restype = ctypes.c_int
        lib.ssdhip_x3_maxpool_nhwc.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 9 + [ctypes.c_void_p]
        lib._x3pool_bound = True
    b, c2, h, w = y2.shape
    if not (y2.is_cuda and y2.dtype == torch.float16 and (c2 // 2) % 8 == 0 and c2 % 2 == 0 and _nhwc_ok(y2)):
        raise SsdHipError("x3_maxpool takes a float16 (B, 2 C, H, W) channels_last pair map with C % 8 == 0")
    k, s, p = int(kernel), int(stride), int(padding)

    def out(n):
        v = n + 2 * p - k
        o = (-(-v // s) if ceil_mode else v // s) + 1
        if ceil_mode and (o - 1) * s >= n + p:               # torch: the last window must start inside the map or its left padding
            o -= 1
        return o
    ho, wo = out(h), out(w)
    y = torch.empty((b, ho, wo, c2), dtype=torch.float16, device=y2.device).permute(0, 3, 1, 2)
    with torch.cuda.device(y2.device):
        rc = lib.ssdhip_x3_maxpool_nhwc(_ptr(y2), _ptr(y), b, h, w, c2 // 2, k, s, p, ho, wo, current_stream_ptr(y2.device))
    check(rc, "ssdhip_x3_maxpool_nhwc")
    return y


def x3_l2_normalize(y2, gamma, scale=1.0):
    """L2Normalization of a float16 (B, 2C, H, W) channels_last pair map whose true values are (hi + lo) * scale -> the pair map of
    gamma * x / max(||x||_2, 1e-6) over the channel axis, stored with divisor 1 (ssdhip_x3_l2_normalize_nhwc: the float32 result of
    l2_normalize on the merged map, re-split)."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_x3norm_bound", False):
        lib.ssdhip_x3_l2_normalize_nhwc.restype = ctypes.c_int
        lib.ssdhip_x3_l2_normalize_nhwc.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_longlong, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
        lib._x3norm_bound = True
    b, c2, h, w = y2.shape
    if not (y2.is_cuda and y2.dtype == torch.float16 and (c2 // 2) % 8 == 0 and c2 % 2 == 0 and _nhwc_ok(y2)):
        raise SsdHipError("x3_l2_normalize takes a float16 (B, 2 C, H, W) channels_last pair map with C % 8 == 0")
    g = gamma.detach()
    if g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != c2 // 2 or g.device != y2.device:
        raise SsdHipError("gamma must be a contiguous float32 tensor of C elements on the map's device")
    y = torch.empty((b, h, w, c2), dtype=torch.float16, device=y2.device).permute(0, 3, 1, 2)
    with torch.cuda.device(y2.device):
        rc = lib.ssdhip_x3_l2_normalize_nhwc(_ptr(y2), _ptr(g), _ptr(y), b * h * w, c2 // 2, ctypes.c_float(float(scale)),
                                             current_stream_ptr(y2.device))
    check(rc, "ssdhip_x3_l2_normalize_nhwc")
    return y


def conv1_1_x3(x, weight, bias, relu=True):
    """conv1_1 of the reference-precision path: float32 (B, 3, H, W) channels_last images, float32 (64, 3, 3, 3) filters -> the split
    float16 (B, 128, H, W) map (ssdhip_conv1_1_x3_nhwc)."""
    torch = _torch()
    lib = _x3_glue(load())
    b, c, h, w = x.shape
    if not (x.is_cuda and x.dtype == torch.float32 and c == 3 and _nhwc_ok(x) and tuple(weight.shape) == (64, 3, 3, 3)):
        raise SsdHipError("conv1_1_x3 takes float32 (B, 3, H, W) channels_last images and (64, 3, 3, 3) filters")
    wk = weight.detach().float().permute(0, 2, 3, 1).contiguous()          # (co, kh, kw, ci)
    bk = bias.detach().float().contiguous() if bias is not None else None
    y = torch.empty((b, h, w, 128), dtype=torch.float16, device=x.device).permute(0, 3, 1, 2)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_conv1_1_x3_nhwc(_ptr(x), _ptr(wk), _ptr(bk), _ptr(y), b, h, w, int(bool(relu)), current_stream_ptr(x.device))
    check(rc, "ssdhip_conv1_1_x3_nhwc")
    return y


def conv1_1_x3_pre(images, weight, bias, mean=None, divide=None, swap=None, relu=True):
    """conv1_1 of the reference-precision path straight from the generator's float32 (B, H, W, 3) images: the graph's input Lambdas (mean
    subtraction, stddev division, channel swap) are applied while the kernel stages its input (ssdhip_conv1_1_x3_pre_nhwc) -> the split
    float16 (B, 128, H, W) map."""
    torch = _torch()
    lib = _x3_glue(load())
    if not getattr(lib, "_c11pre_bound", False):
        lib.ssdhip_conv1_1_x3_pre_nhwc.restype = ctypes.c_int
        lib.ssdhip_conv1_1_x3_pre_nhwc.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 4
        lib._c11pre_bound = True
    if not (images.is_cuda and images.dtype == torch.float32 and images.dim() == 4 and images.shape[3] == 3 and images.is_contiguous()
            and tuple(weight.shape) == (64, 3, 3, 3)):
        raise SsdHipError("conv1_1_x3_pre takes contiguous float32 (B, H, W, 3) images and (64, 3, 3, 3) filters")
    b, h, w, _ = images.shape
    wk = weight.detach().float().permute(0, 2, 3, 1).contiguous()          # (co, kh, kw, ci)
    bk = bias.detach().float().contiguous() if bias is not None else None
    f3 = lambda v: (ctypes.c_float * 3)(*[float(t) for t in v]) if v is not None else None
    i3 = (ctypes.c_int * 3)(*[int(t) for t in swap]) if swap is not None else None
    y = torch.empty((b, h, w, 128), dtype=torch.float16, device=images.device).permute(0, 3, 1, 2)
    with torch.cuda.device(images.device):
        rc = lib.ssdhip_conv1_1_x3_pre_nhwc(_ptr(images), _ptr(wk), _ptr(bk), _ptr(y), b, h, w, int(bool(relu)), f3(mean), f3(divide), i3,
                                            current_stream_ptr(images.device))
    check(rc, "ssdhip_conv1_1_x3_pre_nhwc")
    return y


def x3_split(v):
    """float32 (B, C, H, W) in channels_last memory -> float16 (B, 2C, H, W) channels_last = [hi | lo], hi = fl16(v), lo = fl16(v - hi):
    the activation layout of conv2d_x3."""
    torch = _torch()
    if v.is_cuda and v.dtype == torch.float32 and v.dim() == 4 and v.shape[1] % 8 == 0:
        if not _nhwc_ok(v):
            v = v.contiguous(memory_format=torch.channels_last)
        if _nhwc_ok(v):
            lib = _x3_glue(load())
            b, c, h, w = v.shape
            out = torch.empty((b, h, w, 2 * c), dtype=torch.float16, device=v.device).permute(0, 3, 1, 2)
            with torch.cuda.device(v.device):
                rc = lib.ssdhip_x3_split_nhwc(_ptr(v), _ptr(out), b * h * w, c, current_stream_ptr(v.device))
            check(rc, "ssdhip_x3_split_nhwc")
            return out
    hi = v.to(torch.float16)
    lo = (v - hi.float()).to(torch.float16)
    return torch.cat([hi, lo], dim=1).contiguous(memory_format=torch.channels_last)


def x3_pack_weight(weight, slab64=False):
    """float32 (Cout, Cin, k, k) filters -> (float16 (Cout, 3 Cin, k, k) channels_last = [w hi | w lo | w hi] of weight * 2^e, oscale =
    2^-e), e chosen so that the largest scaled filter lies in [256, 512): hi and lo parts stay in float16's normal range.
    slab64 (Cin == 64, 3x3 'same', Cout % 128 == 0: conv2_1 on the slab kernel): (Cout, 256, 3, 3) = [w hi | w hi | w lo | 0]."""
    import math
    torch = _torch()
    w = weight.detach().float()
    m = float(w.abs().max().item())
    e = 8 - int(math.floor(math.log2(m))) if m > 0 and math.isfinite(m) else 0
    ws = w * (2.0 ** e)
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    if slab64:
        if w.shape[1] != 64:
            raise SsdHipError("slab64 packing is for 64 input channels")
        return torch.cat([hi, hi, lo, torch.zeros_like(hi)], dim=1).contiguous(memory_format=torch.channels_last), 2.0 ** -e
    return torch.cat([hi, lo, hi], dim=1).contiguous(memory_format=torch.channels_last), 2.0 ** -e


def conv2d_x3(x2, packed_weight, bias, oscale, stride=1, padding=0, dilation=1, relu=True, pool=False, out_f32=False):
    """Reference-precision convolution on the float16 MFMA path (ssdhip_conv2d_x3_nhwc_f16): x2 = x3_split(activation),
    (packed_weight, oscale) = x3_pack_weight(filters), bias float32 or None.  Returns the split float16 (B, 2 Cout, Ho, Wo) map, or
    with out_f32 the float32 (B, Cout, Ho, Wo) one (both channels_last)."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_x3_bound", False):
        lib.ssdhip_conv2d_x3_nhwc_f16.restype = ctypes.c_int
        lib.ssdhip_conv2d_x3_nhwc_f16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 12 + [ctypes.c_float, ctypes.c_void_p]
        lib._x3_bound = True
    require_cuda(x2.permute(0, 2, 3, 1), "x2")
    if x2.dtype != torch.float16 or packed_weight.dtype != torch.float16:
        raise SsdHipError("conv2d_x3 takes float16 split activations and packed float16 filters")
    b, c2, h, w = x2.shape
    cout, c3, kh, kw = packed_weight.shape
    slab64 = c2 == 128 and c3 == 256 and int(kh) == 3                       # x3_pack_weight(..., slab64=True)
    if c2 % 2 or (c3 != 3 * (c2 // 2) and not slab64) or kh != kw or not packed_weight.permute(0, 2, 3, 1).is_contiguous():
        raise SsdHipError("packed filters must be (Cout, 3 C, k, k) channels_last for a (B, 2 C, H, W) input")
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous()):
        raise SsdHipError("bias must be contiguous float32")
    span = int(dilation) * (int(kh) - 1) + 1
    ho, wo = (h + 2 * int(padding) - span) // int(stride) + 1, (w + 2 * int(padding) - span) // int(stride) + 1
    if pool:
        ho, wo = (h + 1) // 2, (w + 1) // 2
    if ho < 1 or wo < 1:
        raise SsdHipError("convolution output would be empty")
    import os
    c = c2 // 2
    if slab64 and not (int(stride) == 1 and int(padding) == 1 and int(dilation) == 1 and cout % 128 == 0):
        raise SsdHipError("slab64 filters are for a 3x3 'same' convolution with Cout % 128 == 0")
    # round 6: small maps take the image-resident kernel AHEAD of the slab kernel too (conv5_x: 6.72 -> 6.57 ms per step, r06n);
    # SSDHIP_X3_IMAGE = 0: never, 1: only where the slab kernel does not apply
    image_first = (os.environ.get("SSDHIP_X3_IMAGE", "2") == "2" and not pool and not slab64 and h * w <= 384 and c % 64 == 0
                   and b * (cout // 64) >= int(os.environ.get("SSDHIP_X3_IMAGE_MIN_TILES", "128")))
    if (int(kh) == 3 and int(stride) == 1 and int(padding) == 1 and int(dilation) == 1 and (c % 128 == 0 or slab64) and cout % 128 == 0
            and (slab64 or os.environ.get("SSDHIP_X3_NO_HALO", "0") != "1") and not image_first):
        # the slab kernel (csrc/ssdhip_convh.hip): the deep 3x3 layers and the packed heads; it writes split pairs, merged here when
        # the caller wants float32
        if not getattr(lib, "_x3h_bound", False):
            lib.ssdhip_conv3x3_halo_x3_nhwc_f16.restype = ctypes.c_int
            lib.ssdhip_conv3x3_halo_x3_nhwc_f16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7 + [ctypes.c_float, ctypes.c_void_p]
            lib._x3h_bound = True
        y2 = torch.empty((b, ho, wo, 2 * cout), dtype=torch.float16, device=x2.device).permute(0, 3, 1, 2)
        with torch.cuda.device(x2.device):
            rc = lib.ssdhip_conv3x3_halo_x3_nhwc_f16(_ptr(x2), _ptr(packed_weight), _ptr(bias), _ptr(y2), b, h, w, c, cout, int(bool(relu)),
                                                     int(bool(pool)), ctypes.c_float(float(oscale)), current_stream_ptr(x2.device))
        check(rc, "ssdhip_conv3x3_halo_x3_nhwc_f16")
        return x3_merge(y2) if out_f32 else y2
    y = (torch.empty((b, ho, wo, cout), dtype=torch.float32, device=x2.device) if out_f32 else
         torch.empty((b, ho, wo, 2 * cout), dtype=torch.float16, device=x2.device)).permute(0, 3, 1, 2)
    if (not pool and not slab64 and int(kh) in (1, 3) and h * w <= 384 and ho * wo <= 384 and c % 64 == 0 and cout % 64 == 0
            and b * (cout // 64) >= (int(os.environ.get("SSDHIP_X3_IMAGE1_MIN_TILES", "128")) if int(kh) == 1 else 128)
            and 1 <= int(stride) <= 4 and 1 <= int(dilation) <= 16
            and 0 <= int(padding) <= int(dilation) * (int(kh) // 2) and os.environ.get("SSDHIP_X3_IMAGE", "2") != "0"
            and hasattr(lib, "ssdhip_conv2d_image_x3_nhwc_f16")):
        # round 6: small maps (fc6, fc7, conv6_x) with the image's slices resident in LDS (csrc/ssdhip_convimg.hip, X3): the
        # implicit-GEMM form below gathers every tap's pixels again and moves 2.5-3.5 x the bytes per FLOP from L2
        if not getattr(lib, "_x3img_bound", False):
            lib.ssdhip_conv2d_image_x3_nhwc_f16.restype = ctypes.c_int
            lib.ssdhip_conv2d_image_x3_nhwc_f16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 11 + [ctypes.c_float, ctypes.c_void_p]
            lib._x3img_bound = True
        with torch.cuda.device(x2.device):
            rc = lib.ssdhip_conv2d_image_x3_nhwc_f16(_ptr(x2), _ptr(packed_weight), _ptr(bias), _ptr(y), b, h, w, c, cout, int(kh),
                                                     int(stride), int(padding), int(dilation), int(bool(relu)), int(bool(out_f32)),
                                                     ctypes.c_float(float(oscale)), current_stream_ptr(x2.device))
        check(rc, "ssdhip_conv2d_image_x3_nhwc_f16")
        return y
    with torch.cuda.device(x2.device):
        rc = lib.ssdhip_conv2d_x3_nhwc_f16(_ptr(x2), _ptr(packed_weight), _ptr(bias), _ptr(y), b, h, w, c2 // 2, cout, int(kh), int(stride),
                                           int(padding), int(dilation), int(bool(relu)), int(bool(pool)), int(bool(out_f32)),
                                           ctypes.c_float(float(oscale)), current_stream_ptr(x2.device))
    check(rc, "ssdhip_conv2d_x3_nhwc_f16")
    return y


def conv2d_same_pool2(x, weight, bias, dilation=1, relu=True):
    """'same' convolution + bias + ReLU + 2x2 / stride-2 max-pool ('same' = windows clipped to the map) in ONE libssdhip
    MFMA kernel.  x (B, Cin, H, W) bf16 NHWC memory -> (B, Cout, ceil(H/2), ceil(W/2))."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_convpool_bound", False):
        lib.ssdhip_conv2d_same_pool2_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv2d_same_pool2_nhwc_bf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 8 + [ctypes.c_void_p]
        lib._convpool_bound = True
    x, (b, h, w, cin) = _nhwc_bf16(x, "x")
    cout, cin_w, kh, kw = weight.shape
    if weight.dtype != torch.bfloat16 or cin_w != cin or kh != kw:
        raise SsdHipError("weight must be bfloat16 (Cout, %d, k, k)" % cin)
    wt = weight if weight.permute(0, 2, 3, 1).is_contiguous() else weight.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    y = torch.empty((b, (h + 1) // 2, (w + 1) // 2, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_conv2d_same_pool2_nhwc_bf16(_ptr(x), _ptr(wt), _ptr(bias), _ptr(y), b, h, w, cin, cout, int(kh), int(dilation),
                                                    int(bool(relu)), current_stream_ptr(x.device))
    check(rc, "ssdhip_conv2d_same_pool2_nhwc_bf16")
    return y


def conv2d_same_group(xs, weights, biases=None, relu=False):
    """Several independent 'same' convolutions (kernel 1 or 3, stride 1, dilation 1) in ONE libssdhip launch.
    xs[i] (B, Cin_i, H_i, W_i) bf16 NHWC memory, weights[i] (Cout_i, Cin_i, k, k) bf16 channels_last -> list of outputs."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_convgroup_bound", False):
        lib.ssdhip_conv2d_same_group_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv2d_same_group_nhwc_bf16.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 11 + [ctypes.c_int, ctypes.c_void_p]
        lib._convgroup_bound = True
    n = len(xs)
    keep, xp, wp, bp, yp, dims, ys = [], [], [], [], [], [], []
    for i in range(n):
        x, (b, h, w, cin) = _nhwc_bf16(xs[i], "x")
        wt = weights[i]
        cout, cin_w, kh, kw = wt.shape
        if wt.dtype != torch.bfloat16 or cin_w != cin or kh != kw:
            raise SsdHipError("weight %d must be bfloat16 (Cout, %d, k, k)" % (i, cin))
        if not wt.permute(0, 2, 3, 1).is_contiguous():
            wt = wt.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        y = torch.empty((b, h, w, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
        keep += [x, wt]
        ys.append(y)
        xp.append(x.data_ptr()); wp.append(wt.data_ptr()); yp.append(y.data_ptr())
        bp.append(biases[i].data_ptr() if (biases is not None and biases[i] is not None) else 0)
        dims.append((b, h, w, cin, cout, int(kh), 1))
    parr = lambda v: (ctypes.c_void_p * n)(*v)
    iarr = lambda k: (ctypes.c_int * n)(*[d[k] for d in dims])
    dev = ys[0].device
    with torch.cuda.device(dev):
        rc = lib.ssdhip_conv2d_same_group_nhwc_bf16(n, parr(xp), parr(wp), parr(bp), parr(yp), iarr(0), iarr(1), iarr(2), iarr(3),
                                                    iarr(4), iarr(5), iarr(6), int(bool(relu)), current_stream_ptr(dev))
    check(rc, "ssdhip_conv2d_same_group_nhwc_bf16")
    return ys


_CU_COUNT = {}


def conv3x3_c64(x, weight, bias, relu=True, pool=False):
    """3x3 'same' convolution of a 64-channel map + bias + ReLU [+ 2x2/2 'same' max-pool]: the resident-weight, halo-tile kernel of
    csrc/ssdhip_conv64.hip.  x (B, 64, H, W) bf16 NHWC memory, weight (Cout, 64, 3, 3) bf16 channels_last, Cout % 64 == 0."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_c64_bound", False):
        lib.ssdhip_conv3x3_c64_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv3x3_c64_nhwc_bf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 8 + [ctypes.c_void_p]
        lib._c64_bound = True
    x, (b, h, w, cin) = _nhwc_bf16(x, "x")
    cout, cin_w, kh, kw = weight.shape
    if weight.dtype != torch.bfloat16 or cin_w != cin or cin != 64 or kh != 3 or kw != 3:
        raise SsdHipError("conv3x3_c64 needs a bfloat16 (Cout, 64, 3, 3) weight and a 64-channel input")
    wt = weight if weight.permute(0, 2, 3, 1).is_contiguous() else weight.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if pool else (h, w)
    y = torch.empty((b, ho, wo, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    key = str(x.device)
    if key not in _CU_COUNT:
        _CU_COUNT[key] = int(torch.cuda.get_device_properties(x.device).multi_processor_count)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_conv3x3_c64_nhwc_bf16(_ptr(x), _ptr(wt), _ptr(bias), _ptr(y), b, h, w, cin, cout, int(bool(relu)), int(bool(pool)),
                                              _CU_COUNT[key], current_stream_ptr(x.device))
    check(rc, "ssdhip_conv3x3_c64_nhwc_bf16")
    return y


def conv3x3_c64_pool_keep(x, weight, bias, relu=True):
    """Conv2D(relu) -> MaxPooling2D(2, 2, 'same') of a 64-channel map in ONE launch that writes BOTH the full-resolution activation and
    the pooled map (csrc/ssdhip_conv64.hip, KEEP: the training step needs the former for its backward pass).  Returns (y, pooled);
    bit-identical to conv3x3_c64(pool=False) followed by bias_act_maxpool."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_c64k_bound", False):
        lib.ssdhip_conv3x3_c64_pool_keep_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv3x3_c64_pool_keep_nhwc_bf16.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
        lib._c64k_bound = True
    x, (b, h, w, cin) = _nhwc_bf16(x, "x")
    cout, cin_w, kh, kw = weight.shape
    if weight.dtype != torch.bfloat16 or cin_w != cin or cin != 64 or kh != 3 or kw != 3:
        raise SsdHipError("conv3x3_c64_pool_keep needs a bfloat16 (Cout, 64, 3, 3) weight and a 64-channel input")
    wt = weight if weight.permute(0, 2, 3, 1).is_contiguous() else weight.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    y = torch.empty((b, h, w, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    pooled = torch.empty((b, (h + 1) // 2, (w + 1) // 2, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    key = str(x.device)
    if key not in _CU_COUNT:
        _CU_COUNT[key] = int(torch.cuda.get_device_properties(x.device).multi_processor_count)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_conv3x3_c64_pool_keep_nhwc_bf16(_ptr(x), _ptr(wt), _ptr(bias), _ptr(y), _ptr(pooled), b, h, w, cin, cout, int(bool(relu)),
                                                        _CU_COUNT[key], current_stream_ptr(x.device))
    check(rc, "ssdhip_conv3x3_c64_pool_keep_nhwc_bf16")
    return y, pooled


def conv3x3_halo_pool_keep(x, weight, bias, relu=True):
    """Conv2D(relu) -> MaxPooling2D(2, 2, 'same') through the slab kernel in ONE launch that writes BOTH the full-resolution activation
    and the pooled map (csrc/ssdhip_convh.hip, KEEP; Cin, Cout % 128 == 0).  Returns (y, pooled), bit-identical to conv3x3_halo(pool=False)
    and conv3x3_halo(pool=True); None when the geometry is not the slab kernel's."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_halo_keep_bound", False):
        lib.ssdhip_conv3x3_halo_pool_keep_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv3x3_halo_pool_keep_nhwc_bf16.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
        lib._halo_keep_bound = True
    cout, cin_w, kh, kw = weight.shape
    if not x.is_cuda or x.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16 or kh != 3 or kw != 3 or cin_w % 128 or cout % 128 or x.shape[1] != cin_w:
        return None
    x, (b, h, w, cin) = _nhwc_bf16(x, "x")
    wt = weight if weight.permute(0, 2, 3, 1).is_contiguous() else weight.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    y = torch.empty((b, h, w, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    pooled = torch.empty((b, (h + 1) // 2, (w + 1) // 2, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_conv3x3_halo_pool_keep_nhwc_bf16(_ptr(x), _ptr(wt), _ptr(bias), _ptr(y), _ptr(pooled), b, h, w, cin, cout, int(bool(relu)),
                                                         current_stream_ptr(x.device))
    if rc == -1:                                          # SSDHIP_E_BADARG: not the slab kernel's geometry
        return None
    check(rc, "ssdhip_conv3x3_halo_pool_keep_nhwc_bf16")
    return y, pooled


def conv3x3_cin3(x, weight, bias, relu=True):
    """First layer: 3x3 'same' convolution of a 3-channel image into 64 channels + bias + ReLU (csrc/ssdhip_conv.hip)."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_conv1_bound", False):
        lib.ssdhip_conv3x3_cin3_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv3x3_cin3_nhwc_bf16.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
        lib._conv1_bound = True
    if not x.is_cuda or x.dtype != torch.bfloat16 or x.dim() != 4:
        raise SsdHipError("x must be a 4-D bfloat16 CUDA tensor")
    if not x.permute(0, 2, 3, 1).is_contiguous():
        x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    b, cin, h, w = x.shape
    cout = weight.shape[0]
    wt = weight if weight.permute(0, 2, 3, 1).is_contiguous() else weight.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    y = torch.empty((b, h, w, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    with torch.cuda.device(x.device):
        rc = lib.ssdhip_conv3x3_cin3_nhwc_bf16(_ptr(x), _ptr(wt), _ptr(bias), _ptr(y), b, h, w, cin, cout, int(bool(relu)),
                                               current_stream_ptr(x.device))
    check(rc, "ssdhip_conv3x3_cin3_nhwc_bf16")
    return y


def conv1_block(x, w1, b1, weight, bias, relu=True, pool=False):
    """conv1_1 (3 -> 64 channels, ReLU) -> conv1_2 (64 -> Cout) + bias + ReLU [-> 2x2 / stride-2 'same' max-pool] as ONE kernel
    (csrc/ssdhip_conv64.hip, FRONT): the 64-channel map between the two layers is never written.  x (B, 3, H, W) bf16 NHWC memory;
    w1 (64, 3, 3, 3), weight (Cout, 64, 3, 3) bf16 channels_last.  Bit-identical to conv3x3_cin3 followed by conv3x3_c64."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_conv1blk_bound", False):
        lib.ssdhip_conv1_block_nhwc_bf16.restype = ctypes.c_int
        lib.ssdhip_conv1_block_nhwc_bf16.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
        lib._conv1blk_bound = True
    if not x.is_cuda or x.dtype != torch.bfloat16 or x.dim() != 4 or x.shape[1] != 3:
        raise SsdHipError("x must be a (B, 3, H, W) bfloat16 CUDA tensor")
    if not x.permute(0, 2, 3, 1).is_contiguous():
        x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    b, _, h, w = x.shape
    if tuple(w1.shape) != (64, 3, 3, 3) or weight.shape[1:] != (64, 3, 3) or w1.dtype != torch.bfloat16 or weight.dtype != torch.bfloat16:
        raise SsdHipError("conv1_block needs bfloat16 (64, 3, 3, 3) and (Cout, 64, 3, 3) weights")
    cl = lambda t: t if t.permute(0, 2, 3, 1).is_contiguous() else t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    w1c, w2c = cl(w1), cl(weight)
    cout = weight.shape[0]
    ho, wo = ((h + 1) // 2, (w + 1) // 2) if pool else (h, w)
    y = torch.empty((b, ho, wo, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    dev = x.device
    n_cu = _CU_COUNT.get(dev.index)
    if n_cu is None:
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        _CU_COUNT[dev.index] = n_cu
    with torch.cuda.device(dev):
        rc = lib.ssdhip_conv1_block_nhwc_bf16(_ptr(x), _ptr(w1c), _ptr(b1), _ptr(w2c), _ptr(bias), _ptr(y), b, h, w, cout, int(bool(relu)),
                                              int(bool(pool)), int(n_cu), current_stream_ptr(dev))
    check(rc, "ssdhip_conv1_block_nhwc_bf16")
    return y


# ------------------------------------------------------------------------------------------------
# public box utilities (csrc/ssdhip_boxes.hip)
# ------------------------------------------------------------------------------------------------
CONVERSIONS = {"minmax2centroids": 0, "centroids2minmax": 1, "corners2centroids": 2, "centroids2corners": 3,
               "minmax2corners": 4, "corners2minmax": 4}


def _bind_boxes(lib):
    if getattr(lib, "_boxes_bound", False):
        return
    c_int, c_vp, c_ll, c_dbl, c_sz = ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_double, ctypes.c_size_t
    lib.ssdhip_convert_coordinates.restype = c_int
    lib.ssdhip_convert_coordinates.argtypes = [c_vp, c_int, c_vp, c_ll, c_int, c_int, c_int, c_int, c_vp]
    lib.ssdhip_iou_result_dtype.restype = c_int
    lib.ssdhip_iou_result_dtype.argtypes = [c_int, c_int, c_int]
    lib.ssdhip_box_overlap.restype = c_int
    lib.ssdhip_box_overlap.argtypes = [c_int, c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]
    lib.ssdhip_match_bipartite_greedy.restype = c_int
    lib.ssdhip_match_bipartite_greedy.argtypes = [c_vp, c_int, c_int, c_vp, c_vp]
    lib.ssdhip_match_multi_workspace_bytes.restype = c_sz
    lib.ssdhip_match_multi_workspace_bytes.argtypes = [c_int, c_int]
    lib.ssdhip_match_multi.restype = c_int
    lib.ssdhip_match_multi.argtypes = [c_vp, c_int, c_int, c_dbl, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]
    lib.ssdhip_greedy_nms_workspace_bytes.restype = c_sz
    lib.ssdhip_greedy_nms_workspace_bytes.argtypes = [c_int]
    lib.ssdhip_greedy_nms.restype = c_int
    lib.ssdhip_greedy_nms.argtypes = [c_vp, c_int, c_int, c_int, c_int, c_vp, c_int, c_dbl, c_int, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]
    lib._boxes_bound = True


def _boxes_lib():
    lib = load()
    _bind_boxes(lib)
    return lib


def _float_device(a, name):
    """NumPy array / torch tensor -> contiguous CUDA tensor of float32 or float64 (other dtypes -> float64, which is
    what NumPy's arithmetic would promote integer boxes to in every formula of the box utilities)."""
    torch = _torch()
    if isinstance(a, np.ndarray):
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        return to_device(a)
    if not torch.is_tensor(a):
        return to_device(np.asarray(a, dtype=np.float64))
    if a.dtype not in (torch.float32, torch.float64):
        a = a.to(torch.float64)
    return to_device(a)


def _dt(t):
    torch = _torch()
    return F32 if t.dtype == torch.float32 else F64


def convert_coordinates(t, start_index, conversion, border_pixels):
    """t: CUDA float32/float64 tensor, last axis holds the 4 coordinates from start_index.  Returns a float64 tensor."""
    torch = _torch()
    lib = _boxes_lib()
    require_cuda(t, "tensor")
    L = int(t.shape[-1])
    rows = t.numel() // L if L else 0
    out = torch.empty(t.shape, dtype=torch.float64, device=t.device)
    if t.numel() == 0:
        return out
    with torch.cuda.device(t.device):
        rc = lib.ssdhip_convert_coordinates(_ptr(t), _dt(t), _ptr(out), rows, L, int(start_index), CONVERSIONS[conversion],
                                            BORDER[border_pixels], current_stream_ptr(t.device))
    check(rc, "ssdhip_convert_coordinates")
    return out


def box_overlap(op, b1, b2, coords, mode, border_pixels):
    """op 0 iou / 1 intersection_area; b1 (m,4), b2 (n,4) CUDA tensors; mode 'outer_product' | 'element-wise'."""
    torch = _torch()
    lib = _boxes_lib()
    require_cuda(b1, "boxes1")
    require_cuda(b2, "boxes2")
    m, n = int(b1.shape[0]), int(b2.shape[0])
    rdt = torch.float32 if lib.ssdhip_iou_result_dtype(_dt(b1), _dt(b2), COORDS[coords]) == F32 else torch.float64
    outer = mode == "outer_product"
    out = torch.empty((m, n) if outer else (max(m, n) if min(m, n) > 0 else 0,), dtype=rdt, device=b1.device)
    if out.numel() == 0:
        return out
    with torch.cuda.device(b1.device):
        rc = lib.ssdhip_box_overlap(int(op), _ptr(b1), _dt(b1), m, _ptr(b2), _dt(b2), n, COORDS[coords], 0 if outer else 1,
                                    BORDER[border_pixels], _ptr(out), current_stream_ptr(b1.device))
    check(rc, "ssdhip_box_overlap")
    return out


def match_bipartite_greedy(w):
    torch = _torch()
    lib = _boxes_lib()
    require_cuda(w, "weight_matrix")
    m, n = int(w.shape[0]), int(w.shape[1])
    out = torch.zeros((m,), dtype=torch.int32, device=w.device)
    if m == 0:
        return out
    with torch.cuda.device(w.device):
        rc = lib.ssdhip_match_bipartite_greedy(_ptr(w), m, n, _ptr(out), current_stream_ptr(w.device))
    check(rc, "ssdhip_match_bipartite_greedy")
    return out


def match_multi(w, threshold):
    torch = _torch()
    lib = _boxes_lib()
    require_cuda(w, "weight_matrix")
    m, n = int(w.shape[0]), int(w.shape[1])
    gt = torch.empty((max(n, 1),), dtype=torch.int32, device=w.device)
    col = torch.empty((max(n, 1),), dtype=torch.int32, device=w.device)
    cnt = torch.zeros((1,), dtype=torch.int32, device=w.device)
    ws = workspaces.get(w.device, "match_multi", lib.ssdhip_match_multi_workspace_bytes(m, n))
    with torch.cuda.device(w.device):
        rc = lib.ssdhip_match_multi(_ptr(w), m, n, float(threshold), _ptr(gt), _ptr(col), _ptr(cnt), _ptr(ws), ws.numel(),
                                    current_stream_ptr(w.device))
    check(rc, "ssdhip_match_multi")
    k = int(cnt.item())
    return gt[:k], col[:k]


def greedy_nms_rows(rows, seg_offsets, score_col, box_col, iou_threshold, coords, border_pixels):
    """rows: CUDA float64 (n_total, L); seg_offsets: host int array (S+1,).  Returns (kept_idx (n_total,), kept_count (S,))
    as CUDA int32 tensors (see include/ssdhip.h)."""
    torch = _torch()
    lib = _boxes_lib()
    require_cuda(rows, "rows")
    n_total, L = int(rows.shape[0]), int(rows.shape[1])
    S = len(seg_offsets) - 1
    dev = rows.device
    off = torch.from_numpy(np.asarray(seg_offsets, dtype=np.int32)).to(dev)
    kept = torch.empty((max(n_total, 1),), dtype=torch.int32, device=dev)
    cnt = torch.zeros((max(S, 1),), dtype=torch.int32, device=dev)
    ws = workspaces.get(dev, "greedy_nms", lib.ssdhip_greedy_nms_workspace_bytes(n_total))
    with torch.cuda.device(dev):
        rc = lib.ssdhip_greedy_nms(_ptr(rows), n_total, L, int(score_col), int(box_col), _ptr(off), S, float(iou_threshold),
                                   COORDS[coords], BORDER[border_pixels], _ptr(kept), _ptr(cnt), _ptr(ws), ws.numel(),
                                   current_stream_ptr(dev))
    check(rc, "ssdhip_greedy_nms")
    return kept, cnt


# ------------------------------------------------------------------------------------------------
# Evaluator matching (csrc/ssdhip_eval.hip)
# ------------------------------------------------------------------------------------------------
def match_predictions_class(pred, pred_image, gt_boxes, gt_offsets, gt_neutral, matching_iou_threshold, border_pixels):
    """One class of Evaluator.match_predictions.  pred (P,5) float32 [conf, xmin, ymin, xmax, ymax], pred_image (P,) int32,
    gt_boxes (G,4) float64, gt_offsets (n_images+1,) int32, gt_neutral (G,) uint8 or None -- NumPy arrays or CUDA tensors.
    Returns CUDA int32 tensors (order, true_pos, false_pos, cum_true_pos, cum_false_pos), each (P,)."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_eval_bound", False):
        c_int, c_vp, c_dbl, c_sz = ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_size_t
        lib.ssdhip_match_predictions_workspace_bytes.restype = c_sz
        lib.ssdhip_match_predictions_workspace_bytes.argtypes = [c_int, c_int]
        lib.ssdhip_match_predictions.restype = c_int
        lib.ssdhip_match_predictions.argtypes = [c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_dbl, c_int,
                                                 c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]
        lib._eval_bound = True
    pred = to_device(pred, dtype=torch.float32)
    dev = pred.device
    pred_image = to_device(pred_image, device=dev, dtype=torch.int32)
    gt_boxes = to_device(gt_boxes, device=dev, dtype=torch.float64)
    gt_offsets = to_device(gt_offsets, device=dev, dtype=torch.int32)
    neutral = to_device(gt_neutral, device=dev, dtype=torch.uint8) if gt_neutral is not None else None
    P, G, n_images = int(pred.shape[0]), int(gt_boxes.shape[0]), int(gt_offsets.shape[0]) - 1
    outs = [torch.zeros((P,), dtype=torch.int32, device=dev) for _ in range(5)]
    if P == 0:
        return tuple(outs)
    ws = workspaces.get(dev, "match_predictions", lib.ssdhip_match_predictions_workspace_bytes(P, G))
    with torch.cuda.device(dev):
        rc = lib.ssdhip_match_predictions(_ptr(pred), _ptr(pred_image), P, _ptr(gt_boxes) if G else None, _ptr(gt_offsets), _ptr(neutral),
                                          n_images, G, float(matching_iou_threshold), BORDER[border_pixels],
                                          *[_ptr(o) for o in outs], _ptr(ws), ws.numel(), current_stream_ptr(dev))
    check(rc, "ssdhip_match_predictions")
    return tuple(outs)


def match_predictions_all(pred, pred_segment, pred_class, class_start, gt_boxes, gt_offsets, gt_neutral, matching_iou_threshold,
                          border_pixels):
    """Every class of Evaluator.match_predictions in ONE call (ssdhip_match_predictions_multi).  CUDA tensors: pred (P,5) float32
    [conf, xmin, ymin, xmax, ymax] with the classes' predictions concatenated slot by slot, pred_segment (P,) int32 = slot * n_images +
    image index, pred_class (P,) int32 = slot, class_start (n_slots + 1,) int32, gt_boxes (G,4) float64 and gt_offsets
    (n_slots * n_images + 1,) int32 CSR over the segments, gt_neutral (G,) uint8 or None.  Returns ONE CUDA int32 tensor (5, P): rows
    order, true_pos, false_pos, cum_true_pos, cum_false_pos (every slot's stretch sorted by descending confidence)."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_eval_multi_bound", False):
        c_int, c_vp, c_dbl, c_sz = ctypes.c_int, ctypes.c_void_p, ctypes.c_double, ctypes.c_size_t
        lib.ssdhip_match_predictions_workspace_bytes.restype = c_sz
        lib.ssdhip_match_predictions_workspace_bytes.argtypes = [c_int, c_int]
        lib.ssdhip_match_predictions_multi.restype = c_int
        lib.ssdhip_match_predictions_multi.argtypes = [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_int, c_dbl, c_int,
                                                       c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]
        lib._eval_multi_bound = True
    require_cuda(pred, "pred")
    dev = pred.device
    P, G = int(pred.shape[0]), int(gt_boxes.shape[0])
    n_segments, n_slots = int(gt_offsets.shape[0]) - 1, int(class_start.shape[0]) - 1
    out = torch.zeros((5, P), dtype=torch.int32, device=dev)
    if P == 0:
        return out
    ws = workspaces.get(dev, "match_predictions", lib.ssdhip_match_predictions_workspace_bytes(P, G))
    with torch.cuda.device(dev):
        rc = lib.ssdhip_match_predictions_multi(_ptr(pred), _ptr(pred_segment), _ptr(pred_class), P, _ptr(gt_boxes) if G else None,
                                                _ptr(gt_offsets), _ptr(gt_neutral) if gt_neutral is not None else None, n_segments, G,
                                                _ptr(class_start), n_slots, float(matching_iou_threshold), BORDER[border_pixels],
                                                *[ctypes.c_void_p(out[i].data_ptr()) for i in range(5)], _ptr(ws), ws.numel(),
                                                current_stream_ptr(dev))
    check(rc, "ssdhip_match_predictions_multi")
    return out


def box_filter(boxes, box_image, image_hw, check_overlap, check_min_area, check_degenerate, criterion, lower, upper, min_area,
               border_pixels):
    """Batched BoxFilter: boxes (G,4) float64 corners, box_image (G,) int32, image_hw (n_images,2) float64 -> CUDA uint8 keep mask."""
    torch = _torch()
    lib = _boxes_lib()
    if not getattr(lib, "_bf_bound", False):
        c_int, c_vp, c_dbl = ctypes.c_int, ctypes.c_void_p, ctypes.c_double
        lib.ssdhip_box_filter.restype = c_int
        lib.ssdhip_box_filter.argtypes = [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_dbl, c_dbl, c_dbl, c_int, c_vp, c_vp]
        lib._bf_bound = True
    boxes = to_device(boxes, dtype=torch.float64)
    dev = boxes.device
    box_image = to_device(box_image, device=dev, dtype=torch.int32)
    image_hw = to_device(image_hw, device=dev, dtype=torch.float64)
    G = int(boxes.shape[0])
    keep = torch.zeros((G,), dtype=torch.uint8, device=dev)
    if G == 0:
        return keep
    with torch.cuda.device(dev):
        rc = lib.ssdhip_box_filter(_ptr(boxes), _ptr(box_image), _ptr(image_hw), G, int(image_hw.shape[0]), int(bool(check_overlap)),
                                   int(bool(check_min_area)), int(bool(check_degenerate)), {"center_point": 0, "iou": 1, "area": 2}[criterion],
                                   float(lower), float(upper), float(min_area), BORDER[border_pixels], _ptr(keep), current_stream_ptr(dev))
    check(rc, "ssdhip_box_filter")
    return keep


# ---- image half of the augmentation (csrc/ssdhip_image.hip) ----------------------------------------------------------------------
IMG_U8, IMG_F32, IMG_F64 = 0, 1, 2
IMG_OPS = {"end": 0, "to_f32": 1, "to_u8": 2, "brightness": 3, "contrast": 4, "saturation": 5, "hue": 6, "rgb2hsv": 7, "hsv2rgb": 8,
           "rgb2gray": 9, "swap": 10}
IMG_PROG = 16


def _image_lib():
    lib = load()
    if not getattr(lib, "_image_bound", False):
        c_int, c_vp, c_ll = ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong
        lib.ssdhip_image_program.restype = c_int
        lib.ssdhip_image_program.argtypes = [c_vp, c_int, c_vp, c_int, c_int, c_ll, c_vp, c_vp, c_vp]
        lib.ssdhip_image_resize_u8.restype = c_int
        lib.ssdhip_image_resize_u8.argtypes = [c_vp, c_vp] + [c_int] * 6 + [c_vp, c_vp, c_int, c_vp, c_vp, c_int, c_vp]
        lib.ssdhip_image_hist_u8.restype = c_int
        lib.ssdhip_image_hist_u8.argtypes = [c_vp, c_ll, c_int, c_int, c_vp, c_vp]
        lib.ssdhip_image_lut_u8.restype = c_int
        lib.ssdhip_image_lut_u8.argtypes = [c_vp, c_vp, c_ll, c_int, c_int, c_vp, c_vp]
        lib._image_bound = True
    return lib


def _img_dtype_code(t):
    torch = _torch()
    code = {torch.uint8: IMG_U8, torch.float32: IMG_F32, torch.float64: IMG_F64}.get(t.dtype)
    if code is None:
        raise SsdHipError("images are uint8, float32 or float64")
    return code


def image_program(images, ops, args, out_dtype):
    """Run per-image pointwise programs (ssdhip_image_program): images (B, H, W, 3) CUDA uint8 | float32 | float64 contiguous; ops (B, 16) int32,
    args (B, 16) float64 (host arrays or CUDA tensors); out_dtype a torch dtype (what the programs end in).  Returns (B, H, W, 3)."""
    torch = _torch()
    lib = _image_lib()
    require_cuda(images, "images")
    if images.dim() != 4 or images.shape[3] != 3 or not images.is_contiguous():
        raise SsdHipError("images must be a contiguous (B, H, W, 3) tensor")
    b, h, w, _ = images.shape
    ops = to_device(ops, device=images.device, dtype=torch.int32).contiguous()
    args = to_device(args, device=images.device, dtype=torch.float64).contiguous()
    if tuple(ops.shape) != (b, IMG_PROG) or tuple(args.shape) != (b, IMG_PROG):
        raise SsdHipError("ops / args must be (%d, %d)" % (b, IMG_PROG))
    out = torch.empty((b, h, w, 3), dtype=out_dtype, device=images.device)
    in_code = _img_dtype_code(images)
    with torch.cuda.device(images.device):
        rc = lib.ssdhip_image_program(_ptr(images), in_code, _ptr(out), _img_dtype_code(out), b, h * w, _ptr(ops), _ptr(args),
                                      current_stream_ptr(images.device))
    check(rc, "ssdhip_image_program")
    return out


def image_resize_u8(images, out_h, out_w, ix, wx, iy, wy):
    """ssdhip_image_resize_u8: images (B, H, W, C) CUDA uint8; ix / wx (out_w, nx), iy / wy (out_h, ny) tap tables (int32 / float64)."""
    torch = _torch()
    lib = _image_lib()
    require_cuda(images, "images")
    if images.dtype != torch.uint8 or images.dim() != 4 or not images.is_contiguous():
        raise SsdHipError("images must be a contiguous (B, H, W, C) uint8 tensor")
    b, h, w, c = images.shape
    dev = images.device
    ix = to_device(ix, device=dev, dtype=torch.int32).contiguous()
    wx = to_device(wx, device=dev, dtype=torch.float64).contiguous()
    iy = to_device(iy, device=dev, dtype=torch.int32).contiguous()
    wy = to_device(wy, device=dev, dtype=torch.float64).contiguous()
    if ix.shape != wx.shape or iy.shape != wy.shape or ix.shape[0] != out_w or iy.shape[0] != out_h:
        raise SsdHipError("tap tables must be (out_w, nx) and (out_h, ny)")
    out = torch.empty((b, out_h, out_w, c), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.ssdhip_image_resize_u8(_ptr(images), _ptr(out), b, h, w, out_h, out_w, c, _ptr(ix), _ptr(wx), int(ix.shape[1]), _ptr(iy),
                                        _ptr(wy), int(iy.shape[1]), current_stream_ptr(dev))
    check(rc, "ssdhip_image_resize_u8")
    return out


def image_resize_gather_u8(images, out_h, out_w, ix, wx, iy, wy, background):
    """ssdhip_image_resize_gather_u8: images (B, H, W, C) CUDA uint8; per-image tap tables ix / wx (B, out_w, nx), iy / wy (B, out_h, ny)
    (index -1 = background), background (B, C) uint8.  Returns the (B, out_h, out_w, C) uint8 batch."""
    torch = _torch()
    lib = _image_lib()
    if not getattr(lib, "_gather_bound", False):
        vp, ci = ctypes.c_void_p, ctypes.c_int
        lib.ssdhip_image_resize_gather_u8.restype = ci
        lib.ssdhip_image_resize_gather_u8.argtypes = [vp, vp] + [ci] * 6 + [vp, vp, ci, vp, vp, ci, vp, vp]
        lib._gather_bound = True
    require_cuda(images, "images")
    if images.dtype != torch.uint8 or images.dim() != 4 or not images.is_contiguous():
        raise SsdHipError("images must be a contiguous (B, H, W, C) uint8 tensor")
    b, h, w, c = images.shape
    dev = images.device
    ix = to_device(ix, device=dev, dtype=torch.int32).contiguous()
    wx = to_device(wx, device=dev, dtype=torch.float64).contiguous()
    iy = to_device(iy, device=dev, dtype=torch.int32).contiguous()
    wy = to_device(wy, device=dev, dtype=torch.float64).contiguous()
    bg = to_device(background, device=dev, dtype=torch.uint8).contiguous()
    if (ix.shape != wx.shape or iy.shape != wy.shape or tuple(ix.shape[:2]) != (b, out_w) or tuple(iy.shape[:2]) != (b, out_h)
            or tuple(bg.shape) != (b, c)):
        raise SsdHipError("tap tables must be (B, out_w, nx) / (B, out_h, ny), background (B, C)")
    out = torch.empty((b, out_h, out_w, c), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.ssdhip_image_resize_gather_u8(_ptr(images), _ptr(out), b, h, w, out_h, out_w, c, _ptr(ix), _ptr(wx), int(ix.shape[2]), _ptr(iy),
                                               _ptr(wy), int(iy.shape[2]), _ptr(bg), current_stream_ptr(dev))
    check(rc, "ssdhip_image_resize_gather_u8")
    return out


def image_resize_cv_u8(images, out_h, out_w, kind, area, ix, wx, iy, wy):
    """ssdhip_image_resize_cv_u8 (cv2.resize's own 8-bit arithmetic): images (B, H, W, C) CUDA uint8; one plan for the batch
    (data_generator/_image_ops.resize_plan): `kind`, `area`, ix / wx (out_w, nx), iy / wy (out_h, ny)."""
    torch = _torch()
    lib = _image_lib()
    if not getattr(lib, "_resize_cv_bound", False):
        vp, ci = ctypes.c_void_p, ctypes.c_int
        lib.ssdhip_image_resize_cv_u8.restype = ci
        lib.ssdhip_image_resize_cv_u8.argtypes = [vp, vp] + [ci] * 8 + [vp, vp, ci, vp, vp, ci, vp]
        lib._resize_cv_bound = True
    require_cuda(images, "images")
    if images.dtype != torch.uint8 or images.dim() != 4 or not images.is_contiguous():
        raise SsdHipError("images must be a contiguous (B, H, W, C) uint8 tensor")
    b, h, w, c = images.shape
    dev = images.device
    ix = to_device(ix, device=dev, dtype=torch.int32).contiguous()
    wx = to_device(wx, device=dev, dtype=torch.float64).contiguous()
    iy = to_device(iy, device=dev, dtype=torch.int32).contiguous()
    wy = to_device(wy, device=dev, dtype=torch.float64).contiguous()
    if ix.shape != wx.shape or iy.shape != wy.shape or ix.shape[0] != out_w or iy.shape[0] != out_h:
        raise SsdHipError("tap tables must be (out_w, nx) and (out_h, ny)")
    out = torch.empty((b, out_h, out_w, c), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.ssdhip_image_resize_cv_u8(_ptr(images), _ptr(out), b, h, w, out_h, out_w, c, int(kind), int(area), _ptr(ix), _ptr(wx),
                                           int(ix.shape[1]), _ptr(iy), _ptr(wy), int(iy.shape[1]), current_stream_ptr(dev))
    check(rc, "ssdhip_image_resize_cv_u8")
    return out


def image_resize_gather_cv_u8(images, out_h, out_w, plans, ix, wx, iy, wy, background):
    """ssdhip_image_resize_gather_cv_u8: images (B, H, W, C) CUDA uint8; plans (B, 4) int32 [kind, area, taps per column, taps per row];
    per-image tables ix / wx (B, out_w, nx), iy / wy (B, out_h, ny) (index -1 = background), background (B, C) uint8."""
    torch = _torch()
    lib = _image_lib()
    if not getattr(lib, "_gather_cv_bound", False):
        vp, ci = ctypes.c_void_p, ctypes.c_int
        lib.ssdhip_image_resize_gather_cv_u8.restype = ci
        lib.ssdhip_image_resize_gather_cv_u8.argtypes = [vp, vp] + [ci] * 6 + [vp, vp, vp, ci, vp, vp, ci, vp, vp]
        lib._gather_cv_bound = True
    require_cuda(images, "images")
    if images.dtype != torch.uint8 or images.dim() != 4 or not images.is_contiguous():
        raise SsdHipError("images must be a contiguous (B, H, W, C) uint8 tensor")
    b, h, w, c = images.shape
    dev = images.device
    plans = to_device(plans, device=dev, dtype=torch.int32).contiguous()
    ix = to_device(ix, device=dev, dtype=torch.int32).contiguous()
    wx = to_device(wx, device=dev, dtype=torch.float64).contiguous()
    iy = to_device(iy, device=dev, dtype=torch.int32).contiguous()
    wy = to_device(wy, device=dev, dtype=torch.float64).contiguous()
    bg = to_device(background, device=dev, dtype=torch.uint8).contiguous()
    if ix.shape[2] < 2:                                        # (the entry point wants room for the linear kernel's two taps)
        pad = lambda t: torch.cat([t, torch.zeros_like(t)], dim=2)
        ix, wx = pad(ix), pad(wx)
    if iy.shape[2] < 2:
        pad = lambda t: torch.cat([t, torch.zeros_like(t)], dim=2)
        iy, wy = pad(iy), pad(wy)
    if (ix.shape != wx.shape or iy.shape != wy.shape or tuple(ix.shape[:2]) != (b, out_w) or tuple(iy.shape[:2]) != (b, out_h)
            or tuple(bg.shape) != (b, c) or tuple(plans.shape) != (b, 4)):
        raise SsdHipError("plans must be (B, 4), tap tables (B, out_w, nx) / (B, out_h, ny), background (B, C)")
    out = torch.empty((b, out_h, out_w, c), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.ssdhip_image_resize_gather_cv_u8(_ptr(images), _ptr(out), b, h, w, out_h, out_w, c, _ptr(plans), _ptr(ix), _ptr(wx),
                                                  int(ix.shape[2]), _ptr(iy), _ptr(wy), int(iy.shape[2]), _ptr(bg), current_stream_ptr(dev))
    check(rc, "ssdhip_image_resize_gather_cv_u8")
    return out


def image_hist_u8(image, channel):
    """256-bin histogram (CUDA int64 tensor) of one channel of an (..., C) uint8 CUDA image."""
    torch = _torch()
    lib = _image_lib()
    require_cuda(image, "image")
    if image.dtype != torch.uint8 or not image.is_contiguous():
        raise SsdHipError("image must be contiguous uint8")
    c = int(image.shape[-1])
    hist = torch.empty((256,), dtype=torch.int32, device=image.device)
    with torch.cuda.device(image.device):
        rc = lib.ssdhip_image_hist_u8(_ptr(image), image.numel() // c, c, int(channel), _ptr(hist), current_stream_ptr(image.device))
    check(rc, "ssdhip_image_hist_u8")
    return hist.to(torch.int64)


def image_lut_u8(image, table, channel_mask):
    """table[image] on the channels of channel_mask (bit c = channel c), the other channels copied: (..., C) uint8 CUDA image."""
    torch = _torch()
    lib = _image_lib()
    require_cuda(image, "image")
    if image.dtype != torch.uint8 or not image.is_contiguous():
        raise SsdHipError("image must be contiguous uint8")
    table = to_device(table, device=image.device, dtype=torch.uint8).contiguous()
    if table.numel() != 256:
        raise SsdHipError("the table has 256 entries")
    out = torch.empty_like(image)
    with torch.cuda.device(image.device):
        rc = lib.ssdhip_image_lut_u8(_ptr(image), _ptr(out), image.numel(), int(image.shape[-1]), int(channel_mask), _ptr(table),
                                     current_stream_ptr(image.device))
    check(rc, "ssdhip_image_lut_u8")
    return out


# ---- the decisions of the original-SSD augmentation chain for a whole batch (csrc/ssdhip_augment.hip) ---------------------------------
class _AugParams(ctypes.Structure):                  # struct ssdhip_augment_params (include/ssdhip.h)
    _fields_ = [("img_height", ctypes.c_int), ("img_width", ctypes.c_int),
                ("expand_prob", ctypes.c_double), ("expand_min_scale", ctypes.c_double), ("expand_max_scale", ctypes.c_double),
                ("crop_prob", ctypes.c_double), ("crop_min_scale", ctypes.c_double), ("crop_max_scale", ctypes.c_double),
                ("crop_min_aspect_ratio", ctypes.c_double), ("crop_max_aspect_ratio", ctypes.c_double),
                ("n_trials", ctypes.c_int), ("n_bounds", ctypes.c_int),
                ("bound_cdf", ctypes.c_double * 8), ("bound_lower", ctypes.c_double * 8), ("bound_upper", ctypes.c_double * 8),
                ("flip_prob", ctypes.c_double),
                ("n_modes", ctypes.c_int), ("interpolation_modes", ctypes.c_int * 8), ("out_height", ctypes.c_int), ("out_width", ctypes.c_int),
                ("max_rounds", ctypes.c_int)]


AUG_MAX_BOXES = 64


def ssd_augment_decide(params, mt_states, labels, n_labels, device):
    """`ssdhip_ssd_augment_decide`: params a dict of the fields of ssdhip_augment_params; mt_states (B, 625) uint32, labels (B, 64, 5)
    float64, n_labels (B,) int32 NumPy arrays (ONE upload) -> (geometry (B, 12) int32 CUDA tensor, fetch) where fetch() downloads
    (geometry (B, 12) int32, labels_out (B, 64, 5) float64, n_out (B,) int32, mt_states_out (B, 625) uint32) as NumPy arrays."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_aug_bound", False):
        lib.ssdhip_ssd_augment_decide.restype = ctypes.c_int
        lib.ssdhip_ssd_augment_decide.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 8
        lib._aug_bound = True
    q = _AugParams()
    for k, v in params.items():
        if isinstance(v, (list, tuple)):
            arr = getattr(q, k)
            for i, e in enumerate(v):
                arr[i] = e
        else:
            setattr(q, k, v)
    B = int(mt_states.shape[0])
    if mt_states.shape != (B, 625) or labels.shape != (B, AUG_MAX_BOXES, 5) or n_labels.shape != (B,):
        raise SsdHipError("ssd_augment_decide: mt_states (B, 625), labels (B, 64, 5), n_labels (B,)")
    # one packed upload: [labels f64 | mt u32 | n i32], one packed download: [labels_out f64 | mt_out u32 | geometry i32 | n_out i32]
    nl, nm = B * AUG_MAX_BOXES * 5 * 8, B * 625 * 4
    host = np.empty((nl + nm + B * 4,), dtype=np.uint8)
    host[:nl] = np.ascontiguousarray(labels, dtype=np.float64).view(np.uint8).ravel()
    host[nl:nl + nm] = np.ascontiguousarray(mt_states, dtype=np.uint32).view(np.uint8).ravel()
    host[nl + nm:] = np.ascontiguousarray(n_labels, dtype=np.int32).view(np.uint8).ravel()
    dev_in = torch.from_numpy(host).to(device)
    no = nl + nm + B * 12 * 4 + B * 4
    dev_out = torch.empty((no,), dtype=torch.uint8, device=device)
    base_in, base_out = dev_in.data_ptr(), dev_out.data_ptr()
    vp = ctypes.c_void_p
    with torch.cuda.device(device):
        rc = lib.ssdhip_ssd_augment_decide(ctypes.byref(q), B, vp(base_in + nl), vp(base_in), vp(base_in + nl + nm), vp(base_out + nl + nm),
                                           vp(base_out), vp(base_out + nl + nm + B * 48), vp(base_out + nl), current_stream_ptr(device))
    check(rc, "ssdhip_ssd_augment_decide")
    geo_dev = dev_out[nl + nm:nl + nm + B * 48].view(torch.int32).view(B, 12)      # stays on the device for augment_taps

    def fetch():
        """(geometry, labels_out, n_out, mt_states_out) as NumPy arrays: ONE download (a host synchronisation: call it last)."""
        out = dev_out.cpu().numpy()
        return (out[nl + nm:nl + nm + B * 48].view(np.int32).reshape(B, 12), out[:nl].view(np.float64).reshape(B, AUG_MAX_BOXES, 5),
                out[nl + nm + B * 48:].view(np.int32), out[nl:nl + nm].view(np.uint32).reshape(B, 625))
    return geo_dev, fetch


class _AugPhoto(ctypes.Structure):
    _fields_ = [("prob", ctypes.c_double * 4), ("lower", ctypes.c_double * 4), ("upper", ctypes.c_double * 4), ("swap_prob", ctypes.c_double)]


def ssd_augment_decide_stream(params, photo, mt_state, labels, n_labels, device):
    """`ssdhip_ssd_augment_decide_stream`: the whole batch on ONE generator stream, photometric decisions included (round 6).  params as
    ssd_augment_decide; photo = dict(prob, lower, upper: four values each for brightness / contrast / saturation / hue, swap_prob);
    mt_state (625,) uint32 (np.random.get_state(): the 624 key words + the position), labels (B, 64, 5) float64, n_labels (B,) int32.
    Returns (ops (B, 16) int32 and args (B, 16) float64 CUDA tensors = the programs of image_program, geometry (B, 12) int32 CUDA tensor,
    fetch) where fetch() downloads (geometry, labels_out, n_out, mt_state_out (625,)) as NumPy arrays."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_augs_bound", False):
        lib.ssdhip_ssd_augment_decide_stream.restype = ctypes.c_int
        lib.ssdhip_ssd_augment_decide_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 10
        lib._augs_bound = True
    q = _AugParams()
    for k, v in params.items():
        if isinstance(v, (list, tuple)):
            arr = getattr(q, k)
            for i, e in enumerate(v):
                arr[i] = e
        else:
            setattr(q, k, v)
    ph = _AugPhoto()
    for i in range(4):
        ph.prob[i], ph.lower[i], ph.upper[i] = float(photo["prob"][i]), float(photo["lower"][i]), float(photo["upper"][i])
    ph.swap_prob = float(photo["swap_prob"])
    B = int(labels.shape[0])
    if mt_state.shape != (625,) or labels.shape != (B, AUG_MAX_BOXES, 5) or n_labels.shape != (B,):
        raise SsdHipError("ssd_augment_decide_stream: mt_state (625,), labels (B, 64, 5), n_labels (B,)")
    # one packed upload: [labels f64 | mt u32 (625, padded to 632) | n i32];
    # one packed download: [labels_out f64 | args f64 | mt_out u32 (632) | geometry i32 | ops i32 | n_out i32]
    nl, nm = B * AUG_MAX_BOXES * 5 * 8, 632 * 4
    host = np.zeros((nl + nm + B * 4,), dtype=np.uint8)
    host[:nl] = np.ascontiguousarray(labels, dtype=np.float64).view(np.uint8).ravel()
    host[nl:nl + 625 * 4] = np.ascontiguousarray(mt_state, dtype=np.uint32).view(np.uint8).ravel()
    host[nl + nm:] = np.ascontiguousarray(n_labels, dtype=np.int32).view(np.uint8).ravel()
    dev_in = torch.from_numpy(host).to(device)
    na = B * IMG_PROG * 8
    o_args, o_mt, o_geo, o_ops, o_n = nl, nl + na, nl + na + nm, nl + na + nm + B * 48, nl + na + nm + B * 48 + B * IMG_PROG * 4
    dev_out = torch.empty((o_n + B * 4,), dtype=torch.uint8, device=device)
    base_in, base_out = dev_in.data_ptr(), dev_out.data_ptr()
    vp = ctypes.c_void_p
    with torch.cuda.device(device):
        rc = lib.ssdhip_ssd_augment_decide_stream(ctypes.byref(q), ctypes.byref(ph), B, vp(base_in + nl), vp(base_in), vp(base_in + nl + nm),
                                                  vp(base_out + o_ops), vp(base_out + o_args), vp(base_out + o_geo), vp(base_out),
                                                  vp(base_out + o_n), vp(base_out + o_mt), current_stream_ptr(device))
    check(rc, "ssdhip_ssd_augment_decide_stream")
    geo_dev = dev_out[o_geo:o_geo + B * 48].view(torch.int32).view(B, 12)
    ops_dev = dev_out[o_ops:o_ops + B * IMG_PROG * 4].view(torch.int32).view(B, IMG_PROG)
    args_dev = dev_out[o_args:o_args + na].view(torch.float64).view(B, IMG_PROG)

    def fetch():
        """(geometry, labels_out, n_out, mt_state_out) as NumPy arrays: ONE download (a host synchronisation: call it last)."""
        out = dev_out.cpu().numpy()
        return (out[o_geo:o_geo + B * 48].view(np.int32).reshape(B, 12), out[:nl].view(np.float64).reshape(B, AUG_MAX_BOXES, 5),
                out[o_n:].view(np.int32), out[o_mt:o_mt + 625 * 4].view(np.uint32).copy())
    return ops_dev, args_dev, geo_dev, fetch


def augment_plans(geo_dev, H, W, out_h, out_w, n_taps):
    """`ssdhip_augment_plans`: the gather launch's per-image plans (plans (B, 4) int32, ix, wx, iy, wy: CUDA tensors
    (B, out_w | out_h, n_taps)) with cv2.resize's own arithmetic, built on the device from the geometry the decision kernels left there."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_augplans_bound", False):
        lib.ssdhip_augment_plans.restype = ctypes.c_int
        lib.ssdhip_augment_plans.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 6
        lib._augplans_bound = True
    B, dev = int(geo_dev.shape[0]), geo_dev.device
    plans = torch.empty((B, 4), dtype=torch.int32, device=dev)
    ix = torch.empty((B, out_w, n_taps), dtype=torch.int32, device=dev)
    wx = torch.empty((B, out_w, n_taps), dtype=torch.float64, device=dev)
    iy = torch.empty((B, out_h, n_taps), dtype=torch.int32, device=dev)
    wy = torch.empty((B, out_h, n_taps), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.ssdhip_augment_plans(_ptr(geo_dev), B, int(H), int(W), int(out_h), int(out_w), int(n_taps), _ptr(plans), _ptr(ix), _ptr(wx),
                                      _ptr(iy), _ptr(wy), current_stream_ptr(dev))
    check(rc, "ssdhip_augment_plans")
    return plans, ix, wx, iy, wy


def augment_taps(geo_dev, H, W, out_h, out_w, n_taps):
    """`ssdhip_augment_taps`: the gather launch's tap tables (ix, wx, iy, wy: CUDA tensors (B, out_w | out_h, n_taps)) built on the device
    from the geometry ssd_augment_decide left there."""
    torch = _torch()
    lib = load()
    if not getattr(lib, "_augtaps_bound", False):
        lib.ssdhip_augment_taps.restype = ctypes.c_int
        lib.ssdhip_augment_taps.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 6 + [ctypes.c_void_p] * 5
        lib._augtaps_bound = True
    B, dev = int(geo_dev.shape[0]), geo_dev.device
    ix = torch.empty((B, out_w, n_taps), dtype=torch.int32, device=dev)
    wx = torch.empty((B, out_w, n_taps), dtype=torch.float64, device=dev)
    iy = torch.empty((B, out_h, n_taps), dtype=torch.int32, device=dev)
    wy = torch.empty((B, out_h, n_taps), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib.ssdhip_augment_taps(_ptr(geo_dev), B, int(H), int(W), int(out_h), int(out_w), int(n_taps), _ptr(ix), _ptr(wx), _ptr(iy),
                                      _ptr(wy), current_stream_ptr(dev)), "ssdhip_augment_taps")
    return ix, wx, iy, wy
