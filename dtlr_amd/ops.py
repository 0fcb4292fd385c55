"""Device operators of the DTLR forward.  Every function takes/returns CUDA tensors (device
buffers) and enqueues on the current stream.

Each operator is the seam behind which a hand-written gfx950 kernel sits (libdtlr_hip.so through
dtlr_amd._lib).  Operators that do not have their HIP kernel yet are listed in `LIBRARY_BACKED`
and call the ROCm libraries through torch (rocBLAS/hipBLASLt GEMM, MIOpen conv) -- still GPU-only:
nothing here runs on the CPU and nothing imports the oracle.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch
import torch.nn.functional as F

from . import MultiScaleDeformableAttention as _msda
from . import _lib

# Operators served by ROCm libraries through torch: NONE.  Every GEMM / convolution / selection of the path runs a kernel of
# libdtlr_hip.so; a shape a kernel does not take RAISES (DTLRError) instead of silently dropping to hipBLASLt / MIOpen, so a
# checkpoint with an unusual head size can never bench on a library without anyone noticing.  bench.py prints this set.
LIBRARY_BACKED: set = set()

_DT = {torch.float32: _lib.DTLR_F32, torch.bfloat16: _lib.DTLR_BF16, torch.float64: _lib.DTLR_F64, torch.float16: _lib.DTLR_F16}
# The two 16-bit storage / MFMA-operand formats.  bf16 tensors are served by libdtlr_hip.so, fp16 tensors by libdtlr_hip_f16.so (the
# same sources compiled with fp16 as "the" 16-bit format, csrc/dtlr_common.h): an operator picks the library from its operands.
H16 = (torch.bfloat16, torch.float16)


class SplitWeight(torch.Tensor):
    """An fp32-shaped tensor whose BYTES are the split slab image of an fp32 weight (`split_pack` / dtlr_split_pack_weights: per 32-k
    slab of a row, fp16 hi halves then fp16 lo halves).  `linear`, `linear_rowmax` and `conv2d_nhwc` recognise it and run the DTLR_F32S
    kernels (fp32 activations, three fp16 MFMAs per product: fp32-grade results at 16/3 x the rate of the exact-fp32 MFMA).  Views and
    row slices of an image are images of the same rows (the default __torch_function__ keeps the subclass); its VALUES mean nothing to torch."""


def split_pack(w):
    """fp32 weight [N, K] (nn.Linear) or [Cout, KH, KW, Cin] (NHWC convolution), K (resp. KH*KW*Cin) a multiple of 32, on the GPU ->
    SplitWeight of the same shape (dtlr_split_pack_weights)."""
    require_cuda(w, "w")
    w = w.detach().float().contiguous()
    rows = int(w.shape[0])
    K = w.numel() // rows
    if K % 32:
        raise _lib.DTLRError(f"ops.split_pack: K = {K} is not a multiple of 32")
    out = torch.empty_like(w)
    code = _lib.lib().dtlr_split_pack_weights(w.data_ptr(), out.data_ptr(), rows, K, _lib.current_stream())
    _lib.check(code, "dtlr_split_pack_weights")
    return out.as_subclass(SplitWeight)


def _in_dt(x, w):
    """dtype code of a GEMM's operands: DTLR_F32S when the weight is a split image (the activations are then fp32)."""
    if isinstance(w, SplitWeight):
        if x.dtype != torch.float32:
            raise _lib.DTLRError("a SplitWeight multiplies fp32 activations")
        return _lib.DTLR_F32S
    return _DT[x.dtype]


def _gkind(x, w):
    return "gemm_bf16" if x.dtype in H16 else ("gemm_f32s" if isinstance(w, SplitWeight) else "gemm_f32")


def _L(*ts):
    """the library that serves these tensors / dtypes: the fp16 build if any of them is fp16, else the bf16 build."""
    for t in ts:
        if (t.dtype if torch.is_tensor(t) else t) == torch.float16:
            return _lib.lib(torch.float16)
    return _lib.lib()


def _hdt(w):
    """16-bit dtype a weight is packed in: its own if it already is one (the engine stores weights in its format), else bf16."""
    return w.dtype if w.dtype in H16 else torch.bfloat16


def require_cuda(t: torch.Tensor, what: str = "input") -> None:
    if not t.is_cuda:
        raise RuntimeError(f"dtlr_amd: {what} must live on the GPU (no CPU path; got device {t.device})")
    _lib.lib()      # raises if libdtlr_hip.so is missing


def workspace_reserve(dtype, nbytes: int = 128 << 20) -> None:
    """Pre-size the current stream's scratch workspace of the library serving `dtype` (split-K partial tiles, hidden-split FFN parts:
    include/dtlr_hip.h, dtlr_workspace_reserve) -- outside stream capture, so that a later captured forward allocates nothing and takes the
    same kernels as an eager one.  128 MiB covers every shape of a 32-line batch (the largest user: 8 parts x 12288 rows x 1 KB)."""
    _lib.check(_L(dtype).dtlr_workspace_reserve(int(nbytes), _lib.current_stream()), "dtlr_workspace_reserve")


def workspace_retired_bytes(dtype=None) -> int:
    """bytes of replaced workspace buffers the library keeps allocated (a captured graph may still hold their addresses)"""
    return int(_L(dtype if dtype is not None else torch.float32).dtlr_workspace_retired_bytes())


# --------------------------------------------------------------------------------------------
# bench.py sets this to a list to time every MFMA-class launch (GEMM / implicit-GEMM conv / fused FFN) with HIP events on
# the launch stream: entries are (start_event, end_event, kind, flops).
MFMA_EVENTS = None
MFMA_EVENTS_MIN_FLOPS = 2.0e9     # only launches this large are timed: an event pair costs a few microseconds of stream time


class _Timed:
    """with _Timed(kind, flops): <launch>  -- records a HIP event pair around the launch when MFMA_EVENTS is a list."""
    __slots__ = ("kind", "flops", "nbytes", "tag", "symbol", "a", "st")

    def __init__(self, kind, flops, nbytes=0.0, tag=None, symbol=None):
        # symbol = the ONE kernel this launch runs (as rocprofv3 names it), when the call is a single kernel: bench.py keys
        # roofline_by_kernel by it so that each row can be recomputed from profiles/*_kernel_stats.csv
        self.kind, self.flops, self.nbytes, self.tag, self.symbol = kind, flops, nbytes, tag, symbol

    def __enter__(self):
        self.a = None
        if MFMA_EVENTS is not None and self.flops >= MFMA_EVENTS_MIN_FLOPS:
            self.st = torch.cuda.current_stream()
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record(self.st)
        return self

    def __exit__(self, *exc):
        if self.a is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record(self.st)
            MFMA_EVENTS.append((self.a, b, self.kind, self.flops, self.nbytes, self.tag, self.symbol))
        return False


def linear(x, w, b=None, relu=False, residual=None, a2=None, row_mask=None, out_dtype=None):
    """y = epilogue((x [+ a2]) @ w.T): + b -> ReLU (relu=True/1) or exact GELU (relu=3) -> zero rows where row_mask -> + residual
    -> ReLU (relu=2, the ResNet bottleneck tail).
    x [..., K] and w [N, K] share a dtype (bf16 or fp32); b fp32 [N]; row_mask bool [...];
    residual [..., N] in the output dtype.  HIP MFMA kernel (dtlr_gemm_nt); no library fallback
    except shapes the kernel does not take (K not a multiple of the 128-byte slab)."""
    K = x.shape[-1]
    N = w.shape[0]
    out_dtype = out_dtype or x.dtype
    slab = 64 if x.dtype in H16 else 32
    if a2 is not None and a2.numel() != x.numel():
        # row-broadcast A2 ([S, K] against [B, S, K]: the position embedding of an unpadded batch): its own entry point
        if not (x.dtype in H16 + (torch.float32,) and w.dtype == x.dtype and a2.dtype == x.dtype and out_dtype == x.dtype and K % slab == 0
                and not relu and residual is None and row_mask is None and a2.shape[-1] == K and (x.numel() // K) % (a2.numel() // K) == 0):
            raise _lib.DTLRError("ops.linear: a row-broadcast a2 needs same-dtype operands, K a slab multiple, M a multiple of a2's rows, "
                                 "and no ReLU / residual / row mask")
        x = x if x.is_contiguous() else x.contiguous()
        a2 = a2 if a2.is_contiguous() else a2.contiguous()
        if b is not None and b.dtype != torch.float32:
            b = b.float()
        M, R2 = x.numel() // K, a2.numel() // K
        y = torch.empty(x.shape[:-1] + (N,), dtype=out_dtype, device=x.device)
        es = x.element_size()
        with _Timed(_gkind(x, w), 2.0 * M * N * K,
                    float(M) * K * es + float(R2) * K * es + float(N) * K * es + float(M) * N * es, f"linear M{M} N{N} K{K}+a2bcast{R2}"):
            code = _L(x).dtlr_gemm_nt_a2bcast(x.data_ptr(), a2.data_ptr(), R2, w.data_ptr(), 0 if b is None else b.data_ptr(), y.data_ptr(),
                                                   M, N, K, _in_dt(x, w), _lib.current_stream())
        _lib.check(code, "dtlr_gemm_nt_a2bcast")
        return y
    if x.dtype in H16 + (torch.float32,) and w.dtype == x.dtype and K % slab == 0 \
            and (x.dtype in H16 or out_dtype == torch.float32):
        x = x if x.is_contiguous() else x.contiguous()
        if a2 is not None and not a2.is_contiguous():
            a2 = a2.contiguous()
        if residual is not None:
            residual = residual if residual.is_contiguous() else residual.contiguous()
            assert residual.dtype == out_dtype
        if b is not None and b.dtype != torch.float32:
            b = b.float()
        M = x.numel() // K
        y = torch.empty(x.shape[:-1] + (N,), dtype=out_dtype, device=x.device)
        es, eo = x.element_size(), (2 if out_dtype in H16 else 4)
        nbytes = float(M) * K * es * (2 if a2 is not None else 1) + float(N) * K * es + float(M) * N * eo * (2 if residual is not None else 1)
        tag = f"linear M{M} N{N} K{K}" + ("+a2" if a2 is not None else "") + ("+res" if residual is not None else "") + ("" if out_dtype == x.dtype else "->f32")
        with _Timed(_gkind(x, w), 2.0 * M * N * K, nbytes, tag):
            code = _L(x).dtlr_gemm_nt(x.data_ptr(), 0 if a2 is None else a2.data_ptr(), w.data_ptr(),
                                           0 if b is None else b.data_ptr(), 0 if residual is None else residual.data_ptr(),
                                           0 if row_mask is None else row_mask.data_ptr(), y.data_ptr(),
                                           M, N, K, int(relu), _in_dt(x, w), _DT[out_dtype], _lib.current_stream())
        _lib.check(code, "dtlr_gemm_nt")
        return y
    raise _lib.DTLRError(f"ops.linear: no HIP kernel for x {tuple(x.shape)} {x.dtype} @ w {tuple(w.shape)} {w.dtype} -> {out_dtype} "
                         f"(K must be a multiple of {slab} elements; fp32 operands give fp32 results); there is no library fallback")


def linear_resbcast(x, w, resid, b=None):
    """x @ w.T [+ b] + resid[row % R]: `linear` with a row-BROADCAST residual (dtlr_gemm_nt_resbcast).  x [..., K] (M rows in all), w [N, K]
    (same dtype, or a SplitWeight for fp32 x), resid [R, N] in the output dtype with M % R == 0.  The encoder's [offsets | logits] projection of
    an unpadded batch: (src + pos) W^T + b = src W^T + (pos W^T + b), the second term one [S, N] matrix for every image."""
    require_cuda(x, "x")
    K, N = x.shape[-1], w.shape[0]
    M, R = x.numel() // K, resid.numel() // N
    slab = 64 if x.dtype in H16 else 32
    if x.dtype not in H16 + (torch.float32,) or w.dtype != x.dtype or resid.dtype != x.dtype or K % slab or M % R or resid.shape[-1] != N:
        raise _lib.DTLRError(f"ops.linear_resbcast: no HIP kernel for x {tuple(x.shape)} {x.dtype}, w {tuple(w.shape)}, resid {tuple(resid.shape)} {resid.dtype}")
    x = x if x.is_contiguous() else x.contiguous()
    resid = resid if resid.is_contiguous() else resid.contiguous()
    y = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
    es = x.element_size()
    with _Timed(_gkind(x, w), 2.0 * M * N * K, float(M) * K * es + float(N) * K * es + float(M) * N * es + float(R) * N * es, f"linear M{M} N{N} K{K}+resb{R}"):
        code = _L(x).dtlr_gemm_nt_resbcast(x.data_ptr(), w.data_ptr(), 0 if b is None else b.data_ptr(), resid.data_ptr(), R, y.data_ptr(),
                                           M, N, K, _in_dt(x, w), _lib.current_stream())
    _lib.check(code, "dtlr_gemm_nt_resbcast")
    return y


def k256_pack(w):
    """[N, 256] weight (N = 256 or 384; any float dtype / device) -> the fragment-order bf16 image dtlr_gemm_k256 keeps in
    registers: block ((wave * NRT + rt) * 8 + ks) = 64 lanes x 8 elements, lane (m, g) <- W[(wave*NRT + rt)*16 + m][32 ks + 8 g ..]
    (== dtlr_k256_pack_weights; done with tensor ops on the weight's own device)."""
    N, K = w.shape
    assert K == 256 and N in (256, 384)
    nrt = N // 128
    return w.detach().to(_hdt(w)).view(8, nrt, 16, 8, 4, 8).permute(0, 1, 3, 4, 2, 5).contiguous().view(-1)


def gemm_k256(x, wp, n_out: int, b=None, resid=None, row_mask=None, out=None):
    """Weight-resident streaming projection (dtlr_gemm_k256): y = x @ W.T + b [+ resid[m % rows]] with padded rows zeroed.
    x [..., 256] bf16 contiguous; wp = k256_pack(W); resid [rows, n_out] bf16; row_mask [...] bool/uint8;
    out: optional [..., n_out] bf16 view whose last dim is contiguous (a column slice of a wider matrix)."""
    require_cuda(x, "x")
    assert x.dtype in H16 and x.shape[-1] == 256 and wp.dtype in H16 and wp.numel() == n_out * 256
    x = x if x.is_contiguous() else x.contiguous()
    M = x.numel() // 256
    if out is None:
        out = torch.empty(x.shape[:-1] + (n_out,), dtype=x.dtype, device=x.device)
        ldc = n_out
    else:
        assert out.dtype in H16 and out.shape == x.shape[:-1] + (n_out,) and out.stride(-1) == 1
        ldc = out.stride(-2)
        assert all(out.stride(d) == out.stride(d + 1) * out.shape[d + 1] for d in range(out.dim() - 2)), "out: rows must be evenly strided"
    if resid is not None:
        assert resid.dtype in H16 and resid.is_contiguous() and resid.shape[-1] == n_out
    if b is not None and b.dtype != torch.float32:
        b = b.float()
    if row_mask is not None:
        row_mask = row_mask.reshape(-1)
        assert row_mask.numel() == M and row_mask.is_contiguous() and row_mask.dtype in (torch.bool, torch.uint8)
    nbytes = float(M) * 256 * 2 + float(n_out) * 256 * 2 + float(M) * n_out * 2
    with _Timed("gemm_bf16", 2.0 * M * n_out * 256, nbytes, f"k256 M{M} N{n_out} K256" + ("+resb" if resid is not None else "")):
        code = _L(x).dtlr_gemm_k256(x.data_ptr(), wp.data_ptr(), 0 if b is None else b.data_ptr(),
                                         0 if resid is None else resid.data_ptr(), 0 if resid is None else resid.numel() // n_out,
                                         0 if row_mask is None else row_mask.data_ptr(), out.data_ptr(), ldc, M, n_out, _lib.current_stream())
    _lib.check(code, "dtlr_gemm_k256")
    return out


def kres_supported(M: int, N: int, K: int, dtype) -> bool:
    """Shapes dtlr_gemm_kres takes (the HBM-streaming 1x1 convolutions of the ResNet bottlenecks)."""
    return dtype in H16 and K in (64, 128, 256) and (N % 256 == 0 or N in (64, 128, 192)) and N >= 64 and M >= 16384


def kres_pack(w, np_pairs=None):
    """[N, K] weight -> the fragment-order image of dtlr_gemm_kres (== dtlr_gemm_kres_pack_weights): column slices of 256 NP channels,
    block ((((slice 8 + wave) NP + p) 2 + e) KS + ks) lane (m, g) <- W[256 NP slice + 32 (wave NP + p) + 8 (m >> 2) + 4 e + (m & 3)][32 ks + 8 g ..]."""
    N, K = w.shape
    if N in (64, 128, 192):                    # one zero-padded 256-channel column
        w = torch.cat([w.detach().to(_hdt(w)), torch.zeros((256 - N, K), dtype=_hdt(w), device=w.device)])
        N = 256
    assert K in (64, 128, 256, 384) and N % 256 == 0
    NP = np_pairs or (2 if (N % 512 == 0 and K <= 128) else 1)
    ns, KS = N // (256 * NP), K // 32
    # row index = 256 NP sl + 32 (wave NP + p) + 8 mh + 4 e + ml  with m = 4 mh + ml ; column = 32 ks + 8 g + x
    v = w.detach().to(_hdt(w)).view(ns, 8, NP, 4, 2, 4, KS, 4, 8)          # sl, wave, p, mh, e, ml, ks, g, x
    return v.permute(0, 1, 2, 4, 6, 7, 3, 5, 8).contiguous().view(-1)             # sl, wave, p, e, ks, [g, mh, ml] = lane, x


def gemm_kres(x, wp, n_out: int, b=None, residual=None, relu: bool = False):
    """relu?(x @ W.T + b + residual) with the weight resident in registers and the rows of x / residual streamed through LDS
    (dtlr_gemm_kres); wp = kres_pack(W).  x [..., K] bf16, residual [..., n_out] bf16 or None."""
    require_cuda(x, "x")
    K = x.shape[-1]
    assert x.dtype in H16 and wp.dtype in H16 and wp.numel() == max(n_out, 256) * K
    x = x if x.is_contiguous() else x.contiguous()
    M = x.numel() // K
    if residual is not None:
        assert residual.dtype in H16 and residual.shape[-1] == n_out and residual.numel() == M * n_out
        residual = residual if residual.is_contiguous() else residual.contiguous()
    if b is not None and b.dtype != torch.float32:
        b = b.float()
    out = torch.empty(x.shape[:-1] + (n_out,), dtype=x.dtype, device=x.device)
    nbytes = float(M) * K * 2 + float(n_out) * K * 2 + float(M) * n_out * 2 * (2 if residual is not None else 1)
    with _Timed("gemm_bf16", 2.0 * M * n_out * K, nbytes, f"kres M{M} N{n_out} K{K}" + ("+res" if residual is not None else "")):
        code = _L(x).dtlr_gemm_kres(x.data_ptr(), wp.data_ptr(), 0 if b is None else b.data_ptr(),
                                         0 if residual is None else residual.data_ptr(), out.data_ptr(), M, n_out, K, 1 if relu else 0,
                                         _lib.current_stream())
    _lib.check(code, "dtlr_gemm_kres")
    return out


def gemm_kres_chain(x, wp, b=None, x2=None, residual=None, relu: bool = True, wp2=None, b2=None, n2: int = 0):
    """A layer1 bottleneck tail chained with its neighbours (dtlr_gemm_kres_chain):
        y  = relu?([x | x2] @ W.T + b (+ residual))     x [..., 64]; exactly one of x2 [..., 64] (first bottleneck: the 1x1 shortcut
                                                         convolution as K columns 64..127, W = [W3 | Wd], b = b3 + bd) and residual [..., 256]
        t  = relu(y @ W2.T + b2)                         the next bottleneck's first 1x1 convolution (n2 = 64 or 128), when wp2 is given
    wp = kres_pack(W [256, 64 or 128]), wp2 = kres_pack(W2 [n2, 256]).  Returns (y, t or None)."""
    require_cuda(x, "x")
    assert x.dtype in H16 and wp.dtype == x.dtype and x.shape[-1] == 64 and (x2 is None) != (residual is None)
    x = x if x.is_contiguous() else x.contiguous()
    M = x.numel() // 64
    K = 64
    if x2 is not None:
        assert x2.dtype == x.dtype and x2.shape[-1] == 64 and x2.numel() == x.numel()
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        K = 128
    else:
        assert residual.dtype == x.dtype and residual.shape[-1] == 256 and residual.numel() == M * 256
        residual = residual if residual.is_contiguous() else residual.contiguous()
    assert wp.numel() == 256 * K
    if wp2 is not None:
        assert wp2.dtype == x.dtype and n2 in (64, 128) and wp2.numel() == 256 * 256 and (x2 is None or n2 == 64)
    else:
        assert x2 is not None
    b = None if b is None else (b if b.dtype == torch.float32 else b.float())
    b2 = None if b2 is None else (b2 if b2.dtype == torch.float32 else b2.float())
    y = torch.empty(x.shape[:-1] + (256,), dtype=x.dtype, device=x.device)
    t = torch.empty(x.shape[:-1] + (n2,), dtype=x.dtype, device=x.device) if wp2 is not None else None
    nbytes = float(M) * K * 2 + 256.0 * K * 2 + float(M) * 256 * 2 * (2 if residual is not None else 1) + (float(M) * n2 * 2 + 512.0 * n2 if t is not None else 0.0)
    flops = 2.0 * M * 256 * K + (2.0 * M * n2 * 256 if t is not None else 0.0)
    with _Timed("gemm_bf16", flops, nbytes, f"kres_chain M{M} K{K}" + ("+res" if residual is not None else "+cat") + (f" ->N{n2}" if t is not None else "")):
        code = _L(x).dtlr_gemm_kres_chain(x.data_ptr(), 0 if x2 is None else x2.data_ptr(), wp.data_ptr(), 0 if b is None else b.data_ptr(),
                                          0 if residual is None else residual.data_ptr(), y.data_ptr(), M, 1 if relu else 0,
                                          0 if wp2 is None else wp2.data_ptr(), 0 if b2 is None else b2.data_ptr(),
                                          0 if t is None else t.data_ptr(), n2, _lib.current_stream())
    _lib.check(code, "dtlr_gemm_kres_chain")
    return y, t


def gemm_kres_cat_s2(t, x, wp, b=None, relu: bool = True):
    """layer2's first bottleneck tail with the strided shortcut convolution as extra K columns (dtlr_gemm_kres_cat_s2):
        y[b, i, j] = relu?([t[b, i, j] | x[b, 2 i, 2 j]] @ W.T + b),   W = [W3 | Wd] [512, 384], b = b3 + bd
    t [B, Hout, Wout, 128], x [B, Hin, Win, 256] NHWC 16-bit, Hout = (Hin - 1) // 2 + 1; wp = kres_pack(W).  Returns y [B, Hout, Wout, 512]."""
    require_cuda(t, "t")
    assert t.dtype in H16 and x.dtype == t.dtype and wp.dtype == t.dtype and t.dim() == 4 and x.dim() == 4
    B, Hin, Win, Cx = x.shape
    Hout, Wout = (Hin - 1) // 2 + 1, (Win - 1) // 2 + 1
    assert Cx == 256 and tuple(t.shape) == (B, Hout, Wout, 128) and wp.numel() == 512 * 384
    t = t if t.is_contiguous() else t.contiguous()
    x = x if x.is_contiguous() else x.contiguous()
    b = None if b is None else (b if b.dtype == torch.float32 else b.float())
    y = torch.empty((B, Hout, Wout, 512), dtype=t.dtype, device=t.device)
    M = B * Hout * Wout
    with _Timed("gemm_bf16", 2.0 * M * 512 * 384, float(M) * (128 + 256 + 512) * 2 + 512.0 * 384 * 2, f"kres_cat_s2 M{M} N512 K128+256s2"):
        code = _L(t).dtlr_gemm_kres_cat_s2(t.data_ptr(), x.data_ptr(), wp.data_ptr(), 0 if b is None else b.data_ptr(), y.data_ptr(),
                                           B, Hin, Win, 1 if relu else 0, _lib.current_stream())
    _lib.check(code, "dtlr_gemm_kres_cat_s2")
    return y


def kres_pack_bcast384(w):
    """[384, 256] weight -> the zero-padded 512-channel image of dtlr_gemm_kres_bcast384 (== dtlr_gemm_kres_pack_weights_bcast384)."""
    assert tuple(w.shape) == (384, 256)
    wpad = torch.cat([w.detach().to(_hdt(w)), torch.zeros((128, 256), dtype=_hdt(w), device=w.device)])
    return kres_pack(wpad, np_pairs=2)


def gemm_kres_bcast384(x, wp, resid):
    """x @ W.T + resid[m % rows] for W [384, 256]: the weight-resident streaming kernel with the row-broadcast residual DMA'd through LDS
    (dtlr_gemm_kres_bcast384); wp = kres_pack_bcast384(W); resid [rows, 384] bf16, rows % 64 == 0, M % rows == 0."""
    require_cuda(x, "x")
    assert x.dtype in H16 and x.shape[-1] == 256 and wp.numel() == 512 * 256 and resid.dtype in H16
    assert resid.is_contiguous() and resid.shape[-1] == 384
    x = x if x.is_contiguous() else x.contiguous()
    M = x.numel() // 256
    rows = resid.numel() // 384
    out = torch.empty(x.shape[:-1] + (384,), dtype=x.dtype, device=x.device)
    nbytes = float(M) * 256 * 2 + 384.0 * 256 * 2 + float(M) * 384 * 2
    with _Timed("gemm_bf16", 2.0 * M * 384 * 256, nbytes, f"kres M{M} N384 K256+resb"):
        code = _L(x).dtlr_gemm_kres_bcast384(x.data_ptr(), wp.data_ptr(), resid.data_ptr(), rows, out.data_ptr(), M, _lib.current_stream())
    _lib.check(code, "dtlr_gemm_kres_bcast384")
    return out


def linear_rowmax(x, w, b=None):
    """max over the output channels of (x @ w.T + b), without materialising the product (dtlr_gemm_nt_rowmax: the GEMM's
    row-max epilogue): x [..., K], w [N, K] (same dtype, 16-bit or fp32), b [N] fp32 -> [...] fp32.  The two-stage selection
    only needs this maximum of the class head (deformable_transformer.py:341-345).  x may be the first K columns of wider rows
    (a [..., :K] view of a contiguous [..., lda] tensor): the kernel then walks A with that row stride."""
    require_cuda(x, "x")
    K, N = x.shape[-1], w.shape[0]
    slab = 64 if x.dtype in H16 else 32
    if x.dtype not in H16 + (torch.float32,) or w.dtype != x.dtype or K % slab:
        raise _lib.DTLRError(f"ops.linear_rowmax: no HIP kernel for x {tuple(x.shape)} {x.dtype} @ w {tuple(w.shape)} {w.dtype}")
    lda = K
    if not x.is_contiguous():
        st = x.stride()
        if x.dim() >= 2 and st[-1] == 1 and st[-2] >= K and all(st[d] == st[d + 1] * x.shape[d + 1] for d in range(x.dim() - 2)):
            lda = st[-2]                                  # K leading columns of wider, evenly strided rows
        else:
            x = x.contiguous()
    M = x.numel() // K
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    es = x.element_size()
    with _Timed(_gkind(x, w), 2.0 * M * N * K, float(M) * K * es + float(N) * K * es + 4.0 * M,
                f"rowmax M{M} N{N} K{K}"):
        code = _L(x).dtlr_gemm_nt_rowmax_lda(x.data_ptr(), lda, w.data_ptr(), 0 if b is None else b.data_ptr(), out.data_ptr(),
                                             M, N, K, _in_dt(x, w), _lib.current_stream())
    _lib.check(code, "dtlr_gemm_nt_rowmax_lda")
    return out


def two_stage_gather(om, proposals, idx):
    """The gathers after the two-stage top-k in one launch (dtlr_two_stage_gather).  om [B,S,768] bf16 ([hi|lo|hi]) or [B,S,256]
    fp32; proposals [B,S,4] fp32; idx [B,k] int64 -> (sel_raw like om with S -> k, sel_x [B,k,256] bf16 = bf16(hi+lo) or None,
    prop_sel [B,k,4], init_box [B,k,4] = sigmoid(prop_sel))."""
    require_cuda(om, "om")
    B, S, Wd = om.shape
    k = idx.shape[1]
    split = om.dtype in H16
    assert (split and Wd == 768) or (om.dtype == torch.float32 and Wd == 256), "om: [B,S,768] bf16 or [B,S,256] fp32"
    assert om.is_contiguous() and proposals.is_contiguous() and proposals.dtype == torch.float32 and idx.dtype == torch.int64
    idx = idx if idx.is_contiguous() else idx.contiguous()
    sel_raw = torch.empty((B, k, Wd), dtype=om.dtype, device=om.device)
    sel_x = torch.empty((B, k, 256), dtype=om.dtype, device=om.device) if split else None
    prop_sel = torch.empty((B, k, 4), dtype=torch.float32, device=om.device)
    init_box = torch.empty((B, k, 4), dtype=torch.float32, device=om.device)
    code = _L(om).dtlr_two_stage_gather(om.data_ptr(), proposals.data_ptr(), idx.data_ptr(), sel_raw.data_ptr(),
                                            None if sel_x is None else sel_x.data_ptr(), prop_sel.data_ptr(), init_box.data_ptr(),
                                            B, S, k, _DT[om.dtype], _lib.current_stream())
    _lib.check(code, "dtlr_two_stage_gather")
    return sel_raw, sel_x, prop_sel, init_box


def layernorm(x, w, b, eps: float = 1e-5, residual=None):
    """LayerNorm(x [+ residual]) over the last dim (HIP kernel: one wavefront per row, fp32 stats).
    x/residual [..., C] fp32 or bf16, contiguous; w, b [C] fp32."""
    C = x.shape[-1]
    if not x.is_contiguous():
        x = x.contiguous()
    if residual is not None and not residual.is_contiguous():
        residual = residual.contiguous()
    y = torch.empty_like(x)
    rows = x.numel() // C
    code = _L(x).dtlr_layernorm(x.data_ptr(), 0 if residual is None else residual.data_ptr(), w.data_ptr(), b.data_ptr(),
                                     y.data_ptr(), rows, C, eps, _DT[x.dtype], _lib.current_stream())
    _lib.check(code, "dtlr_layernorm")
    return y


def proj_pack_w(w):
    """[256,256] projection weight (bf16, any device) -> the fragment-major image dtlr_proj_ln_bf16 streams (bf16, same device)."""
    import numpy as np
    assert tuple(w.shape) == (256, 256)
    src = np.ascontiguousarray(w.detach().to(_hdt(w)).cpu().view(torch.int16).numpy()).view(np.uint16)
    out = np.empty(256 * 256, dtype=np.uint16)
    code = _L(w).dtlr_proj_pack_weights(src.ctypes.data, out.ctypes.data)
    _lib.check(code, "dtlr_proj_pack_weights")
    return torch.from_numpy(out.view(np.int16)).view(_hdt(w)).to(w.device)


def proj_ln(a, wp, b, residual, ln_w, ln_b, eps: float = 1e-5):
    """LayerNorm(residual + a W^T + b) in ONE kernel (dtlr_proj_ln_bf16): the attention block's output projection with its
    post-norm; a, residual [..., 256] bf16, wp = proj_pack_w(W) (bf16, 65536 elements), b / LN params fp32."""
    require_cuda(a, "a")
    assert a.dtype in H16 and residual.dtype in H16 and wp.dtype in H16 and a.shape[-1] == 256
    assert wp.numel() == 256 * 256 and wp.is_contiguous()
    a = a if a.is_contiguous() else a.contiguous()
    residual = residual if residual.is_contiguous() else residual.contiguous()
    y = torch.empty_like(residual)
    M = a.numel() // 256
    with _Timed("proj_ln_bf16", 2.0 * M * 256 * 256, 3.0 * M * 256 * 2 + 256 * 256 * 2):
        code = _L(a).dtlr_proj_ln_bf16(a.data_ptr(), wp.data_ptr(), b.data_ptr(), residual.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(),
                                            eps, y.data_ptr(), M, 256, _lib.current_stream())
    _lib.check(code, "dtlr_proj_ln_bf16")
    return y


def proj_ln_k256_pack(w):
    """[256, 256] projection weight -> the fragment-order image of dtlr_proj_ln_k256 (rows permuted so that a lane's two accumulator
    tiles are 8 consecutive channels): block ((wave*2 + e)*8 + ks) lane (m, g) <- W[32 wave + 8 (m>>2) + 4 e + (m&3)][32 ks + 8 g ..]."""
    assert tuple(w.shape) == (256, 256)
    return w.detach().to(_hdt(w)).view(8, 4, 2, 4, 8, 4, 8).permute(0, 2, 4, 5, 1, 3, 6).contiguous().view(-1)


def proj_ln_k256(a, wp, b, residual, ln_w, ln_b, eps: float = 1e-5):
    """LayerNorm(residual + a W^T + b) for many rows (dtlr_proj_ln_k256: weights resident in registers, the a / residual tiles DMA'd
    through an LDS ring); wp = proj_ln_k256_pack(W).  Same result as proj_ln up to fp32 summation order in the statistics."""
    require_cuda(a, "a")
    assert a.dtype in H16 and residual.dtype in H16 and wp.dtype in H16 and a.shape[-1] == 256 and wp.numel() == 65536
    a = a if a.is_contiguous() else a.contiguous()
    residual = residual if residual.is_contiguous() else residual.contiguous()
    y = torch.empty_like(residual)
    M = a.numel() // 256
    with _Timed("proj_ln_bf16", 2.0 * M * 256 * 256, 3.0 * M * 256 * 2 + 256 * 256 * 2):
        code = _L(a).dtlr_proj_ln_k256(a.data_ptr(), wp.data_ptr(), b.data_ptr(), residual.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(),
                                            eps, y.data_ptr(), M, _lib.current_stream())
    _lib.check(code, "dtlr_proj_ln_k256")
    return y


def proj_ln_split(a, wp, b, keep, ln_w, ln_b, eps: float = 1e-5):
    """LayerNorm(Linear(a with rows where keep == 0 zeroed)) written as [hi | lo | hi] bf16 (dtlr_proj_ln_split_bf16): a [..., 256]
    bf16, wp = proj_pack_w(W), keep [...] uint8/bool or None -> [..., 768] bf16.  hi + lo reproduces the fp32 LayerNorm output to
    2^-17 relative; see split_head_weight for the matching class-head layout."""
    require_cuda(a, "a")
    assert a.dtype in H16 and wp.dtype in H16 and a.shape[-1] == 256 and wp.numel() == 256 * 256
    a = a if a.is_contiguous() else a.contiguous()
    M = a.numel() // 256
    if keep is not None:
        keep = keep.reshape(-1)
        keep = (keep if keep.dtype in (torch.uint8, torch.bool) else (keep != 0)).contiguous()
        assert keep.numel() == M and keep.is_cuda
    y = torch.empty(a.shape[:-1] + (768,), dtype=a.dtype, device=a.device)
    with _Timed("proj_ln_bf16", 2.0 * M * 256 * 256, 4.0 * M * 256 * 2 + 256 * 256 * 2):
        code = _L(a).dtlr_proj_ln_split_bf16(a.data_ptr(), wp.data_ptr(), b.data_ptr(), keep.data_ptr() if keep is not None else None,
                                                  ln_w.data_ptr(), ln_b.data_ptr(), eps, y.data_ptr(), M, 256, _lib.current_stream())
    _lib.check(code, "dtlr_proj_ln_split_bf16")
    return y


def split_head_weight(w, b, pad_to: int = 64, dtype=torch.bfloat16):
    """fp32 head [C, 256] -> bf16 [Cp, 768] = [W_hi | W_hi | W_lo] (+ zero rows up to a multiple of `pad_to`) and an fp32 bias
    [Cp] whose padding is -inf, for use on a proj_ln_split activation: sum_k [hi|lo|hi][k] * [W_hi|W_hi|W_lo][k] =
    hi.W_hi + lo.W_hi + hi.W_lo, the three leading terms of the exact product."""
    C = w.shape[0]
    Cp = -(-C // pad_to) * pad_to
    wf = w.float()
    hi = wf.to(dtype)
    lo = (wf - hi.float()).to(dtype)
    out = torch.zeros((Cp, 768), dtype=dtype, device=w.device)
    out[:C, :256], out[:C, 256:512], out[:C, 512:] = hi, hi, lo
    bias = torch.full((Cp,), float("-inf"), dtype=torch.float32, device=w.device)
    bias[:C] = b.float()
    return out.contiguous(), bias


HEAD_TS_NEG = -3.0e38        # bias of a padded class in head_ts_pack (== HT_NEG of csrc/head_ts.hip)


def head_ts_pack(w, b, dtype=torch.bfloat16):
    """fp32 class head [N, 256] (+ bias [N]) on the GPU -> (image, bias_padded) for dtlr_head_ts: per 32-class chunk 32 KB = [W_hi fragments |
    W_lo fragments], 16 fragments of [64 lanes][8] each (lane l of k-step s <- W[32 c + (l & 31)][16 s + 8 (l >> 5) + e]: ffn_split_pack's
    W1 order), W_hi = dtype(W), W_lo = dtype(W - W_hi); dtlr_head_ts_pad_chunks() zero chunks behind; bias padded to 32 ceil(N / 32) with -3e38."""
    N = w.shape[0]
    assert w.dim() == 2 and w.shape[1] == 256 and dtype in H16
    nc = -(-N // 32)
    pad = int(_lib.lib().dtlr_head_ts_pad_chunks())
    wf = torch.zeros((nc * 32, 256), dtype=torch.float32, device=w.device)
    wf[:N] = w.float()
    hi = wf.to(dtype)
    lo = (wf - hi.float()).to(dtype)
    frag = lambda t: t.view(nc, 32, 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(nc, 8192)        # noqa: E731
    img = torch.zeros((nc + pad, 2 * 8192), dtype=dtype, device=w.device)
    img[:nc, :8192], img[:nc, 8192:] = frag(hi), frag(lo)
    bias = torch.full((nc * 32,), HEAD_TS_NEG, dtype=torch.float32, device=w.device)
    bias[:N] = b.float()
    return img.contiguous().view(-1), bias


def head_ts_supported(N: int, mode: str) -> bool:
    """dtlr_head_ts keeps the padded bias (and, for "logits", a per-wave transpose tile) in LDS: charsets up to 24576 classes for "rowmax",
    15360 for "logits" (head_ts.hip); "logits" also needs N % 4 == 0.  Larger heads run on the tiled GEMM (ops.linear / linear_rowmax)."""
    return N <= (15360 if mode == "logits" else 24576) and (mode != "logits" or N % 4 == 0)


def head_ts(x, img, bias, N: int, mode: str, a_off: int = 0, b_off=None):
    """Token-stationary class head (dtlr_head_ts): x [..., ldx] 16-bit rows; the 256-wide operand A = x[..., a_off:a_off+256] and, when b_off is
    given, B = x[..., b_off:b_off+256] (three products A.Whi + B.Whi + A.Wlo: x = proj_ln_split's [hi | lo | hi] image with a_off 0, b_off 256)
    or only A (two products: x = the 16-bit decoder state).  mode "rowmax" -> [...] fp32, "logits" -> [..., N] fp32.  (img, bias) = head_ts_pack."""
    require_cuda(x, "x")
    assert x.dtype in H16 and img.dtype == x.dtype and x.is_contiguous() and bias.dtype == torch.float32
    ldx = x.shape[-1]
    M = x.numel() // ldx
    nprod = 3 if b_off is not None else 2
    nc = -(-N // 32)
    assert bias.numel() == nc * 32 and img.numel() == (nc + int(_lib.lib().dtlr_head_ts_pad_chunks())) * 16384, "(img, bias) is not head_ts_pack of this N"
    out = torch.empty(x.shape[:-1] + ((N,) if mode == "logits" else ()), dtype=torch.float32, device=x.device)
    with _Timed(_gkind(x, img), 2.0 * M * N * 256 * nprod, float(M) * 256 * 2 * (nprod - 1) + float(N) * 512 * 2 + 4.0 * M * (N if mode == "logits" else 1),
                f"head_ts {mode} M{M} N{N} x{nprod}"):
        code = _L(x).dtlr_head_ts(x.data_ptr(), ldx, a_off, 0 if b_off is None else b_off, img.data_ptr(), bias.data_ptr(), N, nprod,
                                  1 if mode == "logits" else 0, out.data_ptr(), M, _lib.current_stream())
    _lib.check(code, "dtlr_head_ts")
    return out


def ffn_fused_supported(x, w1) -> bool:
    """The fused FFN kernel covers the bf16 engine at d_model 256, d_ff <= 2048 (multiple of 32)."""
    return x.dtype in H16 and x.shape[-1] == 256 and w1.shape[0] % 32 == 0 and w1.shape[0] <= 2048


def ffn_pack_w2(w2):
    """linear2.weight [256, d_ff] -> the chunk-major layout dtlr_ffn_fused_bf16 streams: [d_ff/32, 256, 32]."""
    o, dff = w2.shape
    return w2.view(o, dff // 32, 32).permute(1, 0, 2).contiguous()


FFN32_PAD = 4            # == dtlr_ffn32_pad_chunks(): zero chunks behind each packed image (streamed, never multiplied)


def ffn32_pack(w1, w2):
    """linear1.weight [d_ff, 256], linear2.weight [256, d_ff] -> the two fragment-order images of dtlr_ffn32_bf16
    (== dtlr_ffn32_pack_weights).  W1p[c][s][l][e] = W1[32 c + (l & 31)][16 s + 8 (l >> 5) + e];
    W2p[c][8 s + ct][l][e] = W2[32 ct + (l & 31)][32 c + 8 (2 s + (e >> 2)) + 4 (l >> 5) + (e & 3)]; FFN32_PAD zero chunks appended."""
    d_ff = w1.shape[0]
    assert tuple(w1.shape) == (d_ff, 256) and tuple(w2.shape) == (256, d_ff) and d_ff % 32 == 0 and 64 <= d_ff <= 2048
    nc = d_ff // 32
    a = w1.detach().to(_hdt(w1)).view(nc, 32, 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(nc, 8192)
    b = w2.detach().to(_hdt(w1)).view(8, 32, nc, 2, 2, 2, 4).permute(2, 3, 0, 5, 1, 4, 6).reshape(nc, 8192)
    pad = torch.zeros((FFN32_PAD, 8192), dtype=a.dtype, device=w1.device)
    return torch.cat([a, pad]).contiguous().view(-1), torch.cat([b, pad]).contiguous().view(-1)


def ffn32(x, w1p, b1, w2p, b2, ln_w, ln_b, eps: float = 1e-5, out=None):
    """LayerNorm(x + relu(x W1^T + b1) W2^T + b2) for many rows (dtlr_ffn32_bf16: 32x32x16 MFMAs, one wave per SIMD, 256 rows per
    workgroup); (w1p, w2p) = ffn32_pack(W1, W2).  Same result as ffn_fused up to fp32 summation order."""
    require_cuda(x, "x")
    d_ff = b1.numel()
    assert x.dtype in H16 and x.shape[-1] == 256 and w1p.dtype in H16 and w2p.dtype in H16
    assert w1p.numel() == (d_ff // 32 + FFN32_PAD) * 8192 and w2p.numel() == w1p.numel()
    x = x if x.is_contiguous() else x.contiguous()
    y = torch.empty_like(x) if out is None else out
    assert y.is_contiguous() and y.shape == x.shape and y.dtype == x.dtype
    M = x.numel() // 256
    with _Timed("ffn_fused_bf16", 4.0 * M * 256 * d_ff, 2.0 * M * 256 * 2 + 2.0 * 256 * d_ff * 2, symbol="ffn3_bf16_kernel<0>"):
        code = _L(x).dtlr_ffn32_bf16(x.data_ptr(), w1p.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(),
                                          ln_w.data_ptr(), ln_b.data_ptr(), eps, y.data_ptr(), M, d_ff, _lib.current_stream())
    _lib.check(code, "dtlr_ffn32_bf16")
    return y


def ffn4(x, w1p, b1, w2p, b2, ln_w, ln_b, eps: float = 1e-5, out=None):
    """LayerNorm(x + relu(x W1^T + b1) W2^T + b2) for any number of rows as ONE persistent launch (dtlr_ffn4_bf16, round 6: two 128-row tiles
    per workgroup half a period apart over a cyclic weight stream; epilogues and loads under the other tile's MFMAs); (w1p, w2p) =
    ffn32_pack(W1, W2).  Same result as ffn32 / ffn_fused up to the fp32 summation order over the hidden chunks."""
    require_cuda(x, "x")
    d_ff = b1.numel()
    assert x.dtype in H16 and x.shape[-1] == 256 and w1p.dtype in H16 and w2p.dtype in H16
    assert w1p.numel() == (d_ff // 32 + FFN32_PAD) * 8192 and w2p.numel() == w1p.numel()
    x = x if x.is_contiguous() else x.contiguous()
    y = torch.empty_like(x) if out is None else out
    assert y.is_contiguous() and y.shape == x.shape and y.dtype == x.dtype
    M = x.numel() // 256
    with _Timed("ffn_fused_bf16", 4.0 * M * 256 * d_ff, 2.0 * M * 256 * 2 + 2.0 * 256 * d_ff * 2, symbol="ffn4_bf16_kernel"):
        code = _L(x).dtlr_ffn4_bf16(x.data_ptr(), w1p.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(),
                                    ln_w.data_ptr(), ln_b.data_ptr(), eps, y.data_ptr(), M, d_ff, _lib.current_stream())
    _lib.check(code, "dtlr_ffn4_bf16")
    return y


def ffn_fused(x, w1, b1, w2p, b2, ln_w, ln_b, eps: float = 1e-5, out=None):
    """LayerNorm(x + relu(x W1^T + b1) W2^T + b2) in ONE kernel (dtlr_ffn_fused_bf16): the [M, d_ff]
    intermediate never reaches HBM.  x [..., 256] bf16; W1 [d_ff,256] bf16; w2p = ffn_pack_w2(W2) ([d_ff/32,256,32] bf16);
    biases / LN params fp32."""
    require_cuda(x, "x")
    if w2p.dim() != 3 or w2p.shape[1] != 256 or w2p.shape[2] != 32 or w2p.shape[0] * 32 != w1.shape[0]:
        raise ValueError("ffn_fused: w2p must be ffn_pack_w2(linear2.weight) of shape [d_ff/32, 256, 32]")
    x = x if x.is_contiguous() else x.contiguous()
    y = torch.empty_like(x) if out is None else out
    assert y.is_contiguous() and y.shape == x.shape and y.dtype == x.dtype
    M = x.numel() // x.shape[-1]
    # dtlr_ffn_fused_bf16's dispatch (ffn.hip): one ffn2_bf16_kernel<3> launch for 32768 < M <= 49152, the first structure at or below
    # ... and, when its 128-token tiles fill at most 96 workgroups (M <= 12288) at d_ff >= 512, the hidden-split pair
    # ffn_fused_bf16_kernel<0, false, true> + ffn_fused_finish_kernel: TWO kernels, accounted as a class (symbol None) so that a
    # roofline_by_kernel row always joins to exactly one rocprof symbol
    hidden_split = (M + 127) // 128 <= 96 and w1.shape[0] >= 512 and min(8, 256 // ((M + 127) // 128), (w1.shape[0] // 32) // 8) >= 2
    sym = "ffn2_bf16_kernel<3>" if 32768 < M <= 49152 else ("ffn_fused_bf16_kernel<0, false, false>" if (M <= 32768 and not hidden_split) else None)
    with _Timed("ffn_fused_bf16", 4.0 * M * x.shape[-1] * w1.shape[0], 2.0 * M * 256 * 2 + 2.0 * w1.numel() * 2, symbol=sym):
        code = _L(x).dtlr_ffn_fused_bf16(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(),
                                              ln_w.data_ptr(), ln_b.data_ptr(), eps, y.data_ptr(), M, x.shape[-1], w1.shape[0],
                                              _lib.current_stream())
    _lib.check(code, "dtlr_ffn_fused_bf16")
    return y


def ffn_split_pack(w1, w2):
    """linear1.weight [d_ff, 256], linear2.weight [256, d_ff] (fp32, any device) -> the chunk-major split image dtlr_ffn_split streams (uint8,
    same device): per 32-unit chunk one 64 KB block [W1_hi | W1_lo | W2_hi | W2_lo], each 16 fragments of 1 KB in the fragment order of
    ffn32_pack (W1: lane l <- W1[32 c + (l & 31)][16 s + 8 (l >> 5) + e]; W2: fragment 8 s + ct, lane l <- W2[32 ct + (l & 31)][32 c +
    8 (2 s + (e >> 2)) + 4 (l >> 5) + (e & 3)]); hi = fp16(w), lo = fp16(w - hi).  An odd chunk count is padded to even and
    dtlr_ffn_split_pad_chunks() zero chunks follow (streamed by the kernel's look-ahead, multiplied into nothing)."""
    d_ff = w1.shape[0]
    assert tuple(w1.shape) == (d_ff, 256) and tuple(w2.shape) == (256, d_ff) and d_ff % 32 == 0 and 32 <= d_ff <= 2048
    nc = d_ff // 32
    total = ((nc + 1) & ~1) + int(_lib.lib().dtlr_ffn_split_pad_chunks())
    w1f, w2f = w1.detach().float(), w2.detach().float()
    parts = []
    for wf, kind in ((w1f, 1), (w2f, 2)):
        hi = wf.half()
        for t in (hi, (wf - hi.float()).half()):
            if kind == 1:
                parts.append(t.view(nc, 32, 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(nc, 8192))
            else:
                parts.append(t.view(8, 32, nc, 2, 2, 2, 4).permute(2, 3, 0, 5, 1, 4, 6).reshape(nc, 8192))
    img = torch.cat(parts, 1)                                     # [nc, 4 x 8192 halves] = 64 KB per chunk
    out = torch.zeros((total, 4 * 8192), dtype=torch.float16, device=w1.device)
    out[:nc] = img
    return out.view(torch.uint8).reshape(-1)


def ffn_split(x, wp, b1, b2, ln_w, ln_b, eps: float = 1e-5, out=None):
    """LayerNorm(x + relu(x W1^T + b1) W2^T + b2) for fp32 rows in ONE kernel, every product as three fp16 MFMAs on hi + lo halves
    (dtlr_ffn_split: the split-fp32 engine's FFN block; the [M, d_ff] intermediate never reaches HBM).  x [..., 256] fp32;
    wp = ffn_split_pack(W1, W2); b1 [d_ff], b2 / LN params [256] fp32."""
    require_cuda(x, "x")
    d_ff = b1.numel()
    assert x.dtype == torch.float32 and x.shape[-1] == 256 and wp.dtype == torch.uint8 and d_ff % 32 == 0
    assert wp.numel() == ((((d_ff // 32) + 1) & ~1) + int(_lib.lib().dtlr_ffn_split_pad_chunks())) * 65536, "wp is not ffn_split_pack(W1, W2) of this d_ff"
    x = x if x.is_contiguous() else x.contiguous()
    y = torch.empty_like(x) if out is None else out
    assert y.is_contiguous() and y.shape == x.shape and y.dtype == x.dtype
    M = x.numel() // 256
    # dtlr_ffn_split's dispatch (ffn_split.hip): whole rounds of 256 tiles on ffn_split_kernel<false>; a last round filling at most half
    # of the chip runs as ffn_split_kernel<true> + ffn_split_finish_kernel -- then the launch is THREE kernels: class accounting
    ntiles = (M + 127) // 128
    full, rem = (ntiles // 256) * 256, ntiles % 256
    nc2 = ((d_ff // 32) + 1) & ~1
    ns = min(4, 256 // rem, nc2 // 8) if (rem > 0 and full > 0) else 1
    with _Timed("ffn_fused_f32s", 4.0 * M * 256 * d_ff, 2.0 * M * 256 * 4 + 2.0 * 256 * d_ff * 4, symbol="ffn_split_kernel<false>" if ns < 2 else None):
        code = _lib.lib().dtlr_ffn_split(x.data_ptr(), wp.data_ptr(), b1.data_ptr(), b2.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(),
                                         eps, y.data_ptr(), M, d_ff, _lib.current_stream())
    _lib.check(code, "dtlr_ffn_split")
    return y


def k256s_pack(w):
    """fp32 [256, 256] weight on the GPU -> the resident-operand image of dtlr_gemm_k256s (int16 [2 x 65536]: hi then lo fp16 halves in MFMA
    fragment order; dtlr_k256s_pack_weights)."""
    require_cuda(w, "w")
    assert tuple(w.shape) == (256, 256)
    w = w.detach().float().contiguous()
    out = torch.empty(2 * 65536, dtype=torch.int16, device=w.device)
    _lib.check(_lib.lib().dtlr_k256s_pack_weights(w.data_ptr(), out.data_ptr(), _lib.current_stream()), "dtlr_k256s_pack_weights")
    return out


def gemm_k256s(x, wp, b, residual=None, row_mask=None, ln_w=None, ln_b=None, eps: float = 1e-5):
    """The split-fp32 engine's weight-resident streaming K = N = 256 projection (dtlr_gemm_k256s).  x [..., 256] fp32, wp = k256s_pack(W).
    no LN params:  x W^T + b with the rows flagged in row_mask (bool [M], optional) written as zeros (value_proj + masked_fill);
    residual:      LayerNorm(residual + x W^T + b) (output_proj + residual + norm1);
    LN params, no residual: LayerNorm(b + x W^T) with the PRODUCT of the flagged rows taken as zero (enc_output on the masked memory + norm)."""
    require_cuda(x, "x")
    assert x.dtype == torch.float32 and x.shape[-1] == 256 and wp.dtype == torch.int16 and wp.numel() == 2 * 65536
    x = x if x.is_contiguous() else x.contiguous()
    M = x.numel() // 256
    y = torch.empty_like(x)
    assert (ln_w is None) == (ln_b is None)
    if residual is not None:
        assert residual.dtype == torch.float32 and residual.shape == x.shape and ln_w is not None and row_mask is None
        residual = residual if residual.is_contiguous() else residual.contiguous()
    if row_mask is not None:
        assert row_mask.numel() == M and row_mask.dtype in (torch.bool, torch.uint8)
        row_mask = row_mask if row_mask.is_contiguous() else row_mask.contiguous()
    nbytes = float(M) * 256 * 4 * (3 if residual is not None else 2) + 2.0 * 65536 * 2
    tag = f"k256s M{M}" + ("+res+ln" if residual is not None else "+ln" if ln_w is not None else "")
    with _Timed("gemm_f32s", 2.0 * M * 256 * 256, nbytes, tag):
        code = _lib.lib().dtlr_gemm_k256s(x.data_ptr(), wp.data_ptr(), 0 if b is None else b.data_ptr(), 0 if residual is None else residual.data_ptr(),
                                          0 if row_mask is None else row_mask.data_ptr(), 0 if ln_w is None else ln_w.data_ptr(),
                                          0 if ln_b is None else ln_b.data_ptr(), eps, y.data_ptr(), M, _lib.current_stream())
    _lib.check(code, "dtlr_gemm_k256s")
    return y


def gemm_k256s_multi(x, slices, row_mask=None, res_rows: int = 0):
    """ONE pass over x [..., 256] fp32 for several projections of it (dtlr_gemm_k256s_multi; split-fp32 engine).  slices: list of dicts
    {wp: k256s_pack image of the slice's weight zero-padded to [256, 256], out: fp32 view [..., n] (last dim contiguous, any row stride),
     bias: [n] fp32 or None, residual: fp32 [rows, n] view or None, relu: bool}; n a multiple of 32, <= 256.  res_rows > 0: every residual is
    ONE [res_rows, n] matrix shared by the M / res_rows images (row m pairs with row m % res_rows).  row_mask (bool [M]): those rows are
    written as zeros in every slice."""
    require_cuda(x, "x")
    assert x.dtype == torch.float32 and x.shape[-1] == 256 and x.is_contiguous() and 1 <= len(slices) <= 8
    M = x.numel() // 256
    arr = (_lib.K256sSlice * len(slices))()
    nbytes, ncols = float(M) * 256 * 4, 0
    for i, sl in enumerate(slices):
        out, wp, res = sl["out"], sl["wp"], sl.get("residual")
        n = out.shape[-1]
        assert out.dtype == torch.float32 and out.is_cuda and out.stride(-1) == 1 and out.numel() // n == M and n % 32 == 0 and n <= 256
        o2 = out.reshape(M, n) if out.is_contiguous() else out
        ldc = o2.stride(-2)
        assert wp.dtype == torch.int16 and wp.numel() == 2 * 65536
        arr[i].Wp, arr[i].C, arr[i].ldc, arr[i].n_valid, arr[i].relu = wp.data_ptr(), out.data_ptr(), int(ldc), int(n), int(bool(sl.get("relu")))
        b = sl.get("bias")
        arr[i].bias = 0 if b is None else b.data_ptr()
        if b is not None:
            assert b.dtype == torch.float32 and b.numel() >= n and b.is_contiguous()
        if res is not None:
            assert res.dtype == torch.float32 and res.stride(-1) == 1 and res.shape[-1] == n and res.numel() // n == (res_rows if res_rows else M)
            arr[i].R, arr[i].ldr = res.data_ptr(), int(res.stride(-2))
            nbytes += 4.0 * n * (res_rows if res_rows else M)
        else:
            arr[i].R, arr[i].ldr = 0, 0
        nbytes += 4.0 * M * n + 2.0 * 65536 * 2
        ncols += n
    if row_mask is not None:
        assert row_mask.numel() == M and row_mask.dtype in (torch.bool, torch.uint8) and row_mask.is_contiguous()
    with _Timed("gemm_f32s", 2.0 * M * 256 * ncols, nbytes, f"k256s_multi M{M} N{ncols}x{len(slices)}"):
        code = _lib.lib().dtlr_gemm_k256s_multi(x.data_ptr(), M, ctypes.addressof(arr), len(slices), 0 if row_mask is None else row_mask.data_ptr(),
                                                int(res_rows), _lib.current_stream())
    _lib.check(code, "dtlr_gemm_k256s_multi")


def conv2d_nhwc(x, w, bias, stride: int, padding: int, relu=False, residual=None):
    """NHWC convolution + folded-BN bias [+ residual] [+ ReLU].  x [B,H,W,Cin] contiguous.
    w: [Cout,KH,KW,Cin] contiguous ("OHWI") -> the implicit-GEMM HIP kernel (dtlr_conv2d_nhwc), which
    needs Cin*elem % 128 == 0 (anything else raises: the 3-channel 7x7 stem has its own kernels).
    relu: False/0, True/2 = ReLU after the residual add (bottleneck tail)."""
    if w.dim() == 4 and w.is_contiguous() and w.shape[3] == x.shape[3] and (x.shape[3] * x.element_size()) % 128 == 0:
        B, H, W, Cin = x.shape
        Cout, KH, KW, _ = w.shape
        Ho, Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
        x = x if x.is_contiguous() else x.contiguous()
        y = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
        if residual is not None:
            residual = residual if residual.is_contiguous() else residual.contiguous()
        es = x.element_size()
        nbytes = (float(x.numel()) / (stride * stride if KH == 1 else 1) + float(w.numel()) + float(B) * Ho * Wo * Cout * (2 if residual is not None else 1)) * es
        tag = f"conv{KH}x{KW}s{stride} M{B * Ho * Wo} N{Cout} K{KH * KW * Cin}" + ("+res" if residual is not None else "")
        with _Timed(_gkind(x, w), 2.0 * B * Ho * Wo * Cout * KH * KW * Cin, nbytes, tag):
            code = _L(x).dtlr_conv2d_nhwc(x.data_ptr(), w.data_ptr(), 0 if bias is None else bias.data_ptr(),
                                               0 if residual is None else residual.data_ptr(), y.data_ptr(),
                                               B, H, W, Cin, Cout, KH, KW, stride, padding, 2 if relu else 0,
                                               _in_dt(x, w), _lib.current_stream())
        _lib.check(code, "dtlr_conv2d_nhwc")
        return y
    raise _lib.DTLRError(f"ops.conv2d_nhwc: no HIP kernel for x {tuple(x.shape)} {x.dtype}, w {tuple(w.shape)} (weights must be "
                         "[Cout,KH,KW,Cin] contiguous with Cin * element size a multiple of 128 bytes; the 3-channel stem has its "
                         "own kernels: stem_conv7x7 / stem_conv7x7_f32); there is no library fallback")


def stem_pack_weights(w_oihw, dtype=torch.bfloat16):
    """conv1.weight with the FrozenBN scale folded, [64,3,7,7] (any float dtype, any device) -> the 24 KB fragment-major
    bf16 weight image dtlr_stem_conv7x7 keeps in registers (uint16 tensor on the CPU; move it to the device once)."""
    import numpy as np
    w = np.ascontiguousarray(w_oihw.detach().float().cpu().numpy())
    assert w.shape == (64, 3, 7, 7)
    out = np.zeros(4 * 6 * 64 * 8, dtype=np.uint16)
    code = _L(dtype).dtlr_stem_pack_weights(w.ctypes.data, out.ctypes.data)
    _lib.check(code, "dtlr_stem_pack_weights")
    return torch.from_numpy(out.view(np.int16)).clone()


def stem_pack_weights_f32(w_oihw):
    """conv1.weight (FrozenBN scale folded) [64,3,7,7] -> the k-major fp32 image [147, 64] (k = (ci*7 + kh)*7 + kw) the fp32 stem
    kernel keeps in LDS."""
    assert tuple(w_oihw.shape) == (64, 3, 7, 7)
    return w_oihw.detach().float().reshape(64, 147).t().contiguous()


def stem_conv7x7_f32(x_nchw, wk):
    """ResNet stem 7x7/s2/p3 convolution 3 -> 64 in exact fp32 (the parity engine; HIP kernel dtlr_stem_conv7x7_f32, direct
    convolution on the vector ALUs with the patch and the weights in LDS): x [B,3,H,W] fp32 NCHW -> [B,Ho,Wo,64] fp32 NHWC, no
    bias (the max-pool pass applies the folded-BN shift + ReLU)."""
    require_cuda(x_nchw, "images")
    assert x_nchw.dtype == torch.float32 and x_nchw.dim() == 4 and x_nchw.shape[1] == 3 and tuple(wk.shape) == (147, 64)
    x = x_nchw if x_nchw.is_contiguous() else x_nchw.contiguous()
    B, _, H, W = x.shape
    y = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64), dtype=torch.float32, device=x.device)
    code = _L(x_nchw).dtlr_stem_conv7x7_f32(x.data_ptr(), wk.data_ptr(), y.data_ptr(), B, H, W, _lib.current_stream())
    _lib.check(code, "dtlr_stem_conv7x7_f32")
    return y


def stem_pack_weights_split(w_oihw):
    """conv1.weight (FrozenBN scale folded) [64,3,7,7] fp32 -> (hi, lo): two fragment images of dtlr_stem_conv7x7_f32s (fp16, as uint16 CPU
    tensors): the fp16 build's packer applied to fp16(w) and to w - fp16(w)."""
    wf = w_oihw.detach().float().cpu()
    hi = wf.half().float()
    return stem_pack_weights(hi, torch.float16), stem_pack_weights(wf - hi, torch.float16)


def stem_conv7x7_f32s(x_nchw, wfrag_hi, wfrag_lo):
    """ResNet stem 7x7/s2/p3 convolution 3 -> 64 for the split-fp32 engine (dtlr_stem_conv7x7_f32s: the MFMA formulation of stem_conv7x7
    with the image and the weights as fp16 hi + lo halves, three MFMAs per product): x [B,3,H,W] fp32 NCHW -> [B,Ho,Wo,64] fp32 NHWC, no bias."""
    require_cuda(x_nchw, "images")
    assert x_nchw.dtype == torch.float32 and x_nchw.dim() == 4 and x_nchw.shape[1] == 3
    assert wfrag_hi.numel() == 4 * 6 * 64 * 8 and wfrag_lo.numel() == wfrag_hi.numel() and wfrag_hi.is_cuda and wfrag_lo.is_cuda
    x = x_nchw if x_nchw.is_contiguous() else x_nchw.contiguous()
    B, _, H, W = x.shape
    y = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64), dtype=torch.float32, device=x.device)
    code = _lib.lib().dtlr_stem_conv7x7_f32s(x.data_ptr(), wfrag_hi.data_ptr(), wfrag_lo.data_ptr(), y.data_ptr(), B, H, W, _lib.current_stream())
    _lib.check(code, "dtlr_stem_conv7x7_f32s")
    return y


def stem_conv7x7(x_nchw, wfrag, out_dtype=torch.bfloat16):
    """ResNet stem 7x7/s2/p3 convolution 3 -> 64 on the bf16 MFMA (HIP kernel): x [B,3,H,W] fp32 NCHW -> [B,Ho,Wo,64] bf16
    NHWC, no bias (the max-pool pass applies the folded-BN shift + ReLU)."""
    require_cuda(x_nchw, "images")
    assert x_nchw.dtype == torch.float32 and x_nchw.dim() == 4 and x_nchw.shape[1] == 3
    x = x_nchw if x_nchw.is_contiguous() else x_nchw.contiguous()
    B, _, H, W = x.shape
    y = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64), dtype=out_dtype, device=x.device)
    code = _L(out_dtype).dtlr_stem_conv7x7(x.data_ptr(), wfrag.data_ptr(), y.data_ptr(), B, H, W, _DT[out_dtype], _lib.current_stream())
    _lib.check(code, "dtlr_stem_conv7x7")
    return y


def stem_conv7x7_pool(x_nchw, wfrag, bias, out_dtype=torch.bfloat16):
    """conv1 (7x7/s2/p3, folded-BN scale in the weights) + shift + ReLU + 3x3/s2/p1 max-pool in ONE kernel (dtlr_stem_conv7x7_pool, 16-bit
    engines): x [B,3,H,W] fp32 NCHW -> [B,Hp,Wp,64] NHWC.  The full-resolution 64-channel map never reaches HBM."""
    require_cuda(x_nchw, "images")
    assert x_nchw.dtype == torch.float32 and x_nchw.dim() == 4 and x_nchw.shape[1] == 3 and bias.dtype == torch.float32 and bias.numel() == 64
    x = x_nchw if x_nchw.is_contiguous() else x_nchw.contiguous()
    B, _, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty((B, (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1, 64), dtype=out_dtype, device=x.device)
    code = _L(out_dtype).dtlr_stem_conv7x7_pool(x.data_ptr(), wfrag.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, W, _DT[out_dtype], _lib.current_stream())
    _lib.check(code, "dtlr_stem_conv7x7_pool")
    return y


def maxpool_nhwc(x, k: int = 3, stride: int = 2, padding: int = 1, bias=None, relu: bool = False):
    """3x3/s2/p1 max pooling on NHWC (HIP kernel); with bias/relu: maxpool(relu(x + bias)) in the same pass."""
    assert (k, stride, padding) == (3, 2, 1)
    B, H, W, C = x.shape
    x = x if x.is_contiguous() else x.contiguous()
    y = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=x.dtype, device=x.device)
    code = _L(x).dtlr_maxpool3x3s2_nhwc(x.data_ptr(), y.data_ptr(), 0 if bias is None else bias.data_ptr(), 1 if relu else 0,
                                             B, H, W, C, _DT[x.dtype], _lib.current_stream())
    _lib.check(code, "dtlr_maxpool3x3s2_nhwc")
    return y


def groupnorm_tokens(x, groups: int, w, b, eps: float = 1e-5, out=None):
    """GroupNorm(32, 256) over [B, T, C] tokens of one feature level: statistics per (sample,
    group) over (C/groups channels x T positions) -- models/dino/dino.py:121-134 (HIP kernels)."""
    B, T, C = x.shape
    x = x if x.is_contiguous() else x.contiguous()
    L_ = _L(x)
    ws = torch.empty(L_.dtlr_groupnorm_workspace_bytes(B, T), dtype=torch.uint8, device=x.device)
    # out: a [B, T, C] slice (dim 1) of a larger contiguous [B, S, C] token matrix -- the level is normalised straight into place
    y = torch.empty_like(x) if out is None else out
    assert y.shape == x.shape and y.dtype == x.dtype and y.stride(2) == 1 and y.stride(1) == C
    code = L_.dtlr_groupnorm_tokens_strided(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), y.stride(0), ws.data_ptr(),
                                            B, T, C, groups, eps, _DT[x.dtype], _lib.current_stream())
    _lib.check(code, "dtlr_groupnorm_tokens")
    return y


# bench.py sets this to a list to time every MSDA launch with HIP events recorded on the launch
# stream: entries are (start_event, end_event, N, Lq, S).
MSDA_EVENTS = None


def msda(value, spatial_shapes, level_start_index, loc, attn):
    """value [N,S,M,D]; loc [N,Lq,M,L,P,2] f32; attn [N,Lq,M,L,P] f32 -> [N,Lq,M*D] (HIP kernel)."""
    if MSDA_EVENTS is None:
        return _msda.ms_deform_attn_forward(value, spatial_shapes, level_start_index, loc, attn, 64)
    st = torch.cuda.current_stream()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    out = _msda.ms_deform_attn_forward(value, spatial_shapes, level_start_index, loc, attn, 64)
    b.record(st)
    MSDA_EVENTS.append((a, b, value.shape[0], loc.shape[1], value.shape[1]))
    return out


def msda_fused(value, spatial_shapes, level_start_index, ow, ref):
    """MSDeformAttn front end fused into the sampling kernel (L=4, P=4): value [N,S,M,D] (fp32/bf16),
    ow [N,Lq,M*48] raw [offsets|logits] projection (fp32, or bf16 with bf16 value),
    ref [N,Lq,L,2|4] fp32 -> [N,Lq,M*D]."""
    N, S, M, D = value.shape
    Lq = ow.shape[1]
    L = spatial_shapes.shape[0]
    assert ow.shape[2] == M * L * 4 * 3 and ref.dtype == torch.float32 and ref.shape[:3] == (N, Lq, L)
    assert ow.is_contiguous() and ref.is_contiguous()
    # value may be a column slice of a wider [N, S, row_stride] tensor (all decoder layers' value projections in one GEMM)
    vs = value.stride()
    assert vs[3] == 1 and vs[2] == D and vs[0] == S * vs[1] and vs[1] >= M * D, "value: [N,S,M,D] view with contiguous heads"
    out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
    ev = MSDA_EVENTS
    if ev is not None:
        st = torch.cuda.current_stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
    code = _L(value).dtlr_msda_fused_forward_strided(value.data_ptr(), vs[1], spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                                      ow.data_ptr(), ref.data_ptr(), ref.shape[-1], N, S, M, D, L, Lq, 4,
                                                      _DT[value.dtype], _DT[ow.dtype], out.data_ptr(), _lib.current_stream())
    _lib.check(code, "dtlr_msda_fused_forward_strided")
    if ev is not None:
        b.record(st)
        ev.append((a, b, N, Lq, S))
    return out


MSDA_HALO = 8            # columns staged beyond a tile's own (level-0 pixels); tools/msda_sweep.py varies it


def swin_patch_embed(x_nchw, w_kE, b, ln_w, ln_b, out_dtype, eps: float = 1e-5):
    """PatchEmbed of a Swin backbone (dtlr_swin_patch_embed): x [B,3,H,W] fp32 -> [B, ceil(H/4), ceil(W/4), E] out_dtype
    (4x4/s4 convolution with zero padding + LayerNorm).  w_kE [48, E] fp32 = proj.weight.reshape(E, 48).t()."""
    require_cuda(x_nchw, "images")
    assert x_nchw.dtype == torch.float32 and x_nchw.dim() == 4 and x_nchw.shape[1] == 3 and w_kE.shape[0] == 48
    x = x_nchw if x_nchw.is_contiguous() else x_nchw.contiguous()
    B, _, H, W = x.shape
    E = w_kE.shape[1]
    out = torch.empty((B, (H + 3) // 4, (W + 3) // 4, E), dtype=out_dtype, device=x.device)
    code = _L(out_dtype).dtlr_swin_patch_embed(x.data_ptr(), w_kE.data_ptr(), b.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), out.data_ptr(),
                                            B, H, W, E, eps, _DT[out_dtype], _lib.current_stream())
    _lib.check(code, "dtlr_swin_patch_embed")
    return out


def swin_window_attn(qkv, qkv_bias, rpb, n_heads: int, window: int, shift: int):
    """Attention core of a Swin block (dtlr_swin_window_attn): qkv [B,H,W,3C] -> [B,H,W,C]; rpb = swin_dense_bias(table, window)."""
    require_cuda(qkv, "qkv")
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    assert qkv.is_contiguous() and qkv_bias.dtype == torch.float32 and rpb.dtype == torch.float32 and rpb.is_contiguous()
    out = torch.empty((B, H, W, C), dtype=qkv.dtype, device=qkv.device)
    code = _L(qkv).dtlr_swin_window_attn(qkv.data_ptr(), qkv_bias.data_ptr(), rpb.data_ptr(), out.data_ptr(), B, H, W, C, n_heads, window, shift,
                                            _DT[qkv.dtype], _lib.current_stream())
    _lib.check(code, "dtlr_swin_window_attn")
    return out


def swin_dense_bias(table, window: int):
    """relative_position_bias_table [(2w-1)^2, nH] -> the dense, zero-padded bias [nH, ceil16(N), ceil32(N)] fp32 the attention kernel
    reads (swin_transformer.py:96-106,127-130: table[relative_position_index]); built once per block at pack time."""
    N = window * window
    c = torch.stack(torch.meshgrid(torch.arange(window), torch.arange(window), indexing="ij")).flatten(1)
    rel = (c[:, :, None] - c[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += window - 1
    rel[:, :, 1] += window - 1
    rel[:, :, 0] *= 2 * window - 1
    idx = rel.sum(-1).view(-1).to(table.device)
    dense = table.float()[idx].view(N, N, -1).permute(2, 0, 1)
    out = torch.zeros((dense.shape[0], -(-N // 16) * 16, -(-N // 32) * 32), dtype=torch.float32, device=table.device)
    out[:, :N, :N] = dense
    return out.contiguous()


def swin_patch_merge(x, ln_w, ln_b, eps: float = 1e-5):
    """PatchMerging up to its LayerNorm (dtlr_swin_patch_merge): x [B,H,W,C] -> [B, ceil(H/2), ceil(W/2), 4C]."""
    require_cuda(x, "x")
    B, H, W, C = x.shape
    x = x if x.is_contiguous() else x.contiguous()
    y = torch.empty((B, (H + 1) // 2, (W + 1) // 2, 4 * C), dtype=x.dtype, device=x.device)
    code = _L(x).dtlr_swin_patch_merge(x.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), y.data_ptr(), B, H, W, C, eps, _DT[x.dtype], _lib.current_stream())
    _lib.check(code, "dtlr_swin_patch_merge")
    return y


_POS_TABLES = {}


def geometry(mask, level_hw, level_embed, temperature_h: float, temperature_w: float, pos_dtype):
    """Everything that depends only on the padding mask, in one launch (dtlr_geometry): per-level masks, valid ratios,
    sine position embedding + level embedding, encoder reference points, proposals and their validity.
    mask [B,H,W] bool (True = padding); level_hw: host list of 4 (H_l, W_l); level_embed [4,256] fp32.
    -> dict(mask_flat [B,S] bool, keep [B,S] bool, pos [B,S,256] pos_dtype, valid_ratios [B,4,2], enc_ref [B,S,4,2], proposals [B,S,4])."""
    require_cuda(mask, "mask")
    assert mask.dtype == torch.bool and mask.dim() == 3 and len(level_hw) == 4
    mask = mask if mask.is_contiguous() else mask.contiguous()
    B, H, W = mask.shape
    dev = mask.device
    key = (dev, float(temperature_h), float(temperature_w))
    if key not in _POS_TABLES:      # the exact tables PositionEmbeddingSineHW builds (position_encoding.py:95-98), once per device
        i = torch.arange(128, dtype=torch.float32, device=dev)
        e = 2 * torch.div(i, 2, rounding_mode="floor") / 128
        _POS_TABLES[key] = ((temperature_h ** e).contiguous(), (temperature_w ** e).contiguous())
    dim_ty, dim_tx = _POS_TABLES[key]
    S = sum(int(h) * int(w) for h, w in level_hw)
    hw = (ctypes.c_int * 8)(*[int(v) for pair in level_hw for v in pair])
    mask_flat = torch.empty((B, S), dtype=torch.bool, device=dev)
    keep = torch.empty((B, S), dtype=torch.bool, device=dev)
    pos = torch.empty((B, S, 256), dtype=pos_dtype, device=dev)
    vr = torch.empty((B, 4, 2), dtype=torch.float32, device=dev)
    enc_ref = torch.empty((B, S, 4, 2), dtype=torch.float32, device=dev)
    prop = torch.empty((B, S, 4), dtype=torch.float32, device=dev)
    code = _L(pos_dtype).dtlr_geometry(mask.data_ptr(), B, H, W, ctypes.cast(hw, ctypes.c_void_p), level_embed.data_ptr(),
                                    dim_ty.data_ptr(), dim_tx.data_ptr(), _DT[pos_dtype], mask_flat.data_ptr(), keep.data_ptr(),
                                    pos.data_ptr(), vr.data_ptr(), enc_ref.data_ptr(), prop.data_ptr(), _lib.current_stream())
    _lib.check(code, "dtlr_geometry")
    return dict(mask_flat=mask_flat, keep=keep, pos=pos, valid_ratios=vr, enc_ref=enc_ref, proposals=prop)


def msda_encoder_fits(level_hw, dtype, halo: Optional[int] = None) -> bool:
    """Whether the LDS-window encoder kernel's plan fits these level shapes (it stages full-height column windows: canvases
    taller than ~270 px in fp32 / ~550 px in bf16 do not fit, and the caller uses msda_fused, the gather kernel, instead)."""
    hw = (ctypes.c_int * 8)(*[int(v) for pair in level_hw for v in pair])
    rc = _L(dtype).dtlr_msda_encoder_plan_ok(ctypes.cast(hw, ctypes.c_void_p), _DT[dtype], MSDA_HALO if halo is None else int(halo))
    if rc < 0:
        _lib.check(rc, "dtlr_msda_encoder_plan_ok")
    return rc == 1


def msda_encoder(value, level_hw, ow, ref, halo: Optional[int] = None):
    """Encoder MSDA (Lq == S, queries are the level pixels) with LDS-staged value windows.
    value [N,S,M,32] fp32/bf16; level_hw: HOST list of (H_l, W_l); ow [N,S,M*48]; ref [N,S,4,2] fp32."""
    N, S, M, D = value.shape
    assert len(level_hw) == 4 and sum(h * w for h, w in level_hw) == S and ow.shape[1] == S and D == 32
    assert value.is_contiguous() and ow.is_contiguous() and ref.is_contiguous() and ref.dtype == torch.float32
    import ctypes
    hw = (ctypes.c_int * 8)(*[int(v) for pair in level_hw for v in pair])
    out = torch.empty((N, S, M * D), dtype=value.dtype, device=value.device)
    ev = MSDA_EVENTS
    if ev is not None:
        st = torch.cuda.current_stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
    code = _L(value).dtlr_msda_encoder_forward(value.data_ptr(), ow.data_ptr(), ref.data_ptr(), hw, N, M, D, 4, 4, MSDA_HALO if halo is None else int(halo),
                                                _DT[value.dtype], _DT[ow.dtype], out.data_ptr(), _lib.current_stream())
    _lib.check(code, "dtlr_msda_encoder_forward")
    if ev is not None:
        b.record(st)
        ev.append((a, b, N, S, S))
    return out


def msda_encoder_far_fraction(value_dtype, level_hw, ow, ref, n_heads: int = 8, halo: Optional[int] = None):
    """Fraction of the in-map sampling points of an encoder MSDA call that msda_encoder would fetch through its global path (outside the
    staged column window of the query's tile; dtlr_msda_encoder_far_samples).  Synchronises (reads two counters back): a probe."""
    require_cuda(ow, "ow")
    assert len(level_hw) == 4 and ow.is_contiguous() and ref.is_contiguous() and ref.dtype == torch.float32
    import ctypes
    hw = (ctypes.c_int * 8)(*[int(v) for pair in level_hw for v in pair])
    counts = torch.zeros(2, dtype=torch.int64, device=ow.device)
    code = _L(value_dtype).dtlr_msda_encoder_far_samples(ow.data_ptr(), ref.data_ptr(), hw, ow.shape[0], n_heads, MSDA_HALO if halo is None else int(halo), _DT[value_dtype],
                                                    _DT[ow.dtype], counts.data_ptr(), _lib.current_stream())
    _lib.check(code, "dtlr_msda_encoder_far_samples")
    far, inside = counts.tolist()
    return far / max(inside, 1)


def mha(qk, v, n_heads: int, split: bool = False):
    """Self-attention core.  qk [B, L, 2C] (projected q | k), v [B, L, C] -> [B, L, C].  Fused flash-style HIP
    kernel on the matrix cores, scores never leave the chip: bf16 / fp16 (mfma 16x16x32), exact fp32 (mfma 16x16x4), or -- fp32 tensors
    with split=True -- fp32 operands as fp16 hi + lo halves, three fp16 MFMAs per product (DTLR_F32S: the split-fp32 engine)."""
    B, L, C2 = qk.shape
    C = C2 // 2
    hd = C // n_heads
    if qk.dtype not in H16 + (torch.float32,) or hd != 32:
        raise RuntimeError(f"dtlr_amd.ops.mha: unsupported dtype/head_dim {qk.dtype}/{hd}")
    qk, v = qk.contiguous(), v.contiguous()
    L_ = _L(qk)
    ws = torch.empty(L_.dtlr_mha_workspace_bytes(B, L, n_heads, hd), dtype=torch.uint8, device=qk.device)
    out = torch.empty((B, L, C), dtype=qk.dtype, device=qk.device)
    if split and qk.dtype != torch.float32:
        raise RuntimeError("dtlr_amd.ops.mha: split=True takes fp32 tensors")
    code = L_.dtlr_mha_forward(qk.data_ptr(), v.data_ptr(), ws.data_ptr(), out.data_ptr(), B, L, n_heads, hd,
                               _lib.DTLR_F32S if split else _DT[qk.dtype], _lib.current_stream())
    _lib.check(code, "dtlr_mha_forward")
    return out


_DIM_T = {}


def decoder_query_prep(ref, valid_ratios, out_dtype):
    """ref [B,nq,4] fp32, valid_ratios [B,L,2] fp32 -> (ref_in [B,nq,L,4] fp32, sine [B,nq,512] out_dtype)."""
    B, nq, _ = ref.shape
    L = valid_ratios.shape[1]
    key = ref.device
    if key not in _DIM_T:       # the exact table gen_sineembed_for_position builds (models/dino/utils.py:145-146)
        t = torch.arange(128, dtype=torch.float32, device=ref.device)
        _DIM_T[key] = (10000 ** (2 * torch.div(t, 2, rounding_mode="floor") / 128)).contiguous()
    ref = ref.contiguous()
    valid_ratios = valid_ratios.contiguous()
    ref_in = torch.empty((B, nq, L, 4), dtype=torch.float32, device=ref.device)
    sine = torch.empty((B, nq, 512), dtype=out_dtype, device=ref.device)
    code = _L(out_dtype).dtlr_decoder_query_prep(ref.data_ptr(), valid_ratios.data_ptr(), _DIM_T[key].data_ptr(), ref_in.data_ptr(),
                                              sine.data_ptr(), B, nq, L, _DT[out_dtype], _lib.current_stream())
    _lib.check(code, "dtlr_decoder_query_prep")
    return ref_in, sine


def dq_pack(w):
    """[N, K] weight (N, K multiples of 32) -> the fragment-order image dtlr_dec_query_stage streams (== dtlr_dq_pack_weights):
    [unit = N/32][k-step = K/32][tile 2][lane (m, g)][8] <- W[32 u + 16 t + m][32 ks + 8 g ..]."""
    N, K = w.shape
    assert N % 32 == 0 and K % 32 == 0
    v = w.detach().to(_hdt(w)).view(N // 32, 2, 16, K // 32, 4, 8)             # u, t, m, ks, g, e
    return v.permute(0, 3, 1, 4, 2, 5).contiguous().view(-1)                   # u, ks, t, [g, m] = lane, e


def dec_query_stage(ref, valid_ratios, tgt, w0, b0, w1, b1, wqk, bqk, wv, bv):
    """The query stage of a decoder layer in ONE launch (dtlr_dec_query_stage, 16-bit engines): reference boxes per level, sine
    embedding, ref_point_head MLP, and the q | k (on tgt + query_pos) and v (on tgt) input projections of the self-attention.
    ref [B,nq,4] fp32, valid_ratios [B,L,2] fp32, tgt [B,nq,256] 16-bit; weights = dq_pack of the 16-bit [256,512] / [256,256] /
    [512,256] / [256,256] matrices, biases fp32 -> (ref_in [B,nq,L,4] fp32, qpos [B,nq,256], qk [B,nq,512], v [B,nq,256])."""
    require_cuda(tgt, "tgt")
    B, nq, C = tgt.shape
    L = valid_ratios.shape[1]
    assert tgt.dtype in H16 and C == 256 and ref.dtype == torch.float32 and tuple(ref.shape) == (B, nq, 4)
    assert w0.numel() == 256 * 512 and w1.numel() == 256 * 256 and wqk.numel() == 512 * 256 and wv.numel() == 256 * 256
    assert all(t.dtype == tgt.dtype and t.is_contiguous() for t in (w0, w1, wqk, wv))
    key = ref.device
    if key not in _DIM_T:
        t = torch.arange(128, dtype=torch.float32, device=ref.device)
        _DIM_T[key] = (10000 ** (2 * torch.div(t, 2, rounding_mode="floor") / 128)).contiguous()
    ref, valid_ratios, tgt = ref.contiguous(), valid_ratios.contiguous(), tgt.contiguous()
    dev = tgt.device
    ref_in = torch.empty((B, nq, L, 4), dtype=torch.float32, device=dev)
    qpos = torch.empty((B, nq, 256), dtype=tgt.dtype, device=dev)
    qk = torch.empty((B, nq, 512), dtype=tgt.dtype, device=dev)
    v = torch.empty((B, nq, 256), dtype=tgt.dtype, device=dev)
    M = B * nq
    with _Timed("gemm_bf16", 2.0 * M * (512 * 256 + 256 * 256 + 256 * 512 + 256 * 256), float(M) * (256 + 256 + 512 + 256) * 2 + 393216.0 * 2,
                f"dec_query_stage M{M}"):
        code = _L(tgt).dtlr_dec_query_stage(ref.data_ptr(), valid_ratios.data_ptr(), _DIM_T[key].data_ptr(), tgt.data_ptr(),
                                            w0.data_ptr(), b0.data_ptr(), w1.data_ptr(), b1.data_ptr(), wqk.data_ptr(), bqk.data_ptr(),
                                            wv.data_ptr(), bv.data_ptr(), ref_in.data_ptr(), qpos.data_ptr(), qk.data_ptr(), v.data_ptr(),
                                            B, nq, L, _DT[tgt.dtype], _lib.current_stream())
    _lib.check(code, "dtlr_dec_query_stage")
    return ref_in, qpos, qk, v


def box_mlp_refine(x, w1, b1, w2p, b2, w3, b3, ref, mode: int = 0):
    """3-layer box MLP + refinement in ONE launch (dtlr_box_mlp_refine_bf16): x [..,256] bf16, W1 [256,256] bf16,
    w2p = ffn_pack_w2(W2) bf16, W3 [4,256] / biases fp32, ref [..,4] fp32 -> [..,4] fp32.
    mode 0: sigmoid(mlp(x) + inverse_sigmoid(ref)); mode 1: mlp(x) + ref."""
    require_cuda(x, "x")
    assert x.dtype in H16 and x.shape[-1] == 256 and ref.dtype == torch.float32 and w3.dtype == torch.float32
    x = x if x.is_contiguous() else x.contiguous()
    ref = ref if ref.is_contiguous() else ref.contiguous()
    out = torch.empty_like(ref)
    code = _L(x).dtlr_box_mlp_refine_bf16(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), w3.data_ptr(),
                                               b3.data_ptr(), ref.data_ptr(), out.data_ptr(), x.numel() // 256, mode, _lib.current_stream())
    _lib.check(code, "dtlr_box_mlp_refine_bf16")
    return out


def box_head_refine(h, w, b, ref, mode: int = 0):
    """Last layer of the box MLP (256 -> 4) fused with its consumer (HIP kernel, one wavefront per row):
    mode 0: sigmoid(h W^T + b + inverse_sigmoid(ref)) ; mode 1: h W^T + b + ref.  h [..,256] fp32, ref [..,4] fp32."""
    require_cuda(h, "h")
    assert h.dtype == torch.float32 and ref.dtype == torch.float32 and w.dtype == torch.float32 and h.shape[-1] == 256
    h = h if h.is_contiguous() else h.contiguous()
    ref = ref if ref.is_contiguous() else ref.contiguous()
    out = torch.empty_like(ref)
    code = _L(h).dtlr_box_head_refine(h.data_ptr(), w.data_ptr(), b.data_ptr(), ref.data_ptr(), out.data_ptr(),
                                           h.numel() // 256, 256, mode, _lib.current_stream())
    _lib.check(code, "dtlr_box_head_refine")
    return out


def box_refine(delta, ref):
    """sigmoid(delta + inverse_sigmoid(ref)) (fp32)."""
    delta, ref = delta.contiguous(), ref.contiguous()
    out = torch.empty_like(ref)
    code = _L(delta).dtlr_box_refine(delta.data_ptr(), ref.data_ptr(), out.data_ptr(), ref.numel(), _lib.current_stream())
    _lib.check(code, "dtlr_box_refine")
    return out


def topk_rows(scores, k: int):
    """Indices [B,k] int64 of the k largest per row, descending, ties -> lower index (two-stage selection,
    deformable_transformer.py:345).  Scores are always fp32.  HIP kernel: per-row bitonic sort in LDS."""
    B, S = scores.shape
    scores = scores.float().contiguous()
    idx = torch.empty((B, k), dtype=torch.int64, device=scores.device)
    code = _L(scores).dtlr_topk_rows(scores.data_ptr(), idx.data_ptr(), B, S, k, _lib.current_stream())
    _lib.check(code, "dtlr_topk_rows")
    return idx


def decode_blank(logits, boxes, eps: float):
    """Blank/argmax decoder (HIP kernel): logits [B,nq,C], boxes [B,nq,4] -> (labels [B,nq] int32 left-packed
    -1 padded, lengths [B] int32)."""
    require_cuda(logits, "pred_logits")
    B, nq, C = logits.shape
    logits = logits.float().contiguous()
    boxes = boxes.float().contiguous()
    labels = torch.empty((B, nq), dtype=torch.int32, device=logits.device)
    lengths = torch.empty((B,), dtype=torch.int32, device=logits.device)
    code = _L(logits).dtlr_decode_blank(logits.data_ptr(), boxes.data_ptr(), labels.data_ptr(), lengths.data_ptr(), B, nq, C,
                                        float(eps), _lib.current_stream())
    _lib.check(code, "dtlr_decode_blank")
    return labels, lengths


def blank_emissions(logits, boxes, eps: float, scale: float = 1.0):
    """[B, nq, C+1] CTC-style emissions, blank channel first, queries in reading order (dtlr_blank_emissions: per-query sigmoid sums
    chip-wide, the decoders' cx sort per line, one wave per output row) -- get_new_pred_logits (ngram/prediction_helpers.py:5-46) /
    the blank construction of loss_CTC (dino.py:466-502)."""
    require_cuda(logits, "pred_logits")
    logits = logits.float().contiguous()
    boxes = boxes.float().contiguous()
    B, nq, C = logits.shape
    L_ = _lib.lib()
    ws = torch.empty(L_.dtlr_blank_emissions_workspace_bytes(B, nq) // 4, dtype=torch.float32, device=logits.device)
    out = torch.empty((B, nq, C + 1), dtype=torch.float32, device=logits.device)
    code = L_.dtlr_blank_emissions(logits.data_ptr(), boxes.data_ptr(), out.data_ptr(), ws.data_ptr(), B, nq, C, float(scale), float(eps),
                                   _lib.current_stream())
    _lib.check(code, "dtlr_blank_emissions")
    return out


def preprocess_lines(src_u8, offsets, dims, Hc: int, Wc: int, max_downscale: float, mean, std):
    """Resize + ToTensor + Normalize + pad of a batch of uint8 RGB images in ONE launch (dtlr_preprocess_lines).
    src_u8: flat uint8 CUDA tensor (images back to back, HWC); offsets [B] int64 CUDA; dims [B,4] int32 CUDA = (h, w, oh, ow).
    Returns (canvas [B,3,Hc,Wc] fp32, mask [B,Hc,Wc] bool)."""
    require_cuda(src_u8, "src_u8")
    assert src_u8.dtype == torch.uint8 and offsets.dtype == torch.int64 and dims.dtype == torch.int32 and dims.shape[1] == 4
    B = dims.shape[0]
    canvas = torch.empty((B, 3, Hc, Wc), dtype=torch.float32, device=src_u8.device)
    mask = torch.empty((B, Hc, Wc), dtype=torch.bool, device=src_u8.device)
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    code = _L(src_u8).dtlr_preprocess_lines(src_u8.data_ptr(), offsets.data_ptr(), dims.data_ptr(), B, Hc, Wc, float(max_downscale),
                                            ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p),
                                            canvas.data_ptr(), mask.data_ptr(), _lib.current_stream())
    _lib.check(code, "dtlr_preprocess_lines")
    return canvas, mask


def ctc_loss_interleaved(logits, boxes, targets, target_lengths, max_target_length: int, eps: float = 0.003, filler: float = 1e-5):
    """Per-line CTC negative log-likelihood of the reference's evaluation loss (dtlr_ctc_loss_interleaved): logits [B,nq,C],
    boxes [B,nq,4], targets [B,Lmax] int32 (label + 1), target_lengths [B] int32 (CUDA) -> nll [B] fp32 (inf -> 0)."""
    require_cuda(logits, "pred_logits")
    B, nq, C = logits.shape
    logits = logits.float().contiguous()
    boxes = boxes.float().contiguous()
    assert targets.dtype == torch.int32 and target_lengths.dtype == torch.int32 and targets.is_cuda and target_lengths.is_cuda
    targets = targets.contiguous()
    Lmax = targets.shape[1]
    nll = torch.empty((B,), dtype=torch.float32, device=logits.device)
    ws = torch.empty((B * nq,), dtype=torch.float32, device=logits.device)
    code = _L(logits).dtlr_ctc_loss_interleaved(logits.data_ptr(), boxes.data_ptr(), targets.data_ptr() if Lmax > 0 else None,
                                                target_lengths.data_ptr(), nll.data_ptr(), ws.data_ptr(), B, nq, C, Lmax,
                                                int(max_target_length), float(eps), float(filler), _lib.current_stream())
    _lib.check(code, "dtlr_ctc_loss_interleaved")
    return nll


def nms_batched(boxes, scores, iou_threshold: float):
    """Greedy NMS per image on the device (dtlr_nms): boxes [B,n,4] xyxy fp32, scores [B,n] fp32 -> (keep [B,n] int64: kept
    original indices in descending score order, -1 padded; counts [B] int32).  n <= 1024."""
    require_cuda(boxes, "boxes")
    B, n, _ = boxes.shape
    boxes = boxes.float().contiguous()
    scores = scores.float().contiguous()
    keep = torch.empty((B, n), dtype=torch.int64, device=boxes.device)
    counts = torch.empty((B,), dtype=torch.int32, device=boxes.device)
    code = _L(boxes).dtlr_nms(boxes.data_ptr(), scores.data_ptr(), float(iou_threshold), keep.data_ptr(), counts.data_ptr(), B, n,
                               _lib.current_stream())
    _lib.check(code, "dtlr_nms")
    return keep, counts


def topk_flat(x, k: int, apply_sigmoid: bool = False):
    """Per-row top-k of a long [B, n] fp32 matrix (dtlr_topk_flat: exact radix select over the row in global memory + a 1024-key sort):
    -> (values [B,k] fp32 descending, indices [B,k] int64; equal values: lower index first).  k <= 8192."""
    require_cuda(x, "x")
    assert x.dim() == 2
    x = x.float().contiguous()
    B, n = x.shape
    values = torch.empty((B, k), dtype=torch.float32, device=x.device)
    idx = torch.empty((B, k), dtype=torch.int64, device=x.device)
    code = _L(x).dtlr_topk_flat(x.data_ptr(), values.data_ptr(), idx.data_ptr(), B, n, int(k), int(bool(apply_sigmoid)), _lib.current_stream())
    _lib.check(code, "dtlr_topk_flat")
    return values, idx


# --------------------------------------------------------------------------------------------
# Every operator launches on the current stream of the device its FIRST tensor argument lives on (not of whatever device
# happens to be current): a model moved to cuda:1 while cuda:0 is current would otherwise launch on device 0 with device-1
# pointers.  The check costs one integer compare per call when the devices already agree.
def _device_scoped(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for t in args:
            if isinstance(t, torch.Tensor):
                if t.is_cuda and t.device.index != torch.cuda.current_device():
                    with torch.cuda.device(t.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapper


for _name in ("msda_encoder_far_fraction", "gemm_kres", "gemm_kres_chain", "gemm_kres_cat_s2", "gemm_kres_bcast384", "ffn32", "ffn4", "proj_ln_k256", "swin_patch_embed", "swin_window_attn", "swin_patch_merge", "geometry", "linear", "gemm_k256", "linear_rowmax", "two_stage_gather", "layernorm", "proj_ln", "proj_ln_split", "ffn_fused", "conv2d_nhwc", "stem_conv7x7", "stem_conv7x7_f32",
              "maxpool_nhwc", "groupnorm_tokens", "msda", "msda_fused", "msda_encoder", "mha", "decoder_query_prep", "box_mlp_refine",
              "box_head_refine", "box_refine", "topk_rows", "decode_blank", "preprocess_lines", "ctc_loss_interleaved", "nms_batched",
              "topk_flat", "stem_conv7x7_pool", "dec_query_stage", "blank_emissions", "split_pack", "linear_resbcast", "ffn_split", "stem_conv7x7_f32s", "k256s_pack", "gemm_k256s"):
    globals()[_name] = _device_scoped(globals()[_name])
del _name
