"""Numerics of the individual gfx950 kernels against plain fp32 CPU references of the same op."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import msda_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def half(request):
    """the 16-bit format under test: bf16 (libdtlr_hip.so) and IEEE fp16 (libdtlr_hip_f16.so, the same sources)"""
    return request.param


def ulp(half, bf16_exp):
    """2^-bf16_exp for bf16 results, 8x tighter for fp16 (three more significand bits): tolerances below are written for bf16"""
    return 2.0 ** -(bf16_exp + (3 if half == torch.float16 else 0))


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.Generator(np.random.PCG64(seed)).standard_normal(shape) * scale).astype(np.float32))


@pytest.mark.parametrize("C", [256, 512, 2048])
@pytest.mark.parametrize("with_res", [False, True])
def test_layernorm_f32(C, with_res):
    from dtlr_amd import ops
    x, r = _rand((3, 37, C), 1, 2.0) + 0.5, _rand((3, 37, C), 2)
    w, b = _rand((C,), 3) * 0.2 + 1.0, _rand((C,), 4) * 0.1
    want = F.layer_norm(x + r if with_res else x, (C,), w, b, 1e-5)
    got = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-5, r.cuda() if with_res else None).cpu()
    assert (got - want).abs().max() < 5e-6


def test_layernorm_bf16_and_ragged_rows(half):
    from dtlr_amd import ops
    C = 256
    for rows in (1, 3, 4, 5, 1023):
        x, r = _rand((rows, C), rows, 2.0).to(half), _rand((rows, C), rows + 1).to(half)
        w, b = _rand((C,), 3) * 0.2 + 1.0, _rand((C,), 4) * 0.1
        want = F.layer_norm(x.float() + r.float(), (C,), w, b, 1e-5)
        got = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-5, r.cuda()).cpu()
        assert got.dtype == half and got.shape == x.shape
        assert (got.float() - want).abs().max() < 0.03       # one bf16 ulp at |y| <= 4


@pytest.mark.parametrize("ref_dim", [2, 4])
def test_msda_fused_front_end_vs_oracle(ref_dim, half):
    """dtlr_msda_fused_forward == MSDeformAttn.forward lines 97-124 (softmax, locations, sampling)."""
    from dtlr_amd import ops
    from oracle import dtlr_oracle as O
    shapes = [(16, 256), (8, 128), (4, 64), (2, 32)]
    N, M, D, Lq, L, P = 2, 8, 32, 211, 4, 4
    v, s, lsi, _, _ = msda_inputs(N, M, D, Lq, P, shapes, seed=3)
    ow = _rand((N, Lq, M * L * P * 3), 5)
    ow[..., : M * L * P * 2] *= 3.0                                        # offsets of a few pixels
    g = np.random.Generator(np.random.PCG64(9))
    if ref_dim == 2:
        ref = torch.from_numpy(g.uniform(-0.05, 1.05, (N, Lq, L, 2)).astype(np.float32))
    else:
        ref = torch.from_numpy(np.concatenate([g.uniform(0, 1, (N, Lq, L, 2)), g.uniform(0.01, 0.4, (N, Lq, L, 2))], -1).astype(np.float32))
    off = ow[..., : M * L * P * 2].view(N, Lq, M, L, P, 2)
    aw = torch.softmax(ow[..., M * L * P * 2:].view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    loc = O.msda_sampling_locations(ref, off, s, P)
    want = O.ms_deform_attn_core(v, s, loc, aw)
    got = ops.msda_fused(v.cuda(), s.cuda(), lsi.cuda(), ow.cuda(), ref.cuda()).cpu()
    assert (got - want).abs().max() < 5e-6
    # bf16 value (+ fp32 or bf16 projection row)
    gotb = ops.msda_fused(v.to(half).cuda(), s.cuda(), lsi.cuda(), ow.cuda(), ref.cuda()).cpu()
    wantb = O.ms_deform_attn_core(v.to(half).float(), s, loc, aw)
    assert (gotb.float() - wantb).abs().max() <= wantb.abs().max() * ulp(half, 8) + 1e-6
    gotbb = ops.msda_fused(v.to(half).cuda(), s.cuda(), lsi.cuda(), ow.to(half).cuda(), ref.cuda()).cpu()
    owb = ow.to(half).float()
    offb = owb[..., : M * L * P * 2].view(N, Lq, M, L, P, 2)
    awb = torch.softmax(owb[..., M * L * P * 2:].view(N, Lq, M, L * P), -1).view(N, Lq, M, L, P)
    wantbb = O.ms_deform_attn_core(v.to(half).float(), s, O.msda_sampling_locations(ref, offb, s, P), awb)
    assert (gotbb.float() - wantbb).abs().max() <= wantbb.abs().max() * ulp(half, 8) + 1e-5


@pytest.mark.parametrize("B,L", [(2, 900), (1, 37), (3, 128), (1, 1)])
def test_mha_bf16_vs_fp32_reference(B, L, half):
    """Fused attention kernel vs plain fp32 softmax(QK^T/sqrt(d))V on the same bf16-rounded inputs.
    Tolerance: P and O are rounded to bf16 (2^-8 relative) -> |err| <= 2^-6 * max|v|."""
    import math
    from dtlr_amd import ops
    H, hd = 8, 32
    C = H * hd
    qk = (_rand((B, L, 2 * C), 11) * 1.5).to(half)
    v = _rand((B, L, C), 12).to(half)
    q = qk[..., :C].float().view(B, L, H, hd).transpose(1, 2)
    k = qk[..., C:].float().view(B, L, H, hd).transpose(1, 2)
    vv = v.float().view(B, L, H, hd).transpose(1, 2)
    want = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1) @ vv).transpose(1, 2).reshape(B, L, C)
    got = ops.mha(qk.cuda(), v.cuda(), H).float().cpu()
    assert got.shape == want.shape
    assert (got - want).abs().max() <= ulp(half, 6) * v.float().abs().max() + 1e-3, (got - want).abs().max()


def test_mha_softmax_extremes(half):
    """Large score spread (online-softmax rescale path) and a dominant key."""
    import math
    from dtlr_amd import ops
    B, L, H, hd = 1, 200, 8, 32
    C = H * hd
    qk = (_rand((B, L, 2 * C), 21) * 4.0)
    qk[:, 150, C:] *= 6.0                       # one key with huge scores late in the sequence
    qk = qk.to(half)
    v = _rand((B, L, C), 22).to(half)
    q = qk[..., :C].float().view(B, L, H, hd).transpose(1, 2)
    k = qk[..., C:].float().view(B, L, H, hd).transpose(1, 2)
    vv = v.float().view(B, L, H, hd).transpose(1, 2)
    want = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1) @ vv).transpose(1, 2).reshape(B, L, C)
    got = ops.mha(qk.cuda(), v.cuda(), H).float().cpu()
    assert torch.isfinite(got).all()
    assert (got - want).abs().max() <= ulp(half, 6) * v.float().abs().max() + 1e-3


def _gemm_ref(x, w, b, relu, res, a2, mask):
    a = x if a2 is None else x + a2
    y = a.double() @ w.double().t()
    if b is not None:
        y = y + b.double()
    if relu == 1:
        y = y.clamp(min=0)
    if mask is not None:
        y = y.masked_fill(mask[..., None], 0.0)
    if res is not None:
        y = y + res.double()
    if relu == 2:
        y = y.clamp(min=0)
    return y.float()


GEMM_SHAPES = [(128, 128, 32), (300, 166, 256), (129, 384, 64), (1, 4, 256), (1000, 2048, 256), (777, 256, 2048),
               (5440, 256, 256), (64, 7356, 256)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_f32_exact_mfma(M, N, K):
    """fp32-in MFMA GEMM (v_mfma_f32_16x16x4_f32) vs an fp64 reference; all epilogue/prologue combos.
    Includes M/N tails (N = 166, 4, 7356 are not tile multiples) -- asymmetric random operands."""
    from dtlr_amd import ops
    x, a2 = _rand((M, K), 1), _rand((M, K), 2)
    w, b = _rand((N, K), 3) / np.sqrt(K), _rand((N,), 4)
    res = _rand((M, N), 5)
    mask = torch.from_numpy(np.random.Generator(np.random.PCG64(6)).random(M) < 0.3)
    for relu, use_b, use_res, use_a2, use_mask in ((0, False, False, False, False), (1, True, False, False, False),
                                                   (0, True, True, True, True), (2, True, True, False, False)):
        want = _gemm_ref(x, w, b if use_b else None, relu, res if use_res else None, a2 if use_a2 else None, mask if use_mask else None)
        got = ops.linear(x.cuda(), w.cuda(), b.cuda() if use_b else None, relu, res.cuda() if use_res else None,
                         a2.cuda() if use_a2 else None, mask.cuda() if use_mask else None).cpu()
        assert got.shape == want.shape
        assert (got - want).abs().max() < 2e-5 * max(1.0, want.abs().max().item()), (relu, use_b, use_res, use_a2, use_mask)


@pytest.mark.parametrize("M,N,K", [s for s in GEMM_SHAPES if s[2] % 64 == 0])
@pytest.mark.parametrize("out_f32", [False, True])
def test_gemm_bf16(M, N, K, out_f32, half):
    """bf16 MFMA GEMM, fp32 accumulation: vs fp64 reference on the same bf16-rounded operands.
    Tolerance: bf16 rounding of the a+a2 prologue (2^-8 relative per element, averaged over K) and of
    the bf16 output (2^-8 relative)."""
    from dtlr_amd import ops
    x, a2 = _rand((M, K), 1).to(half), _rand((M, K), 2).to(half)
    w, b = (_rand((N, K), 3) / np.sqrt(K)).to(half), _rand((N,), 4)
    od = torch.float32 if out_f32 else half
    res = _rand((M, N), 5).to(od)
    mask = torch.from_numpy(np.random.Generator(np.random.PCG64(6)).random(M) < 0.3)
    for relu, use_b, use_res, use_a2, use_mask in ((0, False, False, False, False), (1, True, False, False, False),
                                                   (0, True, True, True, True), (2, True, True, False, False)):
        xa = (x.float() + a2.float()).to(half).float() if use_a2 else x.float()
        want = _gemm_ref(xa, w.float(), b if use_b else None, relu, res.float() if use_res else None, None, mask if use_mask else None)
        got = ops.linear(x.cuda(), w.cuda(), b.cuda() if use_b else None, relu, res.cuda() if use_res else None,
                         a2.cuda() if use_a2 else None, mask.cuda() if use_mask else None, out_dtype=od).cpu()
        assert got.dtype == od and got.shape == want.shape
        scale = max(1.0, want.abs().max().item())
        tol = (2e-5 if out_f32 else ulp(half, 8)) * scale + 1e-4
        assert (got.float() - want).abs().max() < tol, (relu, use_b, use_res, use_a2, use_mask, (got.float() - want).abs().max().item())


@pytest.mark.parametrize("level_hw,offscale", [([(16, 256), (8, 128), (4, 64), (2, 32)], 2.0),
                                               ([(16, 256), (8, 128), (4, 64), (2, 32)], 40.0),     # far offsets: global path
                                               ([(5, 83), (3, 42), (2, 21), (1, 11)], 3.0),          # odd sizes (not exact halves)
                                               ([(4, 32), (2, 16), (1, 8), (1, 4)], 1.5),
                                               ([(1, 7), (1, 4), (1, 2), (1, 1)], 1.0)])
def test_msda_encoder_lds_vs_oracle(level_hw, offscale, half):
    """LDS-staged encoder kernel == oracle MSDeformAttn (softmax + locations + sampling) for every query
    of every level, including samples that leave the staged window (global path) and level shapes that
    are not exact halves; checked for the default halo and halo 0 (window = tile only)."""
    from dtlr_amd import ops
    from oracle import dtlr_oracle as O
    N, M, D, L, P = 2, 8, 32, 4, 4
    S = sum(h * w for h, w in level_hw)
    v, s, lsi, _, _ = msda_inputs(N, M, D, S, P, level_hw, seed=13)
    ow = _rand((N, S, M * L * P * 3), 15)
    ow[..., : M * L * P * 2] *= offscale
    # encoder reference points with a valid-ratio-like scaling (deformable_transformer.py:479-492)
    vr = torch.tensor([[[1.0, 1.0]] * 4, [[0.75, 1.0]] * 4])
    ref = O.encoder_reference_points(s, vr).contiguous()
    off = ow[..., : M * L * P * 2].view(N, S, M, L, P, 2)
    aw = torch.softmax(ow[..., M * L * P * 2:].view(N, S, M, L * P), -1).view(N, S, M, L, P)
    want = O.ms_deform_attn_core(v, s, O.msda_sampling_locations(ref, off, s, P), aw)
    old = ops.MSDA_HALO
    try:
        for halo in (8, 0):
            ops.MSDA_HALO = halo
            got = ops.msda_encoder(v.cuda(), level_hw, ow.cuda(), ref.cuda()).cpu()
            assert (got - want).abs().max() < 5e-6, (halo, (got - want).abs().max().item())
        ops.MSDA_HALO = 8
        gotb = ops.msda_encoder(v.to(half).cuda(), level_hw, ow.cuda(), ref.cuda()).float().cpu()
        wantb = O.ms_deform_attn_core(v.to(half).float(), s, O.msda_sampling_locations(ref, off, s, P), aw)
        # the 16-bit query phase accumulates each level's 16 corner terms in packed fp16 (~2^-11 per term) whatever the storage format:
        # bf16 results are dominated by their own output rounding (2^-9), fp16 results by that accumulation (a few 2^-11)
        assert (gotb - wantb).abs().max() <= wantb.abs().max() * (ulp(half, 8) if half == torch.bfloat16 else 2.0 ** -9) + 1e-6
        # the BENCHED instantiation -- 16-bit value AND 16-bit [offsets|logits] row, 512 threads, packed-fp16 accumulation -- at three value
        # scales: 1e-3 (products near fp16's subnormal range: accumulated as fp16 pairs, so the absolute error floor is ~2^-24 per term)
        # and 1e3 (sums of 16 terms up to 1e3: far inside fp16's range of 65504)
        owh = ow.to(half)
        offh = owh.float()[..., : M * L * P * 2].view(N, S, M, L, P, 2)
        awh = torch.softmax(owh.float()[..., M * L * P * 2:].view(N, S, M, L * P), -1).view(N, S, M, L, P)
        for vs in (1e-3, 1.0, 1e3):
            vv = (v * vs).to(half)
            goth = ops.msda_encoder(vv.cuda(), level_hw, owh.cuda(), ref.cuda()).float().cpu()
            wanth = O.ms_deform_attn_core(vv.float(), s, O.msda_sampling_locations(ref, offh, s, P), awh)
            # bf16: output rounding alone reaches max * 2^-8 (half an ulp just above a power of two); the fp16 accumulation adds a few 2^-11
            # fp16: the running sums of a level's 16 corner terms are rounded to fp16 at every step (half an ulp = 2^-12 of the sum each):
            # measured up to 3.0 x 2^-11 of the maximum
            tol = wanth.abs().max() * (ulp(half, 8) + 2.0 ** -9 if half == torch.bfloat16 else 2.0 ** -8) + 16 * 2.0 ** -24
            assert (goth - wanth).abs().max() <= tol, (vs, (goth - wanth).abs().max().item(), tol.item())
        # must agree with the gather kernel bit-for-bit in fp32 (same arithmetic order)
        g2 = ops.msda_fused(v.cuda(), s.cuda(), lsi.cuda(), ow.cuda(), ref.cuda()).cpu()
        assert (g2 - got).abs().max() < 1e-6
    finally:
        ops.MSDA_HALO = old


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad", [(2, 16, 40, 64, 64, 3, 1, 1), (3, 9, 33, 128, 128, 3, 2, 1),
                                                         (1, 4, 64, 2048, 256, 3, 2, 1), (2, 5, 7, 32, 48, 3, 1, 1),
                                                         (2, 6, 10, 64, 96, 1, 1, 0)])
def test_conv2d_nhwc_implicit_gemm(B, H, W, Cin, Cout, k, stride, pad, half):
    """Implicit-GEMM NHWC convolution vs torch CPU conv2d (fp64 reference), with bias / residual / ReLU;
    includes stride 2, odd sizes (tile tails) and zero padding taps."""
    import torch.nn.functional as F
    from dtlr_amd import ops
    x = _rand((B, H, W, Cin), 1)
    w = _rand((Cout, Cin, k, k), 2) / np.sqrt(Cin * k * k)
    b = _rand((Cout,), 3)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    res = _rand(tuple(ref.shape), 4)
    w_ohwi = w.permute(0, 2, 3, 1).contiguous()
    for relu, use_res in ((False, False), (True, False), (True, True)):
        want = ref + res.double() if use_res else ref
        want = want.clamp(min=0) if relu else want
        got = ops.conv2d_nhwc(x.cuda(), w_ohwi.cuda(), b.cuda(), stride, pad, relu, res.cuda() if use_res else None).cpu()
        assert got.shape == want.shape
        assert (got - want.float()).abs().max() < 3e-5 * max(1.0, want.abs().max().item()), (relu, use_res)
    if (Cin * 2) % 128 == 0:
        xb, wb = x.to(half), w_ohwi.to(half)
        refb = F.conv2d(xb.float().permute(0, 3, 1, 2).double(), wb.float().permute(0, 3, 1, 2).double(), b.double(),
                        stride=stride, padding=pad).permute(0, 2, 3, 1).clamp(min=0).float()
        gotb = ops.conv2d_nhwc(xb.cuda(), wb.cuda(), b.cuda(), stride, pad, True, None).float().cpu()
        assert (gotb - refb).abs().max() < ulp(half, 8) * max(1.0, refb.abs().max().item()) + 1e-4


def test_splitk_convolutions_overlapped_on_two_streams():
    """Round 5: the split-K workspace is per (device, stream).  Two DIFFERENT split-K convolutions (input_proj[3]'s shape: 3x3 / 2 on a
    4 x 64 map, K = 18432) issued back to back on two HIP streams, repeatedly, with nothing ordering the streams: each must equal its own
    single-stream result bit for bit (with one workspace per device the second launch's partial tiles overwrote the first's between
    its two kernels whenever the streams overlapped)."""
    from dtlr_amd import ops
    xs = [_rand((1, 4, 64, 2048), 11 + i).cuda() for i in range(2)]
    ws = [(_rand((256, 3, 3, 2048), 21 + i) / np.sqrt(9 * 2048)).cuda() for i in range(2)]
    bs = [_rand((256,), 31 + i).cuda() for i in range(2)]
    want = [ops.conv2d_nhwc(xs[i], ws[i], bs[i], 2, 1, False, None) for i in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for st in streams:                                             # each stream's workspace slot is allocated here, outside the race
        with torch.cuda.stream(st):
            ops.conv2d_nhwc(xs[0], ws[0], bs[0], 2, 1, False, None)
    torch.cuda.synchronize()
    bad = 0
    for rep in range(40):
        outs = []
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs.append(ops.conv2d_nhwc(xs[i], ws[i], bs[i], 2, 1, False, None))
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(outs[i], want[i])) for i in range(2))
    assert bad == 0, f"{bad} of 80 overlapped launches differ from their single-stream result"


@pytest.mark.parametrize("case", ["encoder3", "decoder6_masked", "one_ragged_relu_res"])
def test_gemm_k256s_multi_vs_fp64_and_single_slice_kernel(case):
    """dtlr_gemm_k256s_multi (round 6: ONE pass over A for several projections of it) against fp64 on the same fp32 operands and against
    dtlr_gemm_k256s slice by slice (same MFMA order per output: bit-identical where both exist):
      encoder3            -- value_proj | offsets (256) | logits (128, zero-padded image) with the position term as a row-broadcast residual
                             [res_rows, 384] (position-major tile walk), bias only on the first slice, outputs as column views of two buffers;
      decoder6_masked     -- six slices of one [M, 1536] buffer, row mask on all of them, ragged M;
      one_ragged_relu_res -- one slice, full residual [M, 256], ReLU, M not a multiple of 64."""
    from dtlr_amd import ops
    dd = lambda t: t.double()                                       # noqa: E731
    if case == "encoder3":
        S, B = 320, 5
        M = S * B
        x = (_rand((B, S, 256), 61, 1.5) + 0.2).cuda()
        wv, bv = _rand((256, 256), 62) / 16.0, _rand((256,), 63) * 0.5
        wo = _rand((384, 256), 64) / 16.0
        res = _rand((S, 384), 65, 2.0).cuda()
        value = torch.empty((B, S, 256), device="cuda")
        ow = torch.full((B, S, 384), float("nan"), device="cuda")
        wpad = torch.cat([wo[256:], torch.zeros((128, 256))], 0)
        sl = [dict(wp=ops.k256s_pack(wv.cuda()), out=value, bias=bv.cuda()),
              dict(wp=ops.k256s_pack(wo[:256].contiguous().cuda()), out=ow[..., :256], residual=res[:, :256]),
              dict(wp=ops.k256s_pack(wpad.cuda()), out=ow[..., 256:], residual=res[:, 256:])]
        ops.gemm_k256s_multi(x, sl, res_rows=S)
        torch.cuda.synchronize()
        xc = x.cpu()
        want_v = (dd(xc) @ dd(wv).t() + dd(bv)).float()
        want_o = (dd(xc) @ dd(wo).t() + dd(res.cpu())[None]).float()
        assert torch.isfinite(ow).all()
        assert (value.cpu() - want_v).abs().max() < 2e-5 * max(1.0, want_v.abs().max().item())
        assert (ow.cpu() - want_o).abs().max() < 2e-5 * max(1.0, want_o.abs().max().item())
        assert torch.equal(value, ops.gemm_k256s(x, sl[0]["wp"], bv.cuda()))          # same products in the same order
    elif case == "decoder6_masked":
        M = 2 * 333
        x = (_rand((2, 333, 256), 71, 1.5) - 0.1).cuda()
        w, b = _rand((1536, 256), 72) / 16.0, _rand((1536,), 73) * 0.5
        mask = (torch.arange(M) % 6 == 1).cuda()
        out = torch.full((2, 333, 1536), float("nan"), device="cuda")
        bc = b.cuda()
        imgs = [ops.k256s_pack(w[256 * j:256 * (j + 1)].contiguous().cuda()) for j in range(6)]
        ops.gemm_k256s_multi(x, [dict(wp=imgs[j], out=out[..., 256 * j:256 * (j + 1)], bias=bc[256 * j:256 * (j + 1)]) for j in range(6)], row_mask=mask)
        torch.cuda.synchronize()
        want = (dd(x.cpu()) @ dd(w).t() + dd(b)).masked_fill(mask.cpu().view(2, 333, 1), 0.0).float()
        assert torch.isfinite(out).all()
        assert (out.cpu() - want).abs().max() < 2e-5 * max(1.0, want.abs().max().item())
        for j in (0, 5):
            assert torch.equal(out[..., 256 * j:256 * (j + 1)], ops.gemm_k256s(x, imgs[j], bc[256 * j:256 * (j + 1)].contiguous(), row_mask=mask))
        nomask = torch.empty_like(out)
        ops.gemm_k256s_multi(x, [dict(wp=imgs[j], out=nomask[..., 256 * j:256 * (j + 1)], bias=bc[256 * j:256 * (j + 1)]) for j in range(6)])
        keep = ~mask.view(2, 333)
        assert torch.equal(nomask[keep], out[keep]) and bool((out[~keep] == 0).all())
    else:
        M = 64 * 300 + 29                                            # more tiles than workgroups, ragged tail
        x = (_rand((M, 256), 81, 1.5)).cuda()
        w, b, r = _rand((256, 256), 82) / 16.0, _rand((256,), 83) * 0.5, _rand((M, 256), 84, 2.0)
        out = torch.empty((M, 256), device="cuda")
        ops.gemm_k256s_multi(x, [dict(wp=ops.k256s_pack(w.cuda()), out=out, bias=b.cuda(), residual=r.cuda(), relu=True)])
        torch.cuda.synchronize()
        want = (dd(x.cpu()) @ dd(w).t() + dd(b) + dd(r)).clamp(min=0).float()
        assert (out.cpu() - want).abs().max() < 2e-5 * max(1.0, want.abs().max().item())
        assert bool((out >= 0).all())


def test_workspace_growth_never_invalidates_a_captured_graph():
    """Round 6: a workspace pointer baked into a captured HIP graph must stay valid when a later, larger request grows
    that stream's workspace.  On a fresh stream: a small split-K convolution runs eagerly (allocates the slot), the same launch is captured,
    then a 10x larger split-K convolution runs eagerly on the same stream (round 5 hipFree()d the small buffer here), then other
    allocations churn the heap, and the graph is replayed: its output must equal the eager result bit for bit, and the library must
    report the replaced buffer as retired (kept), not freed.  dtlr_workspace_reserve refuses to run under capture."""
    from dtlr_amd import _lib, ops
    x = _rand((1, 4, 64, 2048), 41).cuda()
    w = (_rand((256, 3, 3, 2048), 42) / np.sqrt(9 * 2048)).cuda()
    b = _rand((256,), 43).cuda()
    xbig = _rand((12, 4, 64, 2048), 44).cuda()
    want = ops.conv2d_nhwc(x, w, b, 2, 1, False, None).clone()
    retired0 = ops.workspace_retired_bytes()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for _ in range(2):
            ops.conv2d_nhwc(x, w, b, 2, 1, False, None)            # eager: this stream's slot comes to exist at the small size
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=st):
        out = ops.conv2d_nhwc(x, w, b, 2, 1, False, None)
        rc = _lib.lib().dtlr_workspace_reserve(1 << 30, _lib.current_stream())
    assert rc == -1                                                # DTLR_EINVAL: nothing is allocated while capturing
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    with torch.cuda.stream(st):
        big = ops.conv2d_nhwc(xbig, w, b, 2, 1, False, None)       # 12x the partial tiles: the slot grows
        ops.workspace_reserve(torch.float32, 64 << 20)              # ... and again
    torch.cuda.synchronize()
    assert ops.workspace_retired_bytes() > retired0, "the replaced buffer must be retired (kept allocated), not freed"
    junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(64)]      # would land in a freed buffer's pages
    torch.cuda.synchronize()
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want), "replay after the workspace grew differs from the eager result"
    assert bool(big.isfinite().all())
    del junk


def test_decoder_query_prep_and_box_refine_vs_oracle(half):
    """Fused decoder glue == oracle gen_sineembed_for_position / reference scaling / inverse_sigmoid refinement."""
    from dtlr_amd import ops
    from oracle import dtlr_oracle as O
    g = np.random.Generator(np.random.PCG64(3))
    B, nq, L = 3, 37, 4
    ref = torch.from_numpy(g.uniform(0, 1, (B, nq, 4)).astype(np.float32))
    ref[0, 0] = torch.tensor([0.0, 1.0, 0.0005, 0.9999])             # inverse_sigmoid clamps
    vr = torch.from_numpy(g.uniform(0.5, 1.0, (B, L, 2)).astype(np.float32))
    ref_in, sine = ops.decoder_query_prep(ref.cuda(), vr.cuda(), torch.float32)
    want_in = ref[:, :, None] * torch.cat([vr, vr], -1)[:, None]
    assert (ref_in.cpu() - want_in).abs().max() < 1e-7
    want_sine = O.gen_sineembed_for_position(want_in[:, :, 0, :])
    assert (sine.cpu() - want_sine).abs().max() < 2e-6
    _, sine_b = ops.decoder_query_prep(ref.cuda(), vr.cuda(), half)
    assert (sine_b.float().cpu() - want_sine).abs().max() < ulp(half, 8)
    delta = torch.from_numpy(g.standard_normal((B, nq, 4)).astype(np.float32))
    got = ops.box_refine(delta.cuda(), ref.cuda()).cpu()
    want = (delta + O.inverse_sigmoid(ref)).sigmoid()
    assert (got - want).abs().max() < 1e-6



def test_box_head_refine_vs_reference():
    """Fused 256->4 output layer + refinement (mode 0) / + proposals (mode 1) vs fp64 Linear + the oracle's
    inverse_sigmoid/sigmoid (deformable_transformer.py:734-756, 352-356)."""
    from dtlr_amd import ops
    from oracle import dtlr_oracle as O
    rows = 2 * 900 + 3
    h = torch.relu(_rand((rows, 256), 1))
    w, b = _rand((4, 256), 2) / 16, _rand((4,), 3) * 0.1
    g = np.random.Generator(np.random.PCG64(5))
    ref = torch.from_numpy(g.uniform(-0.1, 1.1, (rows, 4)).astype(np.float32))
    delta = (h.double() @ w.double().t() + b.double())
    want0 = torch.sigmoid(delta + O.inverse_sigmoid(ref).double()).float()
    got0 = ops.box_head_refine(h.cuda(), w.cuda(), b.cuda(), ref.cuda(), mode=0).cpu()
    assert (got0 - want0).abs().max() < 2e-6
    want1 = (delta + ref.double()).float()
    got1 = ops.box_head_refine(h.cuda(), w.cuda(), b.cuda(), ref.cuda(), mode=1).cpu()
    assert (got1 - want1).abs().max() < 1e-5 * max(1.0, want1.abs().max().item())


@pytest.mark.parametrize("B,T", [(2, 4096), (3, 37), (1, 1), (2, 128)])
def test_groupnorm_tokens(B, T, half):
    import torch.nn.functional as F
    from dtlr_amd import ops
    x = _rand((B, T, 256), 1, 2.0) + 0.7
    w, b = _rand((256,), 2) * 0.2 + 1.0, _rand((256,), 3) * 0.1
    want = F.group_norm(x.double().transpose(1, 2), 32, w.double(), b.double(), 1e-5).transpose(1, 2).float()
    got = ops.groupnorm_tokens(x.cuda(), 32, w.cuda(), b.cuda()).cpu()
    assert (got - want).abs().max() < 2e-5
    xb = x.to(half)
    wantb = F.group_norm(xb.double().transpose(1, 2), 32, w.double(), b.double(), 1e-5).transpose(1, 2).float()
    gotb = ops.groupnorm_tokens(xb.cuda(), 32, w.cuda(), b.cuda()).float().cpu()
    assert (gotb - wantb).abs().max() < ulp(half, 7) * max(1.0, wantb.abs().max().item())


@pytest.mark.parametrize("B,H,W,C", [(2, 64, 96, 64), (1, 7, 9, 64), (2, 1, 1, 8), (1, 5, 4, 16)])
def test_maxpool_nhwc(B, H, W, C, half):
    import torch.nn.functional as F
    from dtlr_amd import ops
    x = _rand((B, H, W, C), 5)
    want = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(ops.maxpool_nhwc(x.cuda()).cpu(), want)
    xb = x.to(half)
    wantb = F.max_pool2d(xb.float().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(ops.maxpool_nhwc(xb.cuda()).float().cpu(), wantb)
    # fused stem tail: maxpool(relu(x + bias)) -- bit-exact in fp32 (max and +bias commute under monotone rounding)
    b = _rand((C,), 6)
    wantf = F.max_pool2d(torch.relu(x + b).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(ops.maxpool_nhwc(x.cuda(), bias=b.cuda(), relu=True).cpu(), wantf)
    wantfb = F.max_pool2d(torch.relu(xb.float() + b).permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).to(half).float()
    assert torch.equal(ops.maxpool_nhwc(xb.cuda(), bias=b.cuda(), relu=True).float().cpu(), wantfb)


@pytest.mark.parametrize("B,S,k", [(2, 5440, 900), (3, 172, 30), (1, 7, 7), (2, 6800, 900)])
def test_topk_rows(B, S, k):
    """Per-row top-k indices: descending scores, ties -> lower index; also with a block of exactly tied
    scores and +/-inf entries."""
    from dtlr_amd import ops
    sc = _rand((B, S), 7)
    if S > 100:
        sc[0, 10:60] = sc[0, 5]                 # exact ties
        sc[0, 3] = float("inf")
        sc[0, 4] = float("-inf")
        if B > 1:
            sc[1, :] = 0.25                     # a whole row of ties: the cut falls INSIDE the tie block (radix select on (score, index))
            sc[1, S // 2:] = -0.5
            if B > 2:
                sc[2, ::3] = sc[2, 0]           # ties interleaved with distinct values across the cut
    got = ops.topk_rows(sc.cuda(), k).cpu()
    order = torch.sort(sc, dim=1, descending=True, stable=True)[1][:, :k]       # stable: ties keep the lower index
    assert torch.equal(got, order)


@pytest.mark.parametrize("C,bias", [(23, -5.0), (23, -1.0), (166, -6.0), (166, -3.0), (7356, -9.5)])
def test_decode_blank_kernel_vs_oracle(C, bias):
    """HIP blank decoder == oracle (evaluation.py:116-158 / dino.py:466-502) on both branches of the blank rule,
    both eps conventions, including rows where every query is blank."""
    from dtlr_amd import evaluation as E
    from oracle import dtlr_oracle as O
    g = np.random.Generator(np.random.PCG64(C + int(-bias * 10)))
    B, nq = 3, 900 if C < 1000 else 120
    out = {"pred_logits": torch.from_numpy((g.standard_normal((B, nq, C)) + bias).astype(np.float32)),
           "pred_boxes": torch.from_numpy(g.uniform(0.02, 0.98, (B, nq, 4)).astype(np.float32))}
    out["pred_logits"][0, :, :] -= 6.0                                        # line 0: everything blank
    dev = {k: v.cuda() for k, v in out.items()}
    for eps in (None, 0.003):
        labels, lengths = E.decode_blank_records(dev, eps)
        got = E.records_to_lists(labels, lengths)
        want = O.decode_blank(out, eps)
        assert got == want
        assert (labels.cpu()[torch.arange(nq)[None, :] >= lengths.cpu()[:, None]] == -1).all()


@pytest.mark.parametrize("B,L", [(2, 900), (1, 37), (1, 1)])
def test_mha_f32_vs_fp64_reference(B, L):
    """Exact-fp32 MFMA attention kernel (parity path) vs an fp64 softmax(QK^T/sqrt(d))V."""
    import math
    from dtlr_amd import ops
    H, hd = 8, 32
    C = H * hd
    qk, v = _rand((B, L, 2 * C), 31) * 1.5, _rand((B, L, C), 32)
    q = qk[..., :C].double().view(B, L, H, hd).transpose(1, 2)
    k = qk[..., C:].double().view(B, L, H, hd).transpose(1, 2)
    vv = v.double().view(B, L, H, hd).transpose(1, 2)
    want = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1) @ vv).transpose(1, 2).reshape(B, L, C).float()
    got = ops.mha(qk.cuda(), v.cuda(), H).cpu()
    assert (got - want).abs().max() < 2e-5


@pytest.mark.parametrize("M,dff", [(128, 2048), (300, 2048), (1000, 512), (4097, 2048), (37, 64), (4097, 512),
                                   (50152, 128), (89152, 128), (40000, 128), (33000, 2048), (49153, 512), (900, 2048), (5440, 2048), (12289, 2048)])
def test_ffn_fused_bf16_vs_reference(M, dff, half):
    """Fused FFN + residual + LayerNorm (dtlr_ffn_fused_bf16) vs an fp64 restatement of
    norm(x + linear2(relu(linear1(x)))) (deformable_transformer.py:804-823) on the same bf16-rounded
    inputs, with the intermediate rounded to bf16 as the kernel (and the unfused path) does; and vs the
    unfused HIP path (two GEMMs + LayerNorm).  Ragged M exercises the token-tile tail.  Both kernel structures are covered:
    the first structure up to M = 32768 and for d_ff = 64; above that the second one (whole rounds of 192-token workgroups + a
    128- or 192-token remainder launch).  Round 5: up to 96 tiles (M <= 12288) with d_ff >= 512 the first structure runs split over the hidden
    dimension (2 .. 8 parts of >= 8 chunks + a finish kernel: the single-line latency case -- M = 900 and 5440 are one line's decoder /
    encoder call); M = 12289 is the first size back on the unsplit kernel.  (The product library reads no environment variables: the dispatch is a function of the
    shape only.)"""
    from dtlr_amd import ops
    x = _rand((M, 256), 1).to(half)
    w1 = (_rand((dff, 256), 2) / 16).to(half)
    w2 = (_rand((256, dff), 3) / np.sqrt(dff)).to(half)
    b1, b2 = _rand((dff,), 4) * 0.5, _rand((256,), 5) * 0.5
    gw, gb = 1 + 0.2 * _rand((256,), 6), 0.3 * _rand((256,), 7)
    h = torch.relu(x.double() @ w1.double().t() + b1.double()).to(half).double()
    pre = x.double() + h @ w2.double().t() + b2.double()
    want = torch.nn.functional.layer_norm(pre, (256,), gw.double(), gb.double(), 1e-5).float()
    got = ops.ffn_fused(x.cuda(), w1.cuda(), b1.cuda(), ops.ffn_pack_w2(w2.cuda()), b2.cuda(), gw.cuda(), gb.cuda()).float().cpu()
    assert got.shape == want.shape
    # bf16 output rounding (2^-9 relative) + occasional 1-ulp flips of the bf16 intermediate
    tol = ulp(half, 8) * max(1.0, want.abs().max().item()) + 2e-2
    assert (got - want).abs().max() < tol, (got - want).abs().max().item()
    assert (got - want).abs().mean() < 4e-3
    hh = ops.linear(x.cuda(), w1.cuda(), b1.cuda(), relu=True)
    un = ops.layernorm(ops.linear(hh, w2.cuda(), b2.cuda()), gw.cuda(), gb.cuda(), 1e-5, residual=x.cuda()).float().cpu()
    assert (got - un).abs().max() < 2 * tol


@pytest.mark.parametrize("B,H,W", [(2, 128, 2048), (1, 37, 531), (3, 8, 16), (1, 128, 2560)])
def test_stem_conv7x7_vs_reference(B, H, W, half):
    """Own stem kernel (7x7/s2/p3, 3->64, NCHW fp32 in, NHWC bf16 out) vs torch conv2d in fp64 on the same
    bf16-rounded image and weights; odd sizes exercise the zero padding and the column/row tails."""
    import torch.nn.functional as F
    from dtlr_amd import ops
    x = _rand((B, 3, H, W), 1)
    w = _rand((64, 3, 7, 7), 2) / 12
    frag = ops.stem_pack_weights(w, half)
    got = ops.stem_conv7x7(x.cuda(), frag.cuda(), half).float().cpu()
    want = F.conv2d(x.to(half).double(), w.to(half).double(), None, stride=2, padding=3).permute(0, 2, 3, 1).float()
    assert got.shape == want.shape
    assert (got - want).abs().max() <= ulp(half, 8) * max(1.0, want.abs().max().item()) + 1e-5


@pytest.mark.parametrize("M", [128, 300, 4097, 37])
def test_proj_ln_bf16_vs_reference(M, half):
    """Fused output projection + residual + LayerNorm (dtlr_proj_ln_bf16) vs fp64 on the same bf16 inputs, and vs the
    unfused HIP path (GEMM with residual epilogue + LayerNorm kernel)."""
    from dtlr_amd import ops
    a, r = _rand((M, 256), 1).to(half), _rand((M, 256), 2).to(half)
    w = (_rand((256, 256), 3) / 16).to(half)
    b = _rand((256,), 4) * 0.5
    gw, gb = 1 + 0.2 * _rand((256,), 6), 0.3 * _rand((256,), 7)
    pre = r.double() + a.double() @ w.double().t() + b.double()
    want = torch.nn.functional.layer_norm(pre, (256,), gw.double(), gb.double(), 1e-5).float()
    got = ops.proj_ln(a.cuda(), ops.proj_pack_w(w.cuda()), b.cuda(), r.cuda(), gw.cuda(), gb.cuda()).float().cpu()
    tol = ulp(half, 8) * max(1.0, want.abs().max().item()) + 1e-3
    assert (got - want).abs().max() < tol, (got - want).abs().max().item()
    un = ops.layernorm(ops.linear(a.cuda(), w.cuda(), b.cuda()), gw.cuda(), gb.cuda(), 1e-5, residual=r.cuda()).float().cpu()
    assert (got - un).abs().max() < 3 * tol


@pytest.mark.parametrize("mode", [0, 1])
def test_box_mlp_refine_bf16_vs_reference(mode, half):
    """One-launch box MLP (256->256->256->4) + refinement vs fp64 on bf16-rounded inputs/weights (hidden layer 1 rounded to
    bf16 as the kernel does), and vs the three-launch HIP path."""
    from dtlr_amd import ops
    from oracle import dtlr_oracle as O
    M = 2 * 900 + 5
    x = _rand((M, 256), 1).to(half)
    w1, w2 = (_rand((256, 256), 2) / 16).to(half), (_rand((256, 256), 3) / 16).to(half)
    b1, b2 = _rand((256,), 4) * 0.3, _rand((256,), 5) * 0.3
    w3, b3 = _rand((4, 256), 6) / 16, _rand((4,), 7) * 0.1
    g = np.random.Generator(np.random.PCG64(5))
    ref = torch.from_numpy(g.uniform(-0.1, 1.1, (M, 4)).astype(np.float32))
    h1 = torch.relu(x.double() @ w1.double().t() + b1.double()).to(half).double()
    h2 = torch.relu(h1 @ w2.double().t() + b2.double())
    delta = h2 @ w3.double().t() + b3.double()
    want = (torch.sigmoid(delta + O.inverse_sigmoid(ref).double()) if mode == 0 else delta + ref.double()).float()
    got = ops.box_mlp_refine(x.cuda(), w1.cuda(), b1.cuda(), ops.ffn_pack_w2(w2.cuda()), b2.cuda(), w3.cuda(), b3.cuda(), ref.cuda(), mode).cpu()
    tol = 2e-3 if mode == 0 else 2e-2                       # bf16 rounding of the first hidden layer (flips of 1 ulp)
    assert (got - want).abs().max() < tol, (got - want).abs().max().item()
    assert (got - want).abs().mean() < tol / 20


@pytest.mark.parametrize("size,max_size", [(800, 1333), (20, 96), (48, 200), (33, None)])
def test_preprocess_lines_bit_exact(size, max_size):
    """dtlr_preprocess_lines (resize + ToTensor + Normalize + pad + mask, one launch) == the oracle's restatement of the
    reference's eval transform + collate (datasets/transforms.py:78-109,247-249,552-559; util/misc.py:375-397): fp32 canvas
    and mask BIT-EXACT (the resize is integer arithmetic; the normalisation two correctly rounded fp32 divisions).
    Ragged batch: down- and up-scaling on either axis, identity sizes, a square, a portrait, a grey-scale image."""
    from dtlr_amd import transforms as T
    from oracle import dtlr_oracle as O
    from tests.util import preproc_image
    shapes = [(128, 2048), (64, 900), (40, 40), (200, 120), (17, 1999), (31, 97), (9, 14)] if size == 800 else \
             [(24, 160), (31, 97), (12, 300), (40, 40), (9, 14), (64, 48), (33, 50), (50, 33)]
    imgs = [preproc_image(h, w, 300 + i) for i, (h, w) in enumerate(shapes)]
    grey = preproc_image(30, 70, 999)[:, :, 0].copy()
    want_x, want_m = O.preprocess_lines(imgs + [np.repeat(grey[:, :, None], 3, axis=2)], size, max_size)
    nt = T.preprocess_lines(imgs + [grey], size, max_size)
    assert nt.tensors.is_cuda and nt.tensors.dtype == torch.float32 and nt.mask.dtype == torch.bool
    assert torch.equal(nt.mask.cpu(), want_m)
    assert torch.equal(nt.tensors.cpu(), want_x)


def test_preprocess_lines_golden_and_errors(golden_dir):
    """Against the committed Pillow vectors (tests/golden/g4_preproc.npz) and the C-ABI's shape limits."""
    from dtlr_amd import transforms as T
    from tests.util import preproc_image
    g = np.load(os.path.join(golden_dir, "g4_preproc.npz"))
    mean = torch.tensor(T.IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(T.IMAGENET_STD).view(3, 1, 1)
    for k, (seed, h, w, size, max_size, oh, ow) in enumerate(g["small_cases"].tolist()):
        nt = T.preprocess_lines([preproc_image(h, w, seed)], size, None if max_size < 0 else max_size)
        want = (torch.from_numpy(g[f"s{k}_out"]).permute(2, 0, 1).float() / 255 - mean) / std
        assert tuple(nt.tensors.shape) == (1, 3, oh, ow) and not nt.mask.any()
        assert torch.equal(nt.tensors[0].cpu(), want), (seed, h, w, size)
    with pytest.raises(RuntimeError):                       # 13x down-scaling: outside the kernel's filter footprint
        T.preprocess_lines([preproc_image(400, 30, 1)], 2, None)


def test_ctc_loss_vs_oracle_and_reference_golden(golden_dir):
    """dtlr_ctc_loss_interleaved (+ the 'mean' reduction in evaluation.loss_ctc) against (a) the oracle, which runs torch's own
    CTCLoss on CPU exactly as SetCriterion.loss_CTC does (models/dino/dino.py:457-551), and (b) the values the REAL reference
    criterion produced (tests/golden/g5_ctc.npz).  fp32 log-space recursion over 2 nq steps: tolerance 1e-4 relative."""
    from dtlr_amd import evaluation as E
    from oracle import dtlr_oracle as O
    from tests.util import ctc_case
    g = np.load(os.path.join(golden_dir, "g5_ctc.npz"))
    for k, (seed, B, nq, C, bias, lmax) in enumerate(g["cases"].tolist()):
        outputs, labels = ctc_case(int(seed), int(B), int(nq), int(C), bias, int(lmax))
        dev = {kk: v.cuda() for kk, v in outputs.items()}
        got = E.loss_ctc(dev, labels).item()
        want = O.loss_ctc(outputs, labels).item()
        assert abs(got - want) <= 1e-4 * max(1.0, abs(want)), (k, got, want)
        assert abs(got - float(g[f"loss_{k}"])) <= 1e-4 * max(1.0, abs(float(g[f"loss_{k}"]))), (k, got)


def test_ctc_loss_edge_cases():
    """Impossible alignments give 0 (zero_infinity), empty transcriptions are scored against the all-blank path, repeated
    characters need the filler blank, per-line values match torch's CTCLoss(reduction='none')."""
    from dtlr_amd import ops
    from oracle import dtlr_oracle as O
    from tests.util import ctc_case
    outputs, _ = ctc_case(11, 4, 12, 7, -2.0, 5)
    labels = [[1, 1, 1, 1], [], [0, 6, 3], list(range(7)) * 4]          # the last: 28 labels > 24 steps -> infeasible
    want_each = []
    for b in range(4):
        one = {k: v[b:b + 1] for k, v in outputs.items()}
        want_each.append(O.loss_ctc(one, [labels[b]]).item() * max(len(labels[b]), 1))
    Lmax = max(len(l) for l in labels)
    tt = torch.zeros((4, Lmax), dtype=torch.int32)
    for i, l in enumerate(labels):
        tt[i, : len(l)] = torch.tensor(l, dtype=torch.int32) + 1
    tl = torch.tensor([len(l) for l in labels], dtype=torch.int32)
    nll = ops.ctc_loss_interleaved(outputs["pred_logits"].cuda(), outputs["pred_boxes"].cuda(), tt.cuda(), tl.cuda(), Lmax).cpu()
    assert nll[3].item() == 0.0 and want_each[3] == 0.0
    for b in range(3):
        assert abs(nll[b].item() - want_each[b]) <= 1e-4 * max(1.0, abs(want_each[b])), (b, nll[b].item(), want_each[b])
    with pytest.raises(RuntimeError):                                   # 2 L + 1 > 1024 states
        ops.ctc_loss_interleaved(outputs["pred_logits"].cuda(), outputs["pred_boxes"].cuda(),
                                 torch.ones((4, 600), dtype=torch.int32).cuda(), torch.full((4,), 600, dtype=torch.int32).cuda(), 600)


def test_evaluate_ctc_step_matches_oracle():
    from dtlr_amd import evaluation as E
    from oracle import dtlr_oracle as O
    from tests.util import ctc_case
    outputs, labels = ctc_case(21, 3, 40, 23, -3.0, 15)
    r = E.evaluate_ctc_step({k: v.cuda() for k, v in outputs.items()}, labels)
    preds = O.decode_blank(outputs, 0.003)
    want_cer = sum(O.character_error_rate_engine(p, l) for p, l in zip(preds, labels))
    assert r["n"] == 3 and abs(r["cer_sum"] - want_cer) < 1e-12
    assert abs(r["loss_CTC"] - O.loss_ctc(outputs, labels).item()) <= 1e-4 * max(1.0, r["loss_CTC"])


@pytest.mark.parametrize("M", [128, 300, 4097])
def test_proj_ln_split_and_split_head(M, half):
    """Two-stage front end of the bf16 engine: dtlr_proj_ln_split_bf16 = LayerNorm(Linear(masked a)) as [hi | lo | hi], and the
    class head on [W_hi | W_hi | W_lo].  (1) hi + lo == the fp64 LayerNorm of the same bf16 inputs up to fp32-accumulation noise;
    (2) images 0 and 2 are identical and lo is at most half a bf16 ulp of hi; (3) the split-product scores equal
    (hi + lo) W^T + b computed in fp64 to ~2^-16 relative -- the precision the fp32 MFMA head was kept for; (4) padded head rows
    come out as -inf so a max over the padded width is the max over the real classes."""
    from dtlr_amd import ops
    a = _rand((M, 256), 1).to(half)
    w = (_rand((256, 256), 3) / 16).to(half)
    b = _rand((256,), 4) * 0.5
    gw, gb = 1 + 0.2 * _rand((256,), 6), 0.3 * _rand((256,), 7)
    keep = (torch.arange(M) % 7 != 3)
    pre = (a.double() * keep[:, None]) @ w.double().t() + b.double()
    want = torch.nn.functional.layer_norm(pre, (256,), gw.double(), gb.double(), 1e-5)
    y3 = ops.proj_ln_split(a.cuda(), ops.proj_pack_w(w.cuda()), b.cuda(), keep.cuda(), gw.cuda(), gb.cuda())
    assert tuple(y3.shape) == (M, 768)
    hi, lo, hi2 = y3[:, :256].cpu(), y3[:, 256:512].cpu(), y3[:, 512:].cpu()
    assert torch.equal(hi, hi2)
    rec = hi.double() + lo.double()
    assert (lo.double().abs() <= hi.double().abs() * ulp(half, 8) + 1e-7).all()       # lo is the rounding residue of hi: at most half an ulp (fp16: + its subnormal floor)
    assert (rec - want).abs().max() < 2e-3 * max(1.0, want.abs().max().item())
    assert (rec[~keep] - rec[~keep][0]).abs().max() == 0            # masked rows: LayerNorm of the bias alone, all identical
    C = 166
    hw, hb = _rand((C, 256), 8) / 8, _rand((C,), 9) - 4.6
    w3, b3 = ops.split_head_weight(hw.cuda(), hb.cuda(), dtype=half)
    assert tuple(w3.shape) == (192, 768) and torch.isinf(b3[C:]).all()
    sc = ops.linear(y3, w3, b3, out_dtype=torch.float32).cpu()
    ref = rec @ hw.double().t() + hb.double()
    assert (sc[:, :C].double() - ref).abs().max() < 3e-5 * max(1.0, ref.abs().max().item())
    assert torch.isinf(sc[:, C:]).all() and (sc[:, C:] < 0).all()
    assert torch.equal(sc.max(-1)[0], sc[:, :C].max(-1)[0])
    # without a mask
    y3n = ops.proj_ln_split(a.cuda(), ops.proj_pack_w(w.cuda()), b.cuda(), None, gw.cuda(), gb.cuda())
    wantn = torch.nn.functional.layer_norm(a.double() @ w.double().t() + b.double(), (256,), gw.double(), gb.double(), 1e-5)
    assert ((y3n[:, :256].double() + y3n[:, 256:512].double()).cpu() - wantn).abs().max() < 2e-3 * max(1.0, wantn.abs().max().item())


@pytest.mark.parametrize("B,n,thr", [(3, 900, 0.5), (2, 300, 0.2), (1, 1, 0.5), (2, 65, 0.7), (1, 1024, 0.3)])
def test_nms_batched_vs_oracle(B, n, thr):
    """dtlr_nms == the oracle's greedy NMS (torchvision.ops.nms semantics, models/dino/dino.py:1029-1033): kept indices in the
    same order, on text-line-like boxes (neighbours overlap heavily), with blocks of exactly equal scores (stable order) and
    degenerate zero-area boxes (IoU 0/0 = NaN never suppresses)."""
    from dtlr_amd import ops
    from oracle import dtlr_oracle as O
    g = np.random.Generator(np.random.PCG64(n + int(thr * 100)))
    cx = np.sort(g.uniform(0.02, 0.98, (B, n)), axis=1)
    w = g.uniform(0.004, 0.03, (B, n))
    cy, h = g.uniform(0.4, 0.6, (B, n)), g.uniform(0.3, 0.8, (B, n))
    boxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1).astype(np.float32)
    scores = g.uniform(0, 1, (B, n)).astype(np.float32)
    if n > 60:
        scores[0, 10:40] = scores[0, 5]                     # ties: the lower index wins
        boxes[0, 50] = boxes[0, 51]                          # identical boxes
        boxes[-1, 7, 2:] = boxes[-1, 7, :2]                  # zero area
    bt, st = torch.from_numpy(boxes), torch.from_numpy(scores)
    keep, counts = ops.nms_batched(bt.cuda(), st.cuda(), thr)
    keep, counts = keep.cpu(), counts.cpu()
    for b in range(B):
        want = O.nms(bt[b], st[b], thr)
        assert counts[b].item() == len(want)
        assert torch.equal(keep[b, : len(want)], want)
        assert (keep[b, len(want):] == -1).all()
    with pytest.raises(RuntimeError):
        ops.nms_batched(torch.zeros((1, 1025, 4)).cuda(), torch.zeros((1, 1025)).cuda(), 0.5)


@pytest.mark.parametrize("B,nq,C,k", [(3, 900, 166, 900), (2, 60, 23, 50), (1, 300, 7356, 300), (2, 5, 3, 15), (1, 1000, 300, 1024)])
def test_topk_flat_and_postprocess_vs_oracle(B, nq, C, k):
    """dtlr_topk_flat == a stable descending sort of the flattened logits (ties: lower index first; blocks of exactly tied values
    straddle the cut), values = sigmoid; and PostProcess built on it (models/dino/dino.py:985-1046) == the oracle's: labels, boxes,
    scores for the reference's two uses (num_select = 50 with image sizes; num_select = 900 on a (1,1) canvas)."""
    from dtlr_amd import ops
    from dtlr_amd.dino import PostProcess
    from oracle import dtlr_oracle as O
    g = np.random.Generator(np.random.PCG64(nq + C))
    logits = torch.from_numpy((g.standard_normal((B, nq, C)) * 2 - 3).astype(np.float32))
    flat = logits.view(B, -1)
    if flat.shape[1] > 200:
        flat[0, 100:180] = flat[0, 7]                       # an exact tie block
        flat[-1, ::5] = 0.25                                # many ties, possibly across the cut
    vals, idx = ops.topk_flat(flat.cuda(), k, apply_sigmoid=True)
    order = torch.sort(flat, dim=1, descending=True, stable=True)[1][:, :k]
    assert torch.equal(idx.cpu(), order)
    assert (vals.cpu() - torch.gather(flat, 1, order).sigmoid()).abs().max() < 1e-6
    raw, idx2 = ops.topk_flat(flat.cuda(), k)
    assert torch.equal(idx2.cpu(), order) and torch.equal(raw.cpu(), torch.gather(flat, 1, order))
    if k <= nq:
        boxes = torch.from_numpy(g.uniform(0.05, 0.95, (B, nq, 4)).astype(np.float32))
        out = {"pred_logits": logits, "pred_boxes": boxes}
        ts = torch.tensor([[100.0, 200.0]] * B)
        got = PostProcess(num_select=k)({kk: v.cuda() for kk, v in out.items()}, ts.cuda())
        want = O.post_process(out, ts, k)
        checked = []
        for a, w in zip(got, want):
            # sets of (box, label) agree wherever the scores are not tied; compare through a stable re-sort on the oracle's probabilities
            assert torch.allclose(a["scores"].cpu(), w["scores"], atol=1e-6)
            tie_free = (w["scores"][1:] != w["scores"][:-1])
            same = (a["labels"].cpu() == w["labels"]) & (a["boxes"].cpu() - w["boxes"]).abs().amax(-1).lt(1e-4)
            keep = torch.ones_like(same)
            nxt = torch.topk(out["pred_logits"][len(checked)].sigmoid().view(-1), min(k + 1, nq * C))[0]
            if nxt.numel() > k and nxt[k] == nxt[k - 1]:
                keep[-1] = False                            # the tie straddles the cut: the last member is unspecified
            checked.append(0)
            keep[1:] &= tie_free
            keep[:-1] &= tie_free
            assert same[keep].all()


# ---- round 2 kernels --------------------------------------------------------------------------------------------------------
def _geometry_masks(B, H, W, kind, seed):
    g = np.random.Generator(np.random.PCG64(seed))
    m = torch.zeros((B, H, W), dtype=torch.bool)
    if kind == "rect":                                   # what collate produces: a valid top-left rectangle per image
        for b in range(B):
            h, w = (H, W) if b == 0 else (int(g.integers(H // 2, H + 1)), int(g.integers(W // 3, W + 1)))
            m[b, h:, :] = True
            m[b, :, w:] = True
    elif kind == "random":                               # arbitrary masks: the kernel does real cumulative sums
        m = torch.from_numpy(g.random((B, H, W)) < 0.3)
        m[:, 0, 0] = False
    return m


@pytest.mark.parametrize("H,W,kind", [(128, 2048, "none"), (128, 2048, "rect"), (128, 2560, "rect"), (64, 200, "random"), (448, 1344, "rect")])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_geometry_kernel_vs_oracle(H, W, kind, dtype):
    """dtlr_geometry == the oracle's restatements of interpolate_mask / PositionEmbeddingSineHW / get_valid_ratio /
    get_reference_points / gen_encoder_output_proposals, for unpadded, collate-style and arbitrary masks (incl. a tall canvas and a
    level wider than one 256-token chunk)."""
    from dtlr_amd import ops
    from oracle import dtlr_oracle as O
    B = 3
    mask = _geometry_masks(B, H, W, kind, seed=H + W)
    def down(n, k):                                      # ResNet /8,/16,/32 then the stride-2 3x3 conv
        for _ in range(k):
            n = (n - 1) // 2 + 1
        return n
    level_hw = [(down(H, 3), down(W, 3)), (down(H, 4), down(W, 4)), (down(H, 5), down(W, 5)), (down(H, 6), down(W, 6))]
    le = _rand((4, 256), 7)
    g = ops.geometry(mask.cuda(), level_hw, le.cuda(), 20, 20, dtype)
    masks = [O.interpolate_mask(mask, hw) for hw in level_hw]
    mask_flat = torch.cat([m.flatten(1) for m in masks], 1)
    assert torch.equal(g["mask_flat"].cpu(), mask_flat)
    pos = torch.cat([O.position_embedding_sine_hw(m).flatten(2).transpose(1, 2) + le[l].view(1, 1, -1) for l, m in enumerate(masks)], 1)
    tol = 2e-5 if dtype == torch.float32 else 0.04
    assert (g["pos"].float().cpu() - pos).abs().max() < tol
    vr = torch.stack([O.get_valid_ratio(m) for m in masks], 1)
    assert (g["valid_ratios"].cpu() - vr).abs().max() < 1e-7
    shapes = torch.as_tensor(level_hw, dtype=torch.long)
    ref = O.encoder_reference_points(shapes, vr)
    assert (g["enc_ref"].cpu() - ref).abs().max() < 1e-6
    mem = torch.ones((B, mask_flat.shape[1], 4))
    om, prop = O.gen_encoder_output_proposals(mem, mask_flat, shapes)
    got = g["proposals"].cpu()
    assert torch.equal(torch.isinf(got), torch.isinf(prop))
    fin = ~torch.isinf(prop)
    assert (got[fin] - prop[fin]).abs().max() < 2e-6
    assert torch.equal(g["keep"].cpu(), om[..., 0] != 0)


@pytest.mark.parametrize("B,H,W", [(2, 128, 2048), (1, 37, 101), (3, 32, 256)])
def test_stem_conv7x7_f32_vs_reference(B, H, W):
    """The exact-fp32 stem kernel (parity engine) against F.conv2d on the CPU: 7x7 / stride 2 / pad 3, 3 -> 64, NHWC output."""
    from dtlr_amd import ops
    x = _rand((B, 3, H, W), 11)
    w = _rand((64, 3, 7, 7), 12, 0.1)
    want = F.conv2d(x, w, None, stride=2, padding=3).permute(0, 2, 3, 1)
    got = ops.stem_conv7x7_f32(x.cuda(), ops.stem_pack_weights_f32(w).cuda()).cpu()
    assert got.shape == want.shape
    assert (got - want).abs().max() < 2e-5


@pytest.mark.parametrize("M,N,K,dt", [(5440 * 2, 192, 768, "h16"), (1000, 7356, 768, "h16"), (777, 166, 256, torch.float32),
                                      (130, 23, 256, torch.float32)])
def test_linear_rowmax_vs_reference(M, N, K, dt, half):
    """The GEMM's row-max epilogue == (x @ w.T + b).max(-1) computed from the full product of the same kernel (so the comparison is
    exact up to nothing: max is order-independent) and close to the fp32 CPU product; -inf padded bias rows never win."""
    from dtlr_amd import ops
    dt = half if dt == "h16" else dt
    x = _rand((M, K), 1).to(dt)
    w = _rand((N, K), 2, 0.1).to(dt)
    b = _rand((N,), 3)
    if N == 192:
        b[166:] = float("-inf")
    full = ops.linear(x.cuda(), w.cuda(), b.cuda(), out_dtype=torch.float32)
    got = ops.linear_rowmax(x.cuda(), w.cuda(), b.cuda())
    assert torch.equal(got, full.max(-1)[0])
    want = (x.float() @ w.float().t() + b).max(-1)[0]
    assert (got.cpu() - want).abs().max() < (1e-4 if dt == torch.float32 else 2e-3)
    # A = the leading K columns of wider rows (dtlr_gemm_nt_rowmax_lda): identical to the contiguous copy
    wide = torch.cat([x, _rand((M, 2 * K), 9).to(dt)], -1).cuda()
    assert torch.equal(ops.linear_rowmax(wide[:, :K], w.cuda(), b.cuda()), got)


def test_decode_blank_flags_non_finite_lines():
    """Round 5 (advisor, round 4): the fp16 / split engines turn an activation beyond 65504 into inf and then NaN; their range check runs on
    the first forward only.  The blank decoder now flags a line whose logits are not finite ON THE DEVICE (length -1, no host sync), the
    other lines of the batch decode as before, and the host-side consumer raises instead of returning garbage labels."""
    from dtlr_amd import _lib, evaluation as E, ops
    g = torch.Generator().manual_seed(5)
    logits = (torch.randn((4, 900, 166), generator=g) * 2 - 4).cuda()
    boxes = torch.rand((4, 900, 4), generator=g).cuda()
    lab0, len0 = ops.decode_blank(logits, boxes, 0.03 / 166)
    assert (len0 >= 0).all()
    bad = logits.clone()
    bad[1, 17, 3] = float("nan")
    bad[3, 899, 165] = float("inf") - float("inf")
    lab1, len1 = ops.decode_blank(bad, boxes, 0.03 / 166)
    assert len1.tolist()[1] == -1 and len1.tolist()[3] == -1
    assert len1[0] == len0[0] and len1[2] == len0[2] and torch.equal(lab1[0], lab0[0]) and torch.equal(lab1[2], lab0[2])
    with pytest.raises(_lib.DTLRError):
        E.records_to_lists(lab1, len1)
    assert E.records_to_lists(lab0, len0)[0] == lab0[0, : int(len0[0])].tolist()
    # round 6: +inf / -inf logits leave the sigmoid sum FINITE (sigmoid(+inf) = 1, sigmoid(-inf) = 0) -- they are flagged too,
    # whichever class lane and 64-class chunk they sit in
    for val, (bb, qq, cc) in ((float("inf"), (0, 0, 0)), (float("-inf"), (2, 451, 70)), (float("inf"), (2, 899, 165)), (float("-inf"), (1, 3, 64))):
        inf = logits.clone()
        inf[bb, qq, cc] = val
        lab2, len2 = ops.decode_blank(inf, boxes, 0.03 / 166)
        got = len2.tolist()
        assert got[bb] == -1 and all(got[i] == len0.tolist()[i] for i in range(4) if i != bb), (val, bb, qq, cc, got)


@pytest.mark.parametrize("M,N", [(300, 166), (1000, 7356), (257, 2048), (8704, 7356), (31, 36), (512, 4), (50000, 166), (49300, 36)])
def test_head_ts_vs_fp64_and_tiled_gemm(M, N, half):
    """dtlr_head_ts (token-stationary class head, round 5) in all four forms -- row maximum / fp32 logits x three products on a
    [hi | lo | hi] image / two products on a 16-bit state -- against an fp64 evaluation of the SAME 16-bit operands (exact products, so
    the only difference is fp32 accumulation order) and against the tiled GEMM path it replaces for large charsets; ragged M (clamped
    tail rows), N not a multiple of 32 (padded classes never win and are never stored), one chunk, many chunks; M < 49152 runs the
    4-wave form (128 tokens per workgroup), M >= 49152 the 8-wave form (256)."""
    from dtlr_amd import ops
    xf = _rand((M, 256), 1, 1.2) + 0.1                              # an fp32 row (output_memory after enc_output_norm)
    w, b = _rand((N, 256), 2) / 16.0, _rand((N,), 3) * 0.5 - 2.0
    hi = xf.to(half)
    lo = (xf - hi.float()).to(half)
    x3 = torch.cat([hi, lo, hi], -1).contiguous().cuda()            # proj_ln_split's image
    whi = w.to(half)
    wlo = (w - whi.float()).to(half)
    img, bias = ops.head_ts_pack(w.cuda(), b.cuda(), half)
    d = lambda t: t.double()                                        # noqa: E731
    want3 = d(hi) @ d(whi).t() + d(lo) @ d(whi).t() + d(hi) @ d(wlo).t() + d(b)
    want2 = d(hi) @ d(whi).t() + d(hi) @ d(wlo).t() + d(b)
    tol = 2e-5 * max(1.0, want3.abs().max().item())
    r3 = ops.head_ts(x3, img, bias, N, "rowmax", 0, 256).cpu()
    assert r3.shape == (M,) and (r3 - want3.max(-1)[0].float()).abs().max() < tol
    r2 = ops.head_ts(hi.cuda().contiguous(), img, bias, N, "rowmax").cpu()
    assert (r2 - want2.max(-1)[0].float()).abs().max() < tol
    # the operand offsets: B taken from the THIRD block of the image (= hi again): A Whi + A Whi + A Wlo
    r3b = ops.head_ts(x3, img, bias, N, "rowmax", 0, 512).cpu()
    assert (r3b - (want2 + d(hi) @ d(whi).t()).max(-1)[0].float()).abs().max() < 2 * tol
    if N % 4 == 0:
        y3 = ops.head_ts(x3, img, bias, N, "logits", 0, 256).cpu()
        assert y3.shape == (M, N) and (y3 - want3.float()).abs().max() < tol
        assert torch.equal(y3.max(-1)[0], r3)                       # the two modes run the same arithmetic
        y2 = ops.head_ts(hi.cuda().contiguous(), img, bias, N, "logits").cpu()
        assert (y2 - want2.float()).abs().max() < tol
    # the tiled GEMM forms the engine used for every charset before round 5
    w3, b3 = ops.split_head_weight(w.cuda(), b.cuda(), dtype=half)
    old = ops.linear_rowmax(x3, w3, b3).cpu()
    assert (old - r3).abs().max() < tol
    w2 = torch.cat([whi, wlo], 1).contiguous().cuda()
    old2 = ops.linear(torch.cat([hi, hi], -1).cuda(), w2, b.cuda(), out_dtype=torch.float32).cpu()
    assert (old2.max(-1)[0] - r2).abs().max() < tol
    with pytest.raises(Exception):
        ops.head_ts(x3, img[:-8], bias, N, "rowmax", 0, 256)        # not the image of this N


@pytest.mark.parametrize("dt", ["h16", torch.float32])
def test_two_stage_gather(dt, half):
    from dtlr_amd import ops
    dt = half if dt == "h16" else dt
    B, S, k = 3, 700, 90
    om = _rand((B, S, 768 if dt == half else 256), 5).to(dt)
    prop = _rand((B, S, 4), 6)
    prop[0, 5] = float("inf")
    g = np.random.Generator(np.random.PCG64(2))
    idx = torch.from_numpy(np.stack([g.permutation(S)[:k] for _ in range(B)])).long()
    idx[0, 0] = 5
    raw, sx, ps, ib = ops.two_stage_gather(om.cuda(), prop.cuda(), idx.cuda())
    want = torch.gather(om, 1, idx[..., None].expand(-1, -1, om.shape[-1]))
    assert torch.equal(raw.cpu(), want)
    wp = torch.gather(prop, 1, idx[..., None].expand(-1, -1, 4))
    assert torch.equal(ps.cpu(), wp)
    assert (ib.cpu() - wp.sigmoid()).abs().max() < 1e-6
    if dt == half:
        assert torch.equal(sx.cpu(), (want[..., :256].float() + want[..., 256:512].float()).to(half))
    else:
        assert sx is None


@pytest.mark.parametrize("B,S,k", [(2, 20000, 900), (1, 70000, 900), (2, 16385, 30)])
def test_topk_rows_large_S_global_path(B, S, k):
    """Rows too long for the LDS copy (tall canvases) take the global-memory radix select: same order, ties -> lower index."""
    from dtlr_amd import ops
    x = _rand((B, S), 8)
    x[:, 100:140] = x[:, 99:100]                       # a block of exact ties
    x[0, 7] = float("inf")
    got = ops.topk_rows(x.cuda(), k).cpu()
    key = torch.argsort(-x, dim=1, stable=True)[:, :k]
    assert torch.equal(got, key)


def test_topk_flat_more_than_1024():
    from dtlr_amd import ops
    x = _rand((2, 300 * 166), 4)
    v, i = ops.topk_flat(x.cuda(), 3000, apply_sigmoid=True)
    key = torch.argsort(-x, dim=1, stable=True)[:, :3000]
    assert torch.equal(i.cpu(), key)
    assert (v.cpu() - torch.gather(x, 1, key).sigmoid()).abs().max() < 1e-6


def test_no_library_fallbacks():
    """Shapes the kernels do not take raise instead of dropping to hipBLASLt / MIOpen; nothing is library-backed."""
    from dtlr_amd import _lib, ops
    assert ops.LIBRARY_BACKED == set()
    with pytest.raises(_lib.DTLRError):
        ops.linear(_rand((4, 100), 1).cuda(), _rand((8, 100), 2).cuda())
    with pytest.raises(_lib.DTLRError):
        ops.conv2d_nhwc(_rand((1, 8, 8, 3), 1).cuda(), _rand((4, 3, 3, 3), 2).cuda(), None, 1, 1)


def test_tall_canvas_falls_back_to_gather_msda_and_global_topk(half):
    """A 448x1344 line (eval transform: short side 800 / max 1333 for aspect ratio < 5): the LDS-window encoder kernel's plan does
    not fit (fp32) and S = 9408+... tokens; the engine must fall back to the gather kernel and still match the oracle."""
    from dtlr_amd import ops
    from dtlr_amd.config import DTLRConfig
    from dtlr_amd.dino import DINO
    from dtlr_amd import synth, weights
    from oracle import dtlr_oracle as O
    cfg = DTLRConfig.tiny()
    sd = weights.synthetic_state_dict(cfg, 1)
    imgs = synth.noise_lines(1, 448, 1344, seed=3) + synth.noise_lines(1, 300, 1100, seed=4)
    level_hw = [(56, 168), (28, 84), (14, 42), (7, 21)]
    assert not ops.msda_encoder_fits(level_hw, torch.float32)
    m = DINO(cfg)
    m.load_state_dict(sd)
    m = m.eval().to("cuda:0")
    out = m([i.cuda() for i in imgs], return_debug=True)
    assert out["_debug"]["geometry"]["lds_msda_fits"] is False
    ref = O.dino_forward(sd, cfg, imgs, forced_topk=out["_debug"]["topk_idx"].cpu())
    assert (out["pred_logits"].cpu() - ref["pred_logits"]).abs().max() < 1e-3
    assert (out["pred_boxes"].cpu() - ref["pred_boxes"]).abs().max() < 1e-4


@pytest.mark.parametrize("dt", ["h16", torch.float32])
def test_linear_row_broadcast_a2(dt, half):
    """(x + a2[m % S]) @ w.T + b with a2 = one [S, K] matrix shared by every image of the batch == the full-size a2 path."""
    from dtlr_amd import ops
    dt = half if dt == "h16" else dt
    B, S, K, N = 3, 333, 256, 384
    x, a2 = _rand((B, S, K), 1).to(dt), _rand((S, K), 2).to(dt)
    w, b = _rand((N, K), 3, 0.1).to(dt), _rand((N,), 4)
    got = ops.linear(x.cuda(), w.cuda(), b.cuda(), a2=a2.cuda())
    full = ops.linear(x.cuda(), w.cuda(), b.cuda(), a2=a2[None].expand(B, -1, -1).contiguous().cuda())
    assert torch.equal(got, full)
    want = ((x.float() + a2.float()[None]).to(dt).float() @ w.float().t() + b)
    assert (got.float().cpu() - want).abs().max() < (2e-4 if dt == torch.float32 else 0.05)


@pytest.mark.parametrize("M,N", [(174080, 256), (5440 * 3, 384), (1000, 384), (63, 256), (64 * 257 + 5, 256)])
def test_gemm_k256_weight_resident_vs_tiled_kernel(M, N, half):
    """dtlr_gemm_k256 (weights in registers, tokens streamed through an LDS ring by DMA) == the tiled GEMM on the same operands:
    same MFMA, same k order -> the bf16 results are identical; ragged M, one tile, many tiles per workgroup."""
    from dtlr_amd import ops
    x = _rand((M, 256), 1).to(half).cuda()
    w = _rand((N, 256), 2, 0.1).to(half).cuda()
    b = _rand((N,), 3).cuda()
    want = ops.linear(x, w, b)
    got = ops.gemm_k256(x, ops.k256_pack(w), N, b)
    assert torch.equal(got, want)
    # host packer == the tensor-op packer
    import numpy as np
    from dtlr_amd import _lib
    src = np.ascontiguousarray(w.cpu().view(torch.int16).numpy()).view(np.uint16)
    outp = np.empty(N * 256, dtype=np.uint16)
    assert _lib.lib().dtlr_k256_pack_weights(src.ctypes.data, outp.ctypes.data, N) == 0
    assert np.array_equal(outp, ops.k256_pack(w).cpu().view(torch.int16).numpy().view(np.uint16))


@pytest.mark.parametrize("B,S,N", [(3, 700, 384), (5, 640, 384), (32, 5440, 384), (7, 128, 256)])
def test_gemm_k256_epilogues(B, S, N, half):
    """Row-broadcast residual (m % rows), padding-row zeroing and strided output (column slice of a wider matrix).  S % 64 == 0
    takes the position-major tile order (a workgroup walks one residual tile across the images); S = 700 the row order."""
    from dtlr_amd import ops
    x = _rand((B, S, 256), 1).to(half).cuda()
    w = _rand((N, 256), 2, 0.1).to(half).cuda()
    b = _rand((N,), 3).cuda()
    res = _rand((S, N), 4).to(half).cuda()
    mask = (torch.rand((B, S)) < 0.2).cuda()
    wp = ops.k256_pack(w)
    got = ops.gemm_k256(x, wp, N, None, resid=res)
    want = (x.float() @ w.float().t() + res.float()[None]).to(half)
    assert (got.float() - want.float()).abs().max() <= 0.07          # one bf16 ulp at |y| < 16
    got = ops.gemm_k256(x, wp, N, b, resid=res, row_mask=mask)       # residual and padding rows together
    want = (x.float() @ w.float().t() + b + res.float()[None]).to(half)
    want[mask] = 0
    assert (got.float() - want.float()).abs().max() <= 0.07
    assert (got[mask] == 0).all()
    wide = torch.full((B, S, 1536), 7.0, dtype=half, device="cuda")
    ops.gemm_k256(x, wp, N, b, row_mask=mask, out=wide[..., 384:384 + N])
    full = ops.linear(x, w, b, row_mask=mask)
    assert torch.equal(wide[..., 384:384 + N], full)
    assert (wide[..., :384] == 7.0).all() and (wide[..., 384 + N:] == 7.0).all()
    assert (wide[..., 384:384 + N][mask] == 0).all()


@pytest.mark.parametrize("B,H,W", [(2, 32, 512), (1, 33, 517)])
def test_tall_tile_kernel_n64_conv_and_linear(B, H, W, half):
    """The 256 x 64 tile variant (N <= 64, bf16, M >= 16384: ResNet layer1's 3x3 64->64 convolutions and 256->64 reductions) against
    fp64 CPU references: bias + ReLU, residual + ReLU, ragged last tile, zero-padding taps."""
    import torch.nn.functional as F
    from dtlr_amd import ops
    x = _rand((B, H, W, 64), 1).to(half)
    w = (_rand((64, 64, 3, 3), 2) / 24.0).to(half)
    b = _rand((64,), 3)
    res = _rand((B, H, W, 64), 4).to(half)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2).double(), w.float().double(), b.double(), stride=1, padding=1).permute(0, 2, 3, 1)
    w_ohwi = w.permute(0, 2, 3, 1).contiguous()
    for use_res in (False, True):
        want = (ref + res.double() if use_res else ref).clamp(min=0).float()
        got = ops.conv2d_nhwc(x.cuda(), w_ohwi.cuda(), b.cuda(), 1, 1, True, res.cuda() if use_res else None).float().cpu()
        assert (got - want).abs().max() < ulp(half, 8) * max(1.0, want.abs().max().item()) + 1e-4, use_res
    xl = _rand((B * H * W, 256), 5).to(half)
    wl = (_rand((64, 256), 6) / 16.0).to(half)
    want = (xl.float().double() @ wl.float().double().t() + b.double()).clamp(min=0).float()
    got = ops.linear(xl.cuda(), wl.cuda(), b.cuda(), relu=2).float().cpu()
    assert (got - want).abs().max() < ulp(half, 8) * max(1.0, want.abs().max().item()) + 1e-4


@pytest.mark.parametrize("M", [174080, 65536 + 37, 64, 100])
def test_proj_ln_k256_vs_reference(M, half):
    """The weight-resident output-projection + residual + LayerNorm kernel against an fp32 reference and against the round-1 kernel
    (same arithmetic up to the summation order of the statistics); host packer == tensor-op packer."""
    from dtlr_amd import _lib, ops
    a = _rand((M, 256), 1).to(half)
    r = _rand((M, 256), 2).to(half)
    w = (_rand((256, 256), 3) / 16.0).to(half)
    b, gw, gb = _rand((256,), 4) * 0.1, _rand((256,), 5) * 0.2 + 1.0, _rand((256,), 6) * 0.1
    want = F.layer_norm(r.float() + a.float() @ w.float().t() + b, (256,), gw, gb, 1e-5)
    got = ops.proj_ln_k256(a.cuda(), ops.proj_ln_k256_pack(w).cuda(), b.cuda(), r.cuda(), gw.cuda(), gb.cuda())
    assert (got.float().cpu() - want).abs().max() < 0.04
    old = ops.proj_ln(a.cuda(), ops.proj_pack_w(w).cuda(), b.cuda(), r.cuda(), gw.cuda(), gb.cuda())
    assert (got.float() - old.float()).abs().max() <= 0.0315                 # at most one bf16 ulp at |y| < 8
    assert (got == old).float().mean() > 0.999
    src = np.ascontiguousarray(w.view(torch.int16).numpy()).view(np.uint16)
    outp = np.empty(65536, dtype=np.uint16)
    assert _lib.lib().dtlr_proj_ln_k256_pack_weights(src.ctypes.data, outp.ctypes.data) == 0
    assert np.array_equal(outp, ops.proj_ln_k256_pack(w).view(torch.int16).numpy().view(np.uint16))


@pytest.mark.parametrize("M,d_ff", [(256, 128), (1000, 2048), (65536 + 77, 2048), (174080, 2048), (700, 64), (513, 96), (300, 160), (1, 1024)])
def test_ffn32_vs_reference_and_first_structures(M, d_ff, half):
    """The 32x32x16-MFMA fused FFN (dtlr_ffn32_bf16) against an fp32 reference that rounds the hidden activations to bf16 like the
    kernel does, and against the 16x16x32 kernels (same arithmetic up to fp32 summation order); host packer == tensor-op packer."""
    from dtlr_amd import _lib, ops
    x = _rand((M, 256), 1).to(half)
    w1 = (_rand((d_ff, 256), 2) / 16.0).to(half)
    w2 = (_rand((256, d_ff), 3) / 45.0).to(half)
    b1, b2 = _rand((d_ff,), 4) * 0.1, _rand((256,), 5) * 0.1
    gw, gb = _rand((256,), 6) * 0.2 + 1.0, _rand((256,), 7) * 0.1
    h = torch.relu(x.float() @ w1.float().t() + b1).to(half).float()
    want = F.layer_norm(x.float() + h @ w2.float().t() + b2, (256,), gw, gb, 1e-5)
    w1p, w2p = ops.ffn32_pack(w1.cuda(), w2.cuda())
    got = ops.ffn32(x.cuda(), w1p, b1.cuda(), w2p, b2.cuda(), gw.cuda(), gb.cuda())
    assert (got.float().cpu() - want).abs().max() < 0.05
    if d_ff >= 128:
        old = ops.ffn_fused(x.cuda(), w1.cuda(), b1.cuda(), ops.ffn_pack_w2(w2.cuda()), b2.cuda(), gw.cuda(), gb.cuda())
        assert (got.float() - old.float()).abs().max() <= 0.0315            # at most one bf16 ulp at |y| < 8 (fp16: 8x finer, same bound holds)
        assert (got == old).float().mean() > (0.995 if half == torch.bfloat16 else 0.98)     # fp16's finer grid: two summation orders round apart more often
    a1 = np.ascontiguousarray(w1.view(torch.int16).numpy()).view(np.uint16)
    a2 = np.ascontiguousarray(w2.view(torch.int16).numpy()).view(np.uint16)
    n = (d_ff // 32 + _lib.lib().dtlr_ffn32_pad_chunks()) * 8192
    o1, o2 = np.empty(n, dtype=np.uint16), np.empty(n, dtype=np.uint16)
    assert _lib.lib().dtlr_ffn32_pack_weights(a1.ctypes.data, a2.ctypes.data, o1.ctypes.data, o2.ctypes.data, d_ff) == 0
    assert np.array_equal(o1, w1p.cpu().view(torch.int16).numpy().view(np.uint16))
    assert np.array_equal(o2, w2p.cpu().view(torch.int16).numpy().view(np.uint16))


@pytest.mark.parametrize("M,d_ff", [(1, 2048), (100, 2048), (128, 2048), (129, 2048), (256, 128), (257, 64), (1000, 2048), (5440, 2048), (28800, 2048),
                                    (65536 + 77, 2048), (174080, 2048), (700, 96), (40000, 160), (66000, 1024)])
def test_ffn4_vs_reference_and_ffn32(M, d_ff, half):
    """The persistent fused FFN (dtlr_ffn4_bf16, round 6: two staggered 128-row tiles per workgroup over a cyclic weight stream) against an
    fp32 reference that rounds the hidden activations to the 16-bit format like the kernel does, and against the 32x32x16 kernel it replaces
    (same arithmetic up to the fp32 summation order over the hidden chunks); row counts that leave the last tile ragged, a last PAIR with one
    tile, fewer tiles than compute units, more than two pairs per workgroup; the call is repeated to show it reproduces itself."""
    from dtlr_amd import ops
    x = _rand((M, 256), 1).to(half)
    w1 = (_rand((d_ff, 256), 2) / 16.0).to(half)
    w2 = (_rand((256, d_ff), 3) / 45.0).to(half)
    b1, b2 = _rand((d_ff,), 4) * 0.1, _rand((256,), 5) * 0.1
    gw, gb = _rand((256,), 6) * 0.2 + 1.0, _rand((256,), 7) * 0.1
    h = torch.relu(x.float() @ w1.float().t() + b1).to(half).float()
    want = F.layer_norm(x.float() + h @ w2.float().t() + b2, (256,), gw, gb, 1e-5)
    w1p, w2p = ops.ffn32_pack(w1.cuda(), w2.cuda())
    xc = x.cuda()
    guard = torch.full((M + 64, 256), 7.0, dtype=half, device="cuda")          # rows past M must stay untouched
    got = ops.ffn4(xc, w1p, b1.cuda(), w2p, b2.cuda(), gw.cuda(), gb.cuda(), out=guard[:M])
    assert (got.float().cpu() - want).abs().max() < 0.05
    assert bool((guard[M:] == 7.0).all())
    again = ops.ffn4(xc, w1p, b1.cuda(), w2p, b2.cuda(), gw.cuda(), gb.cuda())
    assert torch.equal(got, again)
    old = ops.ffn32(xc, w1p, b1.cuda(), w2p, b2.cuda(), gw.cuda(), gb.cuda())
    assert (got.float() - old.float()).abs().max() <= 0.0315                # at most one bf16 ulp at |y| < 8
    assert (got == old).float().mean() > (0.995 if half == torch.bfloat16 else 0.98)


@pytest.mark.parametrize("M,N,K,res", [(524288, 256, 64, True), (131072, 512, 128, True), (32768, 1024, 256, True), (16384 + 37, 256, 64, False),
                                         (20000, 2048, 256, True), (16500, 256, 128, True), (17000, 512, 64, True),
                                         (524288, 64, 256, False), (131072 + 5, 128, 256, False), (20000, 64, 64, False), (16400, 192, 128, False)])
def test_gemm_kres_vs_tiled_kernel(M, N, K, res, half):
    """dtlr_gemm_kres (weights resident, A and residual tiles DMA'd through an LDS ring) == the tiled GEMM on the same operands: same MFMA,
    same k order, same fp32 epilogue order (bias, residual, ReLU) -> identical bf16 results; ragged M; host packer == tensor-op packer."""
    from dtlr_amd import _lib, ops
    x = _rand((M, K), 1).to(half).cuda()
    w = (_rand((N, K), 2) / (K ** 0.5)).to(half).cuda()
    b = _rand((N,), 3).cuda()
    r = _rand((M, N), 4).to(half).cuda() if res else None
    do_relu = res or N < 256                                         # narrow outputs: the bottleneck's first convolution (bias + ReLU)
    want = ops.linear(x, w, b, relu=(2 if do_relu else 0), residual=r)
    wp = ops.kres_pack(w)
    got = ops.gemm_kres(x, wp, N, b, r, relu=do_relu)
    assert torch.equal(got, want)
    src = np.ascontiguousarray(w.cpu().view(torch.int16).numpy()).view(np.uint16)
    outp = np.empty(max(N, 256) * K, dtype=np.uint16)
    assert _lib.lib().dtlr_gemm_kres_pack_weights(src.ctypes.data, outp.ctypes.data, N, K) == 0
    assert np.array_equal(outp, wp.cpu().view(torch.int16).numpy().view(np.uint16))


@pytest.mark.parametrize("B,S", [(32, 5440), (3, 640), (1, 64)])
def test_gemm_kres_bcast384_vs_k256_kernel(B, S, half):
    """The [offsets | logits] projection with the row-broadcast residual DMA'd through LDS (zero-padded 512-channel column, position-major
    tiles) against the fp32 reference and the streaming K = 256 kernel; host packer == tensor-op packer."""
    from dtlr_amd import _lib, ops
    x = _rand((B, S, 256), 1).to(half).cuda()
    w = _rand((384, 256), 2, 0.1).to(half).cuda()
    res = _rand((S, 384), 4).to(half).cuda()
    wp = ops.kres_pack_bcast384(w)
    got = ops.gemm_kres_bcast384(x, wp, res)
    want = (x.float() @ w.float().t() + res.float()[None]).to(half)
    assert (got.float() - want.float()).abs().max() <= 0.07          # one bf16 ulp at |y| < 16
    old = ops.gemm_k256(x, ops.k256_pack(w), 384, None, resid=res)
    assert (got.float() - old.float()).abs().max() <= 0.07 and (got == old).float().mean() > 0.99
    src = np.ascontiguousarray(w.cpu().view(torch.int16).numpy()).view(np.uint16)
    outp = np.empty(512 * 256, dtype=np.uint16)
    assert _lib.lib().dtlr_gemm_kres_pack_weights_bcast384(src.ctypes.data, outp.ctypes.data) == 0
    assert np.array_equal(outp, wp.cpu().view(torch.int16).numpy().view(np.uint16))


@pytest.mark.parametrize("offscale,lo,hi", [(0.0, 0.0, 0.0), (2.0, 0.0, 0.002), (40.0, 0.2, 1.0)])
def test_msda_far_sample_probe(offscale, lo, hi, half):
    """dtlr_msda_encoder_far_samples: no sampling point at the reference points themselves leaves the staged windows, small offsets stay
    inside the halo, offsets of tens of pixels mostly do not; bf16 and fp32 projection rows give the same count up to rounding."""
    from dtlr_amd import ops
    from oracle import dtlr_oracle as O
    level_hw = [(16, 256), (8, 128), (4, 64), (2, 32)]
    N, M, L, P = 2, 8, 4, 4
    S = sum(h * w for h, w in level_hw)
    s = torch.tensor(level_hw)
    ow = _rand((N, S, M * L * P * 3), 15)
    ow[..., : M * L * P * 2] *= offscale
    ref = O.encoder_reference_points(s, torch.ones((N, 4, 2))).contiguous()
    f32 = ops.msda_encoder_far_fraction(half, level_hw, ow.cuda(), ref.cuda())
    b16 = ops.msda_encoder_far_fraction(half, level_hw, ow.to(half).cuda(), ref.cuda())
    assert lo <= f32 <= hi, f32
    assert abs(f32 - b16) <= 0.01 + 0.05 * f32


def test_tall128_tile_kernel_linear_and_conv(half):
    """The 256 x 128 tile variant (plain GEMM, N a multiple of 128, K >= 512, M >= 32768) against fp64 CPU references: bias only and bias +
    residual + ReLU on a ragged M, two channel tiles (grid.y); the 3x3 convolution of the same size stays on the 128 x 128 kernel."""
    import torch.nn.functional as F
    from dtlr_amd import ops
    M, N, K = 32768 + 37, 256, 1024
    x = _rand((M, K), 1).to(half)
    w = (_rand((N, K), 2) / 32.0).to(half)
    b = _rand((N,), 3)
    r = _rand((M, N), 4).to(half)
    ref = x.double() @ w.double().t() + b.double()
    got = ops.linear(x.cuda(), w.cuda(), b.cuda()).float().cpu()
    assert (got - ref.float()).abs().max() < ulp(half, 8) * max(1.0, ref.abs().max().item()) + 1e-4
    want = (ref + r.double()).clamp(min=0).float()
    got = ops.linear(x.cuda(), w.cuda(), b.cuda(), relu=2, residual=r.cuda()).float().cpu()
    assert (got - want).abs().max() < ulp(half, 8) * max(1.0, want.abs().max().item()) + 1e-4
    xi = _rand((2, 64, 200, 128), 5).to(half)
    wc = (_rand((256, 128, 3, 3), 6) / 34.0).to(half)
    bc = _rand((256,), 7)
    want = F.conv2d(xi.float().permute(0, 3, 1, 2).double(), wc.double(), bc.double(), stride=1, padding=1).permute(0, 2, 3, 1).clamp(min=0).float()
    got = ops.conv2d_nhwc(xi.cuda(), wc.permute(0, 2, 3, 1).contiguous().cuda(), bc.cuda(), 1, 1, True, None).float().cpu()
    assert (got - want).abs().max() < ulp(half, 8) * max(1.0, want.abs().max().item()) + 1e-4


def test_gemm_kres_and_k256_random_shapes(half):
    """Seeded sweep over the shape space the engine can hand the streaming kernels: ragged row counts, every K / N class, residual and
    ReLU on and off, padding masks -- all bit-equal to the tiled kernel."""
    from dtlr_amd import ops
    rng = np.random.Generator(np.random.PCG64(11))
    for case in range(14):
        K = int(rng.choice([64, 128, 256]))
        N = int(rng.choice([64, 128, 192, 256, 512, 768, 1024]))
        M = int(rng.integers(16384, 41000))
        res = bool(rng.integers(0, 2)) and N >= 256
        relu = bool(rng.integers(0, 2))
        x = _rand((M, K), 100 + case).to(half).cuda()
        w = (_rand((N, K), 200 + case) / (K ** 0.5)).to(half).cuda()
        b = _rand((N,), 300 + case).cuda() if rng.integers(0, 4) else None
        r = _rand((M, N), 400 + case).to(half).cuda() if res else None
        want = ops.linear(x, w, b, relu=(2 if relu else 0), residual=r)
        got = ops.gemm_kres(x, ops.kres_pack(w), N, b, r, relu=relu)
        assert torch.equal(got, want), (case, M, N, K, res, relu)
    for case in range(6):
        N = int(rng.choice([256, 384]))
        M = int(rng.integers(1, 30000))
        x = _rand((M, 256), 500 + case).to(half).cuda()
        w = _rand((N, 256), 600 + case, 0.1).to(half).cuda()
        b = _rand((N,), 700 + case).cuda()
        mask = (torch.rand((M,)) < 0.3).cuda() if case % 2 else None
        want = ops.linear(x, w, b, row_mask=mask)
        got = ops.gemm_k256(x, ops.k256_pack(w), N, b, row_mask=mask)
        assert torch.equal(got, want), (case, M, N)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 37, 131, 64, 64), (2, 64, 200, 128, 128), (32, 8, 128, 256, 256), (3, 40, 150, 64, 128),
                                            (2, 50, 170, 256, 64), (1, 128, 130, 128, 256)])
def test_conv3x3_patch_kernel_vs_reference(B, H, W, Cin, Cout, half):
    """The LDS-resident-patch 3x3 convolution (dtlr_conv3x3_patch_bf16, reached through conv2d_nhwc) against an fp64 reference and the
    implicit-GEMM kernel's entry on a smaller crop: image borders (zero padding), tiles cut by the right / bottom edge, bias and ReLU."""
    import torch.nn.functional as F
    from dtlr_amd import _lib, ops
    assert _lib.lib().dtlr_conv3x3_patch_supported(Cin, Cout) == 1 and B * H * W >= 16384
    x = _rand((B, H, W, Cin), 1).to(half)
    w = (_rand((Cout, Cin, 3, 3), 2) / (3.0 * Cin ** 0.5)).to(half)
    b = _rand((Cout,), 3)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=1, padding=1).permute(0, 2, 3, 1)
    w_ohwi = w.permute(0, 2, 3, 1).contiguous().cuda()
    for relu in (True, False):
        want = (ref.clamp(min=0) if relu else ref).float()
        got = ops.conv2d_nhwc(x.cuda(), w_ohwi, b.cuda(), 1, 1, relu, None).float().cpu()
        assert (got - want).abs().max() < ulp(half, 8) * max(1.0, want.abs().max().item()) + 1e-4, relu
    got = ops.conv2d_nhwc(x.cuda(), w_ohwi, None, 1, 1, False, None).float().cpu()
    assert (got - (ref - b.double()).float()).abs().max() < ulp(half, 8) * max(1.0, ref.abs().max().item()) + 1e-4


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(4, 37, 131, 64, 64), (2, 64, 200, 128, 128), (3, 40, 150, 64, 128), (1, 128, 130, 128, 256),
                                            (2, 32, 512, 64, 64)])
def test_conv3x3_patch_f32s_kernel_vs_fp64_and_tiled_split_gemm(B, H, W, Cin, Cout):
    """Round 6: the split-fp32 form of the LDS-resident-patch 3x3 convolution (dtlr_conv3x3_patch_f32s, reached through conv2d_nhwc with a
    SplitWeight) against fp64 on the same fp32 operands -- image borders (zero padding), tiles cut by the right / bottom edge, bias, ReLU,
    both channel-block counts -- at the split engine's tolerance, and against the implicit-GEMM split kernel it replaces (a crop below
    the routing threshold runs that one)."""
    import torch.nn.functional as F
    from dtlr_amd import _lib, ops
    assert _lib.lib().dtlr_conv3x3_patch_f32s_supported(Cin, Cout) == 1 and B * H * W >= 16384
    x = _rand((B, H, W, Cin), 1, 1.5) + 0.1
    w = _rand((Cout, Cin, 3, 3), 2) / (3.0 * Cin ** 0.5)
    b = _rand((Cout,), 3)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=1, padding=1).permute(0, 2, 3, 1)
    wsp = ops.split_pack(w.permute(0, 2, 3, 1).contiguous().cuda())
    tol = 3e-5 * max(1.0, ref.abs().max().item())
    for relu in (True, False):
        want = (ref.clamp(min=0) if relu else ref).float()
        got = ops.conv2d_nhwc(x.cuda(), wsp, b.cuda(), 1, 1, relu, None).cpu()
        assert torch.isfinite(got).all()
        err = (got - want).abs().max().item()
        print(f"[conv3x3 f32s patch {B}x{H}x{W} {Cin}->{Cout} relu={relu}] max err {err:.2e} (tol {tol:.2e})")
        assert err < tol, (relu, err)
    got = ops.conv2d_nhwc(x.cuda(), wsp, None, 1, 1, False, None).cpu()
    assert (got - (ref - b.double()).float()).abs().max() < tol
    # the implicit-GEMM split kernel on a crop below the patch kernel's routing threshold (M < 16384): the same numbers to fp32 rounding
    hc = max(1, min(H, 16000 // (B * W)))
    xc = x[:, :hc].contiguous()
    small = ops.conv2d_nhwc(xc.cuda(), wsp, b.cuda(), 1, 1, True, None).cpu()
    refc = F.conv2d(xc.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=1, padding=1).permute(0, 2, 3, 1).clamp(min=0).float()
    assert (small - refc).abs().max() < tol
    big = ops.conv2d_nhwc(x.cuda(), wsp, b.cuda(), 1, 1, True, None).cpu()
    if hc >= 3:                                                     # rows 0 .. hc - 2 of the crop see the same 3x3 neighbourhoods as in the full image
        assert (big[:, :hc - 1] - small[:, :hc - 1]).abs().max() < tol


@pytest.mark.parametrize("B,nq", [(32, 900), (3, 900), (2, 100), (1, 37)])
def test_dec_query_stage_equals_the_unfused_path(B, nq, half):
    """dtlr_dec_query_stage (reference boxes per level, sine embedding, ref_point_head, q | k and v projections in one launch) ==
    the operators it replaces (decoder_query_prep, two GEMMs, the [q|k] GEMM with its A + A2 prologue, the v GEMM), bit for bit --
    same MFMA, k ascending, the same roundings at the same places -- including a ragged last workgroup; and close to an fp64
    restatement of deformable_transformer.py:684-692, 904-907."""
    from dtlr_amd import ops
    ref = torch.sigmoid(_rand((B, nq, 4), 1)).cuda()
    vr = (0.5 + 0.5 * torch.rand((B, 4, 2), generator=torch.Generator().manual_seed(2))).cuda()
    tgt = _rand((B, nq, 256), 3).to(half).cuda()
    w0, w1 = (_rand((256, 512), 4) / 22).to(half).cuda(), (_rand((256, 256), 5) / 16).to(half).cuda()
    wqk, wv = (_rand((512, 256), 6) / 16).to(half).cuda(), (_rand((256, 256), 7) / 16).to(half).cuda()
    b0, b1, bqk, bv = _rand((256,), 8).cuda() * 0.1, _rand((256,), 9).cuda() * 0.1, _rand((512,), 10).cuda() * 0.1, _rand((256,), 11).cuda() * 0.1
    ref_in, qpos, qk, v = ops.dec_query_stage(ref, vr, tgt, ops.dq_pack(w0), b0, ops.dq_pack(w1), b1, ops.dq_pack(wqk), bqk, ops.dq_pack(wv), bv)
    ref_in2, sine = ops.decoder_query_prep(ref, vr, half)
    qpos2 = ops.linear(ops.linear(sine, w0, b0, relu=True), w1, b1)
    qk2 = ops.linear(tgt, wqk, bqk, a2=qpos2)
    v2 = ops.linear(tgt, wv, bv)
    assert torch.equal(ref_in, ref_in2)
    assert torch.equal(qpos, qpos2) and torch.equal(qk, qk2) and torch.equal(v, v2)
    # fp64 restatement on the same 16-bit inputs (loose: three chained roundings)
    s64 = sine.double().cpu()
    q64 = torch.relu(s64 @ w0.double().cpu().t() + b0.double().cpu()) @ w1.double().cpu().t() + b1.double().cpu()
    k64 = (tgt.double().cpu() + q64) @ wqk.double().cpu().t() + bqk.double().cpu()
    assert (qpos.double().cpu() - q64).abs().max() < 0.05 and (qk.double().cpu() - k64).abs().max() < 0.1


@pytest.mark.parametrize("B,H,W", [(2, 128, 2048), (1, 37, 531), (3, 8, 16), (1, 128, 2560), (2, 83, 1328), (1, 130, 1000)])
def test_stem_conv_pool_fused_equals_the_two_kernels(B, H, W, half):
    """dtlr_stem_conv7x7_pool (conv1 + FrozenBN shift + ReLU + max-pool in one kernel) == dtlr_stem_conv7x7 followed by the pooling
    pass, BIT FOR BIT (same roundings at the same places), on sizes that exercise every edge: odd conv / pooled sizes, maps narrower
    than one strip, the last partial workgroup in both directions."""
    from dtlr_amd import ops
    x = _rand((B, 3, H, W), 1).cuda()
    w = _rand((64, 3, 7, 7), 2) / 12
    bias = (_rand((64,), 3) * 0.3).cuda()
    frag = ops.stem_pack_weights(w, half).cuda()
    want = ops.maxpool_nhwc(ops.stem_conv7x7(x, frag, half), bias=bias, relu=True)
    got = ops.stem_conv7x7_pool(x, frag, bias, half)
    assert got.shape == want.shape
    assert torch.equal(got, want), (got.float() - want.float()).abs().max().item()


@pytest.mark.parametrize("M", [524288, 16384 + 64 * 5, 1000, 64, 37])
def test_gemm_kres_chain_equals_the_separate_launches(M, half):
    """dtlr_gemm_kres_chain: (a) identity-shortcut tail + the next bottleneck's conv1 (N2 = 64 and 128) == dtlr_gemm_kres twice, BIT FOR
    BIT (same operands, same k order; the second GEMM reads the rounded tile from LDS instead of HBM); (b) the first bottleneck's form
    -- the 1x1 shortcut convolution as K columns 64..127 -- against an fp64 restatement (it is MORE exact than the separate launches:
    the shortcut is not rounded to 16 bits before the add), its second GEMM bit-equal to dtlr_gemm_kres on the stored tile; ragged
    and tiny M included."""
    from dtlr_amd import ops
    t = torch.relu(_rand((M, 64), 1)).to(half).cuda()
    x0 = torch.relu(_rand((M, 64), 2)).to(half).cuda()
    idt = torch.relu(_rand((M, 256), 3)).to(half).cuda()
    w3, wd = (_rand((256, 64), 4) / 8).to(half).cuda(), (_rand((256, 64), 5) / 8).to(half).cuda()
    b3, bd = (_rand((256,), 6) * 0.2).cuda(), (_rand((256,), 7) * 0.2).cuda()
    for n2 in (64, 128):
        w2, b2 = (_rand((n2, 256), 8) / 16).to(half).cuda(), (_rand((n2,), 9) * 0.2).cuda()
        y, c2 = ops.gemm_kres_chain(t, ops.kres_pack(w3), b3, residual=idt, relu=True, wp2=ops.kres_pack(w2), b2=b2, n2=n2)
        y_ref = ops.gemm_kres(t, ops.kres_pack(w3), 256, b3, idt, relu=True)
        assert torch.equal(y, y_ref), (n2, (y.float() - y_ref.float()).abs().max().item())
        c2_ref = ops.gemm_kres(y_ref, ops.kres_pack(w2), n2, b2, None, relu=True)
        assert torch.equal(c2, c2_ref), (n2, (c2.float() - c2_ref.float()).abs().max().item())
    # first-bottleneck form: [t | x0] . [W3 | Wd]^T + (b3 + bd)
    w2, b2 = (_rand((64, 256), 8) / 16).to(half).cuda(), (_rand((64,), 9) * 0.2).cuda()
    wcat = ops.kres_pack(torch.cat([w3, wd], 1).contiguous())
    y, c2 = ops.gemm_kres_chain(t, wcat, b3 + bd, x2=x0, relu=True, wp2=ops.kres_pack(w2), b2=b2, n2=64)
    y_only, none = ops.gemm_kres_chain(t, wcat, b3 + bd, x2=x0, relu=True)
    assert none is None and torch.equal(y, y_only)
    want = torch.relu(t.double().cpu() @ w3.double().cpu().t() + x0.double().cpu() @ wd.double().cpu().t() + (b3 + bd).double().cpu())
    err = (y.double().cpu() - want).abs().max().item()
    assert err <= want.abs().max().item() * ulp(half, 8) + 1e-6, err
    assert torch.equal(c2, ops.gemm_kres(y, ops.kres_pack(w2), 64, b2, None, relu=True))
    # and close to the separate launches (which round the shortcut map first)
    y_sep = ops.gemm_kres(t, ops.kres_pack(w3), 256, b3, ops.gemm_kres(x0, ops.kres_pack(wd), 256, bd, None, relu=False), relu=True)
    assert (y.float() - y_sep.float()).abs().max().item() <= want.abs().max().item() * ulp(half, 7)


@pytest.mark.parametrize("B,Hin,Win", [(32, 32, 512), (2, 21, 332), (1, 5, 7), (3, 2, 2)])
def test_gemm_kres_cat_s2_vs_reference(B, Hin, Win, half):
    """dtlr_gemm_kres_cat_s2 (layer2.0's tail with the stride-2 shortcut convolution as K columns 128..383) against an fp64 restatement
    of relu(bn3(conv3(t)) + downsample(x)) on the same 16-bit operands, and against the two separate launches (which round the shortcut
    map to 16 bits first); odd input sizes (the eval canvas gives a 21 x 332 map), maps smaller than a 64-row tile."""
    from dtlr_amd import ops
    Hout, Wout = (Hin - 1) // 2 + 1, (Win - 1) // 2 + 1
    t = torch.relu(_rand((B, Hout, Wout, 128), 1)).to(half).cuda()
    x = torch.relu(_rand((B, Hin, Win, 256), 2)).to(half).cuda()
    w3, wd = (_rand((512, 128), 3) / 11).to(half).cuda(), (_rand((512, 256), 4) / 16).to(half).cuda()
    b3, bd = (_rand((512,), 5) * 0.2).cuda(), (_rand((512,), 6) * 0.2).cuda()
    y = ops.gemm_kres_cat_s2(t, x, ops.kres_pack(torch.cat([w3, wd], 1).contiguous()), b3 + bd, relu=True)
    assert tuple(y.shape) == (B, Hout, Wout, 512)
    xs = x[:, ::2, ::2].double().cpu()
    assert tuple(xs.shape[1:3]) == (Hout, Wout)
    want = torch.relu(t.double().cpu() @ w3.double().cpu().t() + xs @ wd.double().cpu().t() + (b3 + bd).double().cpu())
    scale = want.abs().max().item()
    err = (y.double().cpu() - want).abs().max().item()
    assert err <= scale * ulp(half, 8) + 1e-6, (err, scale)
    idt = ops.conv2d_nhwc(x, wd.view(512, 1, 1, 256), bd, 2, 0, False, None)
    y_sep = ops.gemm_kres(t.view(-1, 128), ops.kres_pack(w3), 512, b3, idt.reshape(-1, 512), relu=True).view_as(y)
    assert (y.float() - y_sep.float()).abs().max().item() <= scale * ulp(half, 7)


@pytest.mark.parametrize("level_hw,offscale", [([(16, 256), (8, 128), (4, 64), (2, 32)], 2.0), ([(16, 256), (8, 128), (4, 64), (2, 32)], 40.0),
                                               ([(5, 83), (3, 42), (2, 21), (1, 11)], 3.0), ([(1, 7), (1, 4), (1, 2), (1, 1)], 1.0)])
def test_msda_encoder_lds_default_form_both_row_dtypes(level_hw, offscale, half):
    """The default 16-bit query phase since round 4 (the "third form": coordinate clamp + one unsigned compare per axis, v_rcp_f32 softmax
    normalisation, paired weight conversions broadcast through op_sel, division-free staging, MODE.FP16_OVFL saturation; 654 VALU
    instructions per lane-iteration against 747, tools/isa_mix.py; same-box A/B of the step 9.24 -> 9.12 ms) against the oracle, for
    both projection-row dtypes, including samples outside the map and outside the staged windows; values beyond fp16's range stay finite."""
    from dtlr_amd import ops
    from oracle import dtlr_oracle as O
    N, M, D, L, P = 2, 8, 32, 4, 4
    S = sum(h * w for h, w in level_hw)
    v, s, lsi, _, _ = msda_inputs(N, M, D, S, P, level_hw, seed=13)
    ow = _rand((N, S, M * L * P * 3), 15)
    ow[..., : M * L * P * 2] *= offscale
    vr = torch.tensor([[[1.0, 1.0]] * 4, [[0.75, 1.0]] * 4])
    ref = O.encoder_reference_points(s, vr).contiguous()
    vv = v.to(half)
    for owx in (ow, ow.to(half)):
        off = owx.float()[..., : M * L * P * 2].view(N, S, M, L, P, 2)
        aw = torch.softmax(owx.float()[..., M * L * P * 2:].view(N, S, M, L * P), -1).view(N, S, M, L, P)
        want = O.ms_deform_attn_core(vv.float(), s, O.msda_sampling_locations(ref, off, s, P), aw)
        got = ops.msda_encoder(vv.cuda(), level_hw, owx.cuda(), ref.cuda()).float().cpu()
        tol = want.abs().max() * (ulp(half, 8) + 2.0 ** -9 if half == torch.bfloat16 else 2.0 ** -8) + 16 * 2.0 ** -24
        assert torch.isfinite(got).all()
        assert (got - want).abs().max() <= tol, (owx.dtype, (got - want).abs().max().item(), tol.item())
    if half == torch.bfloat16:      # beyond fp16's range: the hardware saturation (MODE.FP16_OVFL) must keep every output finite
        big = (vv.float() * 1e6).to(half)
        assert torch.isfinite(ops.msda_encoder(big.cuda(), level_hw, ow.to(half).cuda(), ref.cuda()).float()).all()


@pytest.mark.parametrize("B,L,spread", [(2, 900, 1.5), (1, 37, 1.5), (3, 200, 4.0), (1, 1184, 1.5), (2, 33, 8.0)])
def test_mha_two_pass_default_vs_fp32_reference(B, L, spread, half):
    """The default 16-bit attention kernel since round 4 (row maxima in a first pass over the staged keys, exp2 with the final maximum in
    the second, row sums as a third output tile of the P V product) against plain fp32 softmax(QK^T / sqrt(d)) V, incl. a ragged last key
    block, a single query block, the largest L the LDS form takes, and a dominant late key."""
    import math
    from dtlr_amd import ops
    H, hd = 8, 32
    C = H * hd
    qk = _rand((B, L, 2 * C), 11) * spread
    qk[:, (3 * L) // 4, C:] *= 5.0
    qk = qk.to(half)
    v = _rand((B, L, C), 12).to(half)
    q = qk[..., :C].float().view(B, L, H, hd).transpose(1, 2)
    k = qk[..., C:].float().view(B, L, H, hd).transpose(1, 2)
    vv = v.float().view(B, L, H, hd).transpose(1, 2)
    want = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1) @ vv).transpose(1, 2).reshape(B, L, C)
    got = ops.mha(qk.cuda(), v.cuda(), H).float().cpu()
    tol = ulp(half, 6) * v.float().abs().max() + 1e-3
    assert torch.isfinite(got).all()
    assert (got - want).abs().max() <= tol, (got - want).abs().max()


@pytest.mark.parametrize("B,L,spread", [(2, 900, 1.5), (1, 37, 1.5), (3, 200, 4.0), (1, 1184, 1.5), (2, 33, 8.0), (1, 1500, 1.5), (2, 608, 1.0)])
def test_mha_split_f32_vs_fp64_reference(B, L, spread):
    """DTLR_F32S attention (fp32 q / k / v as fp16 hi + lo halves, three MFMAs per product, keys staged in chunks) against an fp64
    softmax(QK^T / sqrt(d)) V: fp32-grade (the exact-fp32 kernel's own bound), for one and several chunks (L = 608: exactly one chunk
    of 19 key blocks; 900: two; 1500: three, three query-block groups), ragged last key block, a dominant late key."""
    import math
    from dtlr_amd import ops
    H, hd = 8, 32
    C = H * hd
    qk = _rand((B, L, 2 * C), 11) * spread
    qk[:, (3 * L) // 4, C:] *= 5.0
    v = _rand((B, L, C), 12)
    q = qk[..., :C].double().view(B, L, H, hd).transpose(1, 2)
    k = qk[..., C:].double().view(B, L, H, hd).transpose(1, 2)
    vv = v.double().view(B, L, H, hd).transpose(1, 2)
    want = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1) @ vv).transpose(1, 2).reshape(B, L, C).float()
    got = ops.mha(qk.cuda(), v.cuda(), H, split=True).cpu()
    exact = ops.mha(qk.cuda(), v.cuda(), H).cpu()
    err, err_exact = (got - want).abs().max().item(), (exact - want).abs().max().item()
    print(f"[split mha B{B} L{L}] max err {err:.2e} (exact-fp32 kernel {err_exact:.2e})")
    assert torch.isfinite(got).all() and err < 2e-5 * max(1.0, want.abs().max().item())


# ---- split-fp16 ("f32s") operands: fp32 activations, three fp16 MFMAs per product (round 4) -------------------------------------
def _split_image_reference(w):
    """what dtlr_split_pack_weights must write: per 32-k slab of a row, 4 chunks of 8 fp16 hi then 4 chunks of 8 fp16 lo."""
    rows, K = w.shape[0], w.numel() // w.shape[0]
    wf = w.reshape(rows, K).float()
    hi = wf.half()
    lo = (wf - hi.float()).half()
    img = torch.cat([hi.view(rows, K // 32, 32), lo.view(rows, K // 32, 32)], -1)            # [rows, slabs, 64 halves]
    return img.contiguous().view(torch.int16)


def test_split_pack_image_and_subnormal_halves():
    """the packed image is byte-for-byte the hi | lo slab layout, including weights so small that lo is an fp16 subnormal"""
    from dtlr_amd import ops
    for shape, scale in (((7, 64), 1.0), ((256, 256), 0.05), ((5, 3, 3, 32), 1e-3), ((3, 2048), 30.0)):
        w = _rand(shape, 11, scale)
        got = ops.split_pack(w.cuda())
        assert isinstance(got, ops.SplitWeight) and got.shape == w.shape and got.dtype == torch.float32
        want = _split_image_reference(w)
        assert torch.equal(got.cpu().as_subclass(torch.Tensor).contiguous().view(torch.int16).view(want.shape), want), (shape, scale)
        assert isinstance(got[1:3], ops.SplitWeight)                       # row slices stay images


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_split_vs_fp64(M, N, K):
    """DTLR_F32S GEMM vs an fp64 reference on the SAME fp32 operands, every epilogue / prologue combination of the fp32 kernel.
    The bound is the fp32 engine's own (2e-5 of the output scale: fp32 accumulation order), far below a plain fp16 product
    (5e-4); a second pass with operands of magnitude 1e-3 .. 1e-2 puts every lo half in the fp16 SUBNORMAL range -- the
    product stays fp32-grade only if the matrix cores keep subnormal inputs (they do on gfx950)."""
    from dtlr_amd import ops
    for sx, sw in ((1.0, 1.0), (1e-2, 3e-2)):
        x, a2 = _rand((M, K), 1, sx), _rand((M, K), 2, sx)
        w, b = _rand((N, K), 3, sw) / np.sqrt(K), _rand((N,), 4, sx * sw)
        res = _rand((M, N), 5, sx * sw)
        mask = torch.from_numpy(np.random.Generator(np.random.PCG64(6)).random(M) < 0.3)
        wp = ops.split_pack(w.cuda())
        worst = 0.0
        for relu, use_b, use_res, use_a2, use_mask in ((0, False, False, False, False), (1, True, False, False, False),
                                                       (0, True, True, True, True), (2, True, True, False, False)):
            want = _gemm_ref(x, w, b if use_b else None, relu, res if use_res else None, a2 if use_a2 else None, mask if use_mask else None)
            got = ops.linear(x.cuda(), wp, b.cuda() if use_b else None, relu, res.cuda() if use_res else None,
                             a2.cuda() if use_a2 else None, mask.cuda() if use_mask else None).cpu()
            assert got.shape == want.shape and got.dtype == torch.float32
            scale = max(want.abs().max().item(), 1e-30)
            worst = max(worst, (got - want).abs().max().item() / scale)
            # small operands: a lo half in fp16's subnormal range keeps an ABSOLUTE quantum of 2^-24, so a weight of ~1e-3 carries ~19
            # significand bits instead of 22 (measured 1e-5 of the output scale; a flushed lo would give 2.4e-4)
            bound = 2e-5 if sx == 1.0 else 5e-5
            assert (got - want).abs().max() < bound * scale, (sx, relu, use_b, use_res, use_a2, use_mask, (got - want).abs().max().item() / scale)
        print(f"[split gemm M{M} N{N} K{K} scale {sx:g}] worst error / output scale {worst:.2e}")


def test_gemm_split_rowmax_a2bcast_and_large_values():
    from dtlr_amd import ops
    # row-max epilogue (the two-stage selection scores)
    x, w, b = _rand((777, 256), 1), _rand((166, 256), 2) / 16.0, _rand((166,), 3)
    want = (x.double() @ w.double().t() + b.double()).max(-1)[0].float()
    got = ops.linear_rowmax(x.cuda(), ops.split_pack(w.cuda()), b.cuda()).cpu()
    assert (got - want).abs().max() < 2e-5 * want.abs().max()
    # row-broadcast A2 (position embedding of an unpadded batch)
    x, pos = _rand((3, 640, 256), 4), _rand((640, 256), 5)
    w, b = _rand((384, 256), 6) / 16.0, _rand((384,), 7)
    want = ((x + pos).double() @ w.double().t() + b.double()).float()
    got = ops.linear(x.cuda(), ops.split_pack(w.cuda()), b.cuda(), a2=pos.cuda()).cpu()
    assert (got - want).abs().max() < 2e-5 * want.abs().max()
    # activations up to a few thousand (ResNet maps before a normalisation): hi stays finite, the product exact to fp32 grade
    x, w = _rand((300, 512), 8, 900.0), _rand((128, 512), 9) / 22.0
    want = (x.double() @ w.double().t()).float()
    got = ops.linear(x.cuda(), ops.split_pack(w.cuda())).cpu()
    assert torch.isfinite(got).all() and (got - want).abs().max() < 2e-5 * want.abs().max()


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad", [(2, 16, 40, 64, 64, 3, 1, 1), (3, 9, 33, 128, 128, 3, 2, 1),
                                                         (1, 4, 64, 2048, 256, 3, 2, 1), (2, 5, 7, 32, 48, 3, 1, 1),
                                                         (2, 6, 10, 64, 96, 1, 1, 0), (2, 9, 21, 256, 512, 1, 2, 0)])
def test_conv2d_nhwc_split_vs_fp64(B, H, W, Cin, Cout, k, stride, pad):
    """DTLR_F32S implicit-GEMM convolution (padding taps, stride 2, split-K at small grids) vs torch's fp64 conv2d."""
    from dtlr_amd import ops
    x = _rand((B, H, W, Cin), 1)
    w = _rand((Cout, Cin, k, k), 2) / np.sqrt(Cin * k * k)
    b = _rand((Cout,), 3)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    res = _rand(tuple(ref.shape), 4)
    wp = ops.split_pack(w.permute(0, 2, 3, 1).contiguous().cuda())
    for relu, use_res in ((False, False), (True, False), (True, True)):
        want = ref + res.double() if use_res else ref
        want = want.clamp(min=0) if relu else want
        got = ops.conv2d_nhwc(x.cuda(), wp, b.cuda(), stride, pad, relu, res.cuda() if use_res else None).cpu()
        assert got.shape == want.shape
        assert (got - want.float()).abs().max() < 3e-5 * max(1.0, want.abs().max().item()), (relu, use_res)


@pytest.mark.parametrize("kind", ["f32", "f32s", "h16"])
@pytest.mark.parametrize("B,S,N", [(3, 640, 384), (32, 5440, 384), (2, 77, 166), (1, 128, 256)])
def test_linear_row_broadcast_residual(kind, B, S, N, half):
    """dtlr_gemm_nt_resbcast: x W^T + resid[row % S] against fp64, for the exact-fp32, split-fp32 and 16-bit kernels (ragged M / N tiles,
    one image, the bench shape); and the identity it serves: (src + pos) W^T + b == src W^T + (pos W^T + b)."""
    from dtlr_amd import ops
    K = 256
    x, pos = _rand((B, S, K), 1), _rand((S, K), 2)
    w, b = _rand((N, K), 3) / 16.0, _rand((N,), 4)
    dt = half if kind == "h16" else torch.float32
    xd, posd, wd = x.to(dt).cuda(), pos.to(dt).cuda(), w.to(dt).cuda()
    wk = ops.split_pack(wd) if kind == "f32s" else wd
    res = ops.linear(posd, wk, b.cuda())                                       # [S, N]: pos W^T + b
    got = ops.linear_resbcast(xd, wk, res).float().cpu()
    want = ((xd.double().cpu() @ wd.double().cpu().t()) + res.double().cpu()[None]).float()
    tol = (2e-5 if kind != "h16" else ulp(half, 8)) * max(1.0, want.abs().max().item()) + (0 if kind != "h16" else 1e-4)
    assert got.shape == (B, S, N) and (got - want).abs().max() < tol, (got - want).abs().max().item()
    if kind != "h16":
        full = ((x + pos).double() @ w.double().t() + b.double()).float()
        assert (got - full).abs().max() < 3e-5 * max(1.0, full.abs().max().item())


@pytest.mark.parametrize("M,d_ff", [(128, 2048), (300, 2048), (1000, 512), (4097, 2048), (37, 64), (513, 96), (300, 160), (1, 1024), (174080, 2048), (28800, 2048),
                                    (37768, 2048), (33000, 512), (33000, 96), (65700, 2048)])
def test_ffn_split_vs_fp64(M, d_ff):
    """dtlr_ffn_split (fused FFN block of the split-fp32 engine) against an fp64 evaluation of LayerNorm(x + relu(x W1^T + b1) W2^T + b2)
    on the same fp32 operands: fp32-grade (two split GEMMs + an fp32 LayerNorm), for ragged M, a single row, odd chunk counts
    (d_ff = 96 / 160: one zero chunk is run), the encoder and decoder shapes of the bench; and equal to the unfused split path
    (two DTLR_F32S GEMMs + dtlr_layernorm) to fp32 rounding.  Round 5: M beyond whole rounds of 256 workgroups runs the tail tiles
    split over the hidden dimension (174080: 80 tiles x 3 parts; 37768: 40 ragged tiles x 4; 33000 / 512: 2 tiles x 2; 65700: 2 full rounds + 2
    tiles x 4; 33000 / 96: too few chunks, plain kernel): every row is compared with the unfused path."""
    from dtlr_amd import ops
    big = M > 20000
    x = _rand((M, 256), 1, 1.5) + 0.3
    w1, b1 = _rand((d_ff, 256), 2) / 16.0, _rand((d_ff,), 3) * 0.5
    w2, b2 = _rand((256, d_ff), 4) / np.sqrt(d_ff), _rand((256,), 5) * 0.3
    g, be = _rand((256,), 6) * 0.2 + 1.0, _rand((256,), 7) * 0.1
    wp = ops.ffn_split_pack(w1.cuda(), w2.cuda())
    xd = x.cuda()
    got = ops.ffn_split(xd, wp, b1.cuda(), b2.cuda(), g.cuda(), be.cuda()).cpu()
    assert got.shape == x.shape and torch.isfinite(got).all()
    rows = torch.arange(0, M, max(1, M // 4000)) if big else torch.arange(M)          # the fp64 reference on a sample of the rows
    xs = x[rows].double()
    h = torch.relu(xs @ w1.double().t() + b1.double())
    want = F.layer_norm(xs + h @ w2.double().t() + b2.double(), (256,), g.double(), be.double(), 1e-5).float()
    err = (got[rows] - want).abs().max().item()
    print(f"[ffn_split M{M} d_ff{d_ff}] max err {err:.2e}")
    assert err < 3e-5 * max(1.0, want.abs().max().item()), err
    hid = ops.linear(xd, ops.split_pack(w1.cuda()), b1.cuda(), relu=True)
    unf = ops.layernorm(ops.linear(hid, ops.split_pack(w2.cuda()), b2.cuda()), g.cuda(), be.cuda(), 1e-5, xd).cpu()
    assert (got - unf).abs().max() < 3e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("B,H,W", [(2, 128, 2048), (1, 37, 531), (3, 8, 16), (1, 128, 2560), (2, 83, 1328)])
def test_stem_conv7x7_split_vs_fp64(B, H, W):
    """The split-fp32 stem (image and weights as fp16 hi + lo halves, three MFMAs per product) against an fp64 conv2d on the same fp32
    operands: fp32-grade, for odd sizes, maps narrower than a strip, the eval canvas; and within fp32 rounding of the exact direct kernel."""
    from dtlr_amd import ops
    x = _rand((B, 3, H, W), 11)
    w = _rand((64, 3, 7, 7), 12, 0.1)
    want = F.conv2d(x.double(), w.double(), None, stride=2, padding=3).permute(0, 2, 3, 1).float()
    fh, fl = ops.stem_pack_weights_split(w)
    got = ops.stem_conv7x7_f32s(x.cuda(), fh.cuda(), fl.cuda()).cpu()
    exact = ops.stem_conv7x7_f32(x.cuda(), ops.stem_pack_weights_f32(w).cuda()).cpu()
    assert got.shape == want.shape
    err, err_exact = (got - want).abs().max().item(), (exact - want).abs().max().item()
    print(f"[split stem {B}x{H}x{W}] max err {err:.2e} (exact-fp32 kernel {err_exact:.2e})")
    assert err < 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("M", [64, 1, 37, 300, 4097, 16384 + 5, 174080])
@pytest.mark.parametrize("mode", ["value", "proj_ln", "masked_ln"])
def test_gemm_k256s_vs_fp64(M, mode):
    """dtlr_gemm_k256s (weight-resident streaming projection of the split-fp32 engine) against fp64 on the same fp32 operands, both modes
    (bias + masked rows; bias + residual + LayerNorm), for a single row, ragged tails, more tiles than workgroups, the bench's token count;
    and within fp32 rounding of the tiled split GEMM (+ dtlr_layernorm)."""
    from dtlr_amd import ops
    x = _rand((M, 256), 21, 1.5) + 0.2
    w, b = _rand((256, 256), 22) / 16.0, _rand((256,), 23) * 0.5
    wp = ops.k256s_pack(w.cuda())
    rows = torch.arange(0, M, max(1, M // 4000))
    if mode == "value":
        mask = (torch.arange(M) % 7 == 3) if M > 1 else torch.zeros(1, dtype=torch.bool)
        got = ops.gemm_k256s(x.cuda(), wp, b.cuda(), row_mask=mask.cuda()).cpu()
        want = (x[rows].double() @ w.double().t() + b.double()).masked_fill(mask[rows][:, None], 0.0).float()
        tiled = ops.linear(x.cuda(), ops.split_pack(w.cuda()), b.cuda(), row_mask=mask.cuda()).cpu()
        nomask = ops.gemm_k256s(x.cuda(), wp, None).cpu()
        assert (nomask[rows] - (x[rows].double() @ w.double().t()).float()).abs().max() < 2e-5 * max(1.0, want.abs().max().item())
    elif mode == "masked_ln":                              # LayerNorm(b + (flagged ? 0 : x W^T)): the two-stage front end
        mask = (torch.arange(M) % 5 == 2) if M > 1 else torch.ones(1, dtype=torch.bool)
        g, be = _rand((256,), 25) * 0.2 + 1.0, _rand((256,), 26) * 0.1
        got = ops.gemm_k256s(x.cuda(), wp, b.cuda(), row_mask=mask.cuda(), ln_w=g.cuda(), ln_b=be.cuda()).cpu()
        xm = x.masked_fill(mask[:, None], 0.0)
        want = F.layer_norm(xm[rows].double() @ w.double().t() + b.double(), (256,), g.double(), be.double(), 1e-5).float()
        tiled = ops.layernorm(ops.linear(xm.cuda(), ops.split_pack(w.cuda()), b.cuda()), g.cuda(), be.cuda(), 1e-5).cpu()
        nomask = ops.gemm_k256s(x.cuda(), wp, b.cuda(), ln_w=g.cuda(), ln_b=be.cuda()).cpu()
        wantn = F.layer_norm(x[rows].double() @ w.double().t() + b.double(), (256,), g.double(), be.double(), 1e-5).float()
        assert (nomask[rows] - wantn).abs().max() < 2e-5 * max(1.0, wantn.abs().max().item())
    else:
        r = _rand((M, 256), 24, 2.0)
        g, be = _rand((256,), 25) * 0.2 + 1.0, _rand((256,), 26) * 0.1
        got = ops.gemm_k256s(x.cuda(), wp, b.cuda(), residual=r.cuda(), ln_w=g.cuda(), ln_b=be.cuda()).cpu()
        want = F.layer_norm(r[rows].double() + x[rows].double() @ w.double().t() + b.double(), (256,), g.double(), be.double(), 1e-5).float()
        tiled = ops.layernorm(ops.linear(x.cuda(), ops.split_pack(w.cuda()), b.cuda()), g.cuda(), be.cuda(), 1e-5, r.cuda()).cpu()
    assert got.shape == x.shape and torch.isfinite(got).all()
    err = (got[rows] - want).abs().max().item()
    print(f"[k256s {mode} M{M}] max err {err:.2e}")
    assert err < 2e-5 * max(1.0, want.abs().max().item()), err
    assert (got - tiled).abs().max() < 2e-5 * max(1.0, want.abs().max().item())
