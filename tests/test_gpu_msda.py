"""Parity of the HIP MSDeformAttn forward (through the C ABI) against the oracle.  GPU only."""
import os

import numpy as np
import pytest
import torch

from tests.util import c_oracle_msda, msda_inputs

pytestmark = pytest.mark.gpu


def _dev(*ts):
    return [t.cuda() for t in ts]


def _run(v, s, lsi, loc, aw, step=64):
    from dtlr_amd import MultiScaleDeformableAttention as MSDA
    return MSDA.ms_deform_attn_forward(*_dev(v, s, lsi, loc, aw), step).cpu()


def test_reference_unit_test_shape_f32_f64(golden_dir, oracle_clib):
    """ops/test.py:31-60 -- fp64 allclose default, fp32 rtol=1e-2 atol=1e-3 -- against the golden
    outputs of the reference's ms_deform_attn_core_pytorch."""
    g = np.load(os.path.join(golden_dir, "g1_msda.npz"))
    v, s, loc, aw = (torch.from_numpy(g[k]) for k in ("a_value", "a_shapes", "a_loc", "a_aw"))
    lsi = torch.cat((s.new_zeros((1,)), s.prod(1).cumsum(0)[:-1]))
    o32 = _run(v, s, lsi, loc, aw, step=2)
    assert torch.allclose(o32, torch.from_numpy(g["a_out_f32"]), rtol=1e-2, atol=1e-3)
    assert (o32 - torch.from_numpy(g["a_out_f32"])).abs().max() < 1e-7
    o64 = _run(v.double(), s, lsi, loc.double(), aw.double(), step=2)
    assert torch.allclose(o64, torch.from_numpy(g["a_out_f64"]))


@pytest.mark.parametrize("D", [1, 2, 30, 32, 64, 71])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_channel_counts_vs_c_oracle(oracle_clib, D, dtype):
    """Channel counts of ops/test.py:85-86 (30, 32, 64, 71) + 1/2, generic L/P, odd sizes."""
    v, s, lsi, loc, aw = msda_inputs(2, 3, D, 19, 3, [(5, 7), (3, 4), (1, 2)], seed=100 + D, lo=-0.3, hi=1.3)
    v, loc, aw = v.to(dtype), loc.to(dtype), aw.to(dtype)
    want = c_oracle_msda(oracle_clib, v, s, lsi, loc, aw)
    got = _run(v, s, lsi, loc, aw)
    tol = 1e-12 if dtype == torch.float64 else 2e-6
    assert (got - want).abs().max() < tol


def test_hot_shape_golden_and_oracle(golden_dir, oracle_clib):
    g = np.load(os.path.join(golden_dir, "g1_msda.npz"))
    shapes = [tuple(x) for x in g["b_shapes"].tolist()]
    v, s, lsi, loc, aw = msda_inputs(2, 8, 32, 77, 4, shapes, seed=int(g["b_seed"]), lo=-0.5, hi=1.5)
    assert (_run(v, s, lsi, loc, aw) - torch.from_numpy(g["b_out_f32"])).abs().max() < 2e-6
    shapes = [tuple(x) for x in g["c_shapes"].tolist()]
    v, s, lsi, loc, aw = msda_inputs(1, 8, 32, 5440, 4, shapes, seed=int(g["c_seed"]), lo=-0.25, hi=1.25)
    got = _run(v, s, lsi, loc, aw)
    assert (got[0, ::97] - torch.from_numpy(g["c_out_rows"])).abs().max() < 2e-6
    want = c_oracle_msda(oracle_clib, v, s, lsi, loc, aw)
    assert (got - want).abs().max() < 2e-6


def test_decoder_shape_vs_oracle(oracle_clib):
    shapes = [(16, 256), (8, 128), (4, 64), (2, 32)]
    v, s, lsi, loc, aw = msda_inputs(2, 8, 32, 900, 4, shapes, seed=5, lo=-0.1, hi=1.1)
    assert (_run(v, s, lsi, loc, aw) - c_oracle_msda(oracle_clib, v, s, lsi, loc, aw)).abs().max() < 2e-6


def test_border_semantics(oracle_clib):
    """Exact-border locations: h_im == -1 and h_im == H are excluded (cuh:285-288); points on the
    first/last pixel centre, on the map edge, and far outside."""
    H, W = 4, 6
    pts = [(0.0, 0.0), (1.0, 1.0), (0.5 / W, 0.5 / H), ((W - 0.5) / W, (H - 0.5) / H), (-0.5 / W, 0.5), (0.5, -0.5 / H),
           ((W + 0.5) / W, 0.5), (0.5, (H + 0.5) / H), (-3.0, 0.5), (0.5, 9.0), (0.25, 0.75), (1.0 - 1e-7, 1e-7),
           (-0.4999 / W, 0.3), (0.3, -0.4999 / H), (1 + 0.4999 / W, 0.9), (0.9, 1 + 0.4999 / H)]
    loc = torch.tensor(pts, dtype=torch.float32).view(1, 1, 1, 1, len(pts), 2)
    r = np.random.Generator(np.random.PCG64(1))
    v = torch.from_numpy(r.standard_normal((1, H * W, 1, 8)).astype(np.float32))
    aw = torch.full((1, 1, 1, 1, len(pts)), 1.0 / len(pts))
    s = torch.tensor([[H, W]])
    lsi = torch.tensor([0])
    for p in range(len(pts)):
        one_loc, one_aw = loc[..., p:p + 1, :].contiguous(), torch.ones(1, 1, 1, 1, 1)
        want = c_oracle_msda(oracle_clib, v, s, lsi, one_loc, one_aw)
        got = _run(v, s, lsi, one_loc, one_aw)
        assert (got - want).abs().max() < 1e-6, pts[p]
    assert (_run(v, s, lsi, loc, aw) - c_oracle_msda(oracle_clib, v, s, lsi, loc, aw)).abs().max() < 1e-6


def test_bf16_value(oracle_clib):
    """bf16 storage, fp32 arithmetic: equals the oracle run on the bf16-rounded value up to the final
    bf16 rounding of the output."""
    shapes = [(16, 256), (8, 128), (4, 64), (2, 32)]
    v, s, lsi, loc, aw = msda_inputs(2, 8, 32, 333, 4, shapes, seed=8, lo=-0.1, hi=1.1)
    vb = v.bfloat16()
    want = c_oracle_msda(oracle_clib, vb.float(), s, lsi, loc, aw)
    got = _run(vb, s, lsi, loc, aw)
    assert got.dtype == torch.bfloat16
    assert (got.float() - want).abs().max() <= want.abs().max() * 2 ** -8 + 1e-6
    assert (got.float() - want.bfloat16().float()).abs().max() <= want.abs().max() * 2 ** -7


def test_error_behaviour_matches_reference_wrapper():
    """cuda/ms_deform_attn_cuda.cu:28-52 + ms_deform_attn.h:38."""
    from dtlr_amd import MultiScaleDeformableAttention as MSDA
    v, s, lsi, loc, aw = msda_inputs(3, 2, 4, 5, 2, [(3, 3), (2, 2)], seed=1)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(v, s, lsi, loc, aw, 64)
    dv, ds, dl, dloc, daw = _dev(v, s, lsi, loc, aw)
    with pytest.raises(RuntimeError, match="value tensor has to be contiguous"):
        MSDA.ms_deform_attn_forward(dv.transpose(2, 3).contiguous().transpose(2, 3), ds, dl, dloc, daw, 64)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        MSDA.ms_deform_attn_forward(dv, s, dl, dloc, daw, 64)
    with pytest.raises(RuntimeError, match="must divide"):
        MSDA.ms_deform_attn_forward(dv, ds, dl, dloc, daw, 2)       # batch 3, step 2
    with pytest.raises(NotImplementedError):
        MSDA.ms_deform_attn_backward(dv, ds, dl, dloc, daw, dv, 64)
    assert MSDA.ms_deform_attn_forward(dv, ds, dl, dloc, daw, 3).shape == (3, 5, 8)


def test_reference_function_api():
    """MSDeformAttnFunction.apply(value, shapes, lsi, loc, attn, im2col_step) (ms_deform_attn_func.py:21-29)."""
    from dtlr_amd.ms_deform_attn import MSDeformAttnFunction
    v, s, lsi, loc, aw = msda_inputs(2, 2, 4, 5, 2, [(3, 3), (2, 2)], seed=2)
    out = MSDeformAttnFunction.apply(*_dev(v, s, lsi, loc, aw), 64)
    assert out.shape == (2, 5, 8) and out.is_cuda


def test_full_size_properties():
    """BASELINE bs=32 encoder call (N=32, S=Lq=5440): size-independent properties.
    (1) linear in value; (2) a constant value map sampled strictly inside returns const * sum(attn);
    (3) zero attention -> exactly zero; (4) per-sample independence (batch slice == full batch)."""
    shapes = [(16, 256), (8, 128), (4, 64), (2, 32)]
    v, s, lsi, loc, aw = msda_inputs(32, 8, 32, 5440, 4, shapes, seed=77, lo=-0.2, hi=1.2)
    dv, ds, dl, dloc, daw = _dev(v, s, lsi, loc, aw)
    from dtlr_amd import MultiScaleDeformableAttention as MSDA
    f = lambda val, lo_=dloc, aw_=daw: MSDA.ms_deform_attn_forward(val, ds, dl, lo_, aw_, 64)
    o1 = f(dv)
    v2 = torch.roll(dv, 1, dims=1) * 0.5
    assert (f(dv + v2) - (o1 + f(v2))).abs().max() < 2e-5
    inside = dloc.clamp(0.26, 0.74)        # >= half a pixel away from every border of every level
    oc = f(torch.full_like(dv, 3.0), inside)
    assert (oc - 3.0 * daw.sum((-1, -2)).unsqueeze(-1).expand(-1, -1, -1, 32).reshape(32, 5440, 256)).abs().max() < 1e-4
    assert f(dv, dloc, torch.zeros_like(daw)).abs().max() == 0
    sub = MSDA.ms_deform_attn_forward(dv[5:7].contiguous(), ds, dl, dloc[5:7].contiguous(), daw[5:7].contiguous(), 64)
    assert torch.equal(sub, o1[5:7])
