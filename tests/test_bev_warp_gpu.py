"""GPU parity of the BEV-augmentation warp against the oracle and the
torch (grid_sample) restatement of kornia.warp_affine; backward against torch autograd."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bda(B, rng, flip=True):
    m = torch.eye(4).repeat(B, 1, 1)
    for b in range(B):
        a = float(rng.uniform(-0.4, 0.4))
        s = float(rng.uniform(0.9, 1.1))
        r = torch.tensor([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]], dtype=torch.float32) * s
        if flip and b % 2:
            r[:, 1] = -r[:, 1]
        m[b, :2, :2] = r
    return m


@pytest.mark.parametrize("shape", [(2, 80, 128, 128), (3, 16, 40, 56), (1, 4, 7, 5), (2, 64, 64, 512)])
def test_forward_against_oracle_and_torch(mmt_lib, oracle_mod, shape):
    from mm_training_amd.models.bev_depth import BEVDepth
    from mm_training_amd.ops.bev_warp import bev_warp_affine
    B, C, H, W = shape
    rng = np.random.default_rng(B * 100 + C)
    x = torch.from_numpy(rng.standard_normal((B, H, W, C)).astype(np.float32)).cuda().permute(0, 3, 1, 2)   # channels-last view
    bda = _bda(B, rng).cuda()
    y = bev_warp_affine(x, bda)
    assert y.shape == x.shape and y.is_contiguous(memory_format=torch.channels_last)
    ref = oracle_mod.bev_warp_affine(x.permute(0, 2, 3, 1).cpu().numpy(), bda.cpu().numpy())
    assert np.abs(y.permute(0, 2, 3, 1).cpu().numpy() - ref).max() <= 1e-6      # same coordinates (double -> fp32 once), same fp32 sum order
    t = BEVDepth.bev_augment_image_torch(None, x, bda)
    # grid_sample goes through normalised [-1, 1] coordinates and an LU inverse: source coordinates agree to
    # ~1e-4 px, which on white-noise input (neighbour differences of several units) is ~1e-3 in value
    assert (y - t).abs().max().item() <= 3e-3 and (y - t).abs().mean().item() <= 1e-4
    # identity matrix: exact copy
    eye = torch.eye(4).repeat(B, 1, 1).cuda()
    assert torch.equal(bev_warp_affine(x, eye), x.contiguous(memory_format=torch.channels_last))
    # NCHW-contiguous input takes the same path after a layout change
    assert torch.equal(bev_warp_affine(x.contiguous(), bda), y)


def test_backward_against_torch_autograd(mmt_lib):
    from mm_training_amd.models.bev_depth import BEVDepth
    from mm_training_amd.ops.bev_warp import bev_warp_affine
    rng = np.random.default_rng(3)
    B, C, H, W = 2, 80, 128, 128
    x0 = torch.from_numpy(rng.standard_normal((B, H, W, C)).astype(np.float32)).cuda().permute(0, 3, 1, 2)
    bda = _bda(B, rng).cuda()
    go = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).cuda()
    xa = x0.clone().requires_grad_(True)
    bev_warp_affine(xa, bda).backward(go)
    xb = x0.clone().requires_grad_(True)
    BEVDepth.bev_augment_image_torch(None, xb, bda).backward(go)
    assert (xa.grad - xb.grad).abs().max().item() <= 1e-2 and (xa.grad - xb.grad).abs().mean().item() <= 5e-4
    # adjoint identity <warp(x), g> == <x, warp^T(g)>
    lhs = (bev_warp_affine(x0, bda).double() * go.double()).sum().item()
    rhs = (x0.double() * xa.grad.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs)) + 1e-2


@pytest.mark.parametrize("kind", ["aug", "identity", "rot90", "zoom2", "shrink", "shear_shift"])
def test_backward_is_the_exact_adjoint_of_the_forward(mmt_lib, kind):
    """The backward gathers, for every source cell, the output cells whose bilinear footprint contains it.
    Checked against the transpose of the forward's own matrix (forward applied to every one-hot map),
    accumulated in float64: any missed or doubled candidate cell shows up as a full weight."""
    from mm_training_amd.ops.bev_warp import bev_warp_affine
    rng = np.random.default_rng(11)
    B, C, H, W = 2, 8, 9, 13
    bda = _bda(B, rng)
    if kind == "identity":
        bda = torch.eye(4).repeat(B, 1, 1)
    elif kind == "rot90":
        bda[:, :2, :2] = torch.tensor([[0.0, -1.0], [1.0, 0.0]])
    elif kind == "zoom2":        # one source cell feeds ~16 output cells
        bda[:, :2, :2] *= 2.0
    elif kind == "shrink":       # most output cells sample outside the map
        bda[:, :2, :2] *= 0.45
    elif kind == "shear_shift":
        bda[:, :2, :2] = torch.tensor([[1.0, 0.7], [-0.2, 1.1]])
        bda[:, 0, 2], bda[:, 1, 2] = 2.3, -1.6
    bda = bda.cuda()
    # forward matrix per sample: column j = warp(one-hot map j), identical for every channel
    basis = torch.eye(H * W, device="cuda").reshape(H * W, 1, H, W)
    mats = []
    for b in range(B):
        cols = bev_warp_affine(basis.expand(H * W, 4, H, W).contiguous(), bda[b:b + 1].expand(H * W, 4, 4).contiguous())
        mats.append(cols[:, 0].reshape(H * W, H * W).double())       # [source cell j, output cell]
    go = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).cuda()
    x = torch.zeros(B, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bev_warp_affine(x, bda).backward(go)
    for b in range(B):
        ref = (mats[b] @ go[b].reshape(C, H * W).double().T).T.reshape(C, H, W)
        assert (x.grad[b].double() - ref).abs().max().item() <= 1e-5, kind
    # bit-reproducible (no atomics)
    x2 = torch.zeros_like(x).requires_grad_(True)
    bev_warp_affine(x2, bda).backward(go)
    assert torch.equal(x.grad, x2.grad)


def test_backward_assign_equals_accumulate_on_zeros(mmt_lib):
    """mmt_bev_warp_affine_backward_assign (ABI 15) stores what mmt_bev_warp_affine_backward adds to a zeroed buffer -- bit for bit, into
    a buffer that held NaNs, strided rows on both sides, a sample whose map is zoomed in (cells under more output cells than the
    kernel keeps in LDS take its long path)."""
    from mm_training_amd import _lib
    rng = np.random.default_rng(5)
    B, C, H, W, S_in, S_out = 3, 80, 33, 47, 160, 96
    bda = _bda(B, rng)
    bda[2, :2, :2] *= 2.2
    bda = bda.cuda().contiguous()
    go = torch.from_numpy(rng.standard_normal((B, H, W, S_in)).astype(np.float32)).cuda()
    acc = torch.zeros(B, H, W, S_out, device="cuda")
    asg = torch.full((B, H, W, S_out), float("nan"), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.call("mmt_bev_warp_affine_backward", B, H, W, C, bda.data_ptr(), go.data_ptr(), S_in, acc.data_ptr(), S_out, st)
    _lib.call("mmt_bev_warp_affine_backward_assign", B, H, W, C, bda.data_ptr(), go.data_ptr(), S_in, asg.data_ptr(), S_out, st)
    assert torch.equal(acc[..., :C], asg[..., :C]) and bool(torch.isnan(asg[..., C:]).all())
    _lib.call("mmt_bev_warp_affine_backward", B, H, W, C, bda.data_ptr(), go.data_ptr(), S_in, acc.data_ptr(), S_out, st)      # accumulates
    assert torch.allclose(acc[..., :C], 2 * asg[..., :C], rtol=1e-6, atol=1e-6)


def test_concat_buffer(mmt_lib):
    """The warped camera map lands in channels [0, C) of the camera|LiDAR buffer, the LiDAR map in
    [C, C+C2) (models/bev_depth.py:187-192); gradients flow to both."""
    from mm_training_amd.ops.bev_warp import bev_warp_affine, bev_warp_concat
    rng = np.random.default_rng(4)
    B, C, C2, H, W = 2, 80, 64, 128, 128
    x = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)).cuda().requires_grad_(True)
    other = torch.from_numpy(rng.standard_normal((B, C2, H, W)).astype(np.float32)).cuda().requires_grad_(True)
    bda = _bda(B, rng).cuda()
    fused = bev_warp_concat(x, bda, other)
    assert fused.shape == (B, C + C2, H, W) and fused.is_contiguous(memory_format=torch.channels_last)
    ref = torch.cat([bev_warp_affine(x.detach(), bda), other.detach()], 1)
    assert torch.equal(fused, ref)
    go = torch.from_numpy(rng.standard_normal((B, C + C2, H, W)).astype(np.float32)).cuda()
    fused.backward(go)
    x2 = x.detach().clone().requires_grad_(True)
    bev_warp_affine(x2, bda).backward(go[:, :C].contiguous())
    assert (x.grad - x2.grad).abs().max().item() <= 1e-5        # atomics: summation order differs run to run
    assert torch.equal(other.grad, go[:, C:])


def test_errors(mmt_lib):
    from mm_training_amd import _lib
    from mm_training_amd.ops.bev_warp import bev_warp_affine
    bda = torch.eye(4).repeat(1, 1, 1).cuda()
    with pytest.raises(RuntimeError, match="CUDA"):
        bev_warp_affine(torch.zeros(1, 8, 4, 4), bda)
    with pytest.raises(_lib.MmtError, match="C % 4"):
        bev_warp_affine(torch.zeros(1, 6, 4, 4).cuda(), bda)
