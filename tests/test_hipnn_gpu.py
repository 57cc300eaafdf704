"""hipnn.conv2d (implicit-GEMM convolution on the f32 matrix-core kernels + scl_conv_pack_weights / scl_conv_wgrad_finish) against
torch.nn.functional.conv2d in float64 on the CPU: output, input gradient, weight and bias gradients, for the ResNet back-end's geometries
(model/wav2vec2_resnet_nll.py's 3x3 / 1x1 convolutions, strides 1 and 2, channel counts that need padding to the 4-float vectors).
Tolerance: 2e-5 of each tensor's scale (exact-fp32 products; the backward in the bf16-pair form where asked: 5e-5)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from scl_amd import lib
    lib.load()
    return torch.device("cuda:0")


GEOMS = [  # B, H, W, Ci, Co, k, stride, pad
    (3, 20, 16, 1, 16, 3, (1, 1), (1, 1)),        # first conv: one input channel, padded to 4
    (2, 17, 12, 16, 32, 3, (2, 2), (1, 1)),       # strided block entry
    (2, 9, 10, 32, 6, 1, (2, 2), (0, 0)),         # 1x1 shortcut, Co not a multiple of 4
    (2, 8, 8, 64, 64, 3, (1, 1), (1, 1)),
    (2, 9, 10, 32, 8, 1, (1, 1), (0, 0)),         # 1x1, stride 1: the flat whole-batch weight gradient with a single tap
    (5, 30, 34, 8, 16, 3, (1, 1), (1, 1)),        # enough positions for several split-K slabs of the flat weight gradient
]


@pytest.mark.parametrize("in_place", [False, True])
@pytest.mark.parametrize("x3_bwd", [False, True])
@pytest.mark.parametrize("geom", GEOMS)
def test_conv2d_matches_torch_fp64(dev, geom, x3_bwd, in_place):
    from scl_amd import hipnn
    B, H, W, Ci, Co, k, stride, pad = geom
    g = torch.Generator().manual_seed(B * 1000 + Ci * 10 + Co)
    x = torch.randn(B, H, W, Ci, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) * 0.2
    b = torch.randn(Co, generator=g)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.conv2d(xr.permute(0, 3, 1, 2), wr, br, stride=stride, padding=pad).permute(0, 2, 3, 1)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())

    xd = x.to(dev).requires_grad_(True)
    wd = torch.nn.Parameter(w.to(dev))
    bd = torch.nn.Parameter(b.to(dev))
    prior = torch.randn(w.shape, generator=g)
    if in_place:
        wd.grad = prior.to(dev).clone()          # an attached gradient buffer with earlier contents: the kernel must ADD into it
    hipnn.weights_changed()
    y = hipnn.conv2d(xd, wd, bd, stride, pad, x3_bwd=x3_bwd, grad_in_place=in_place)
    seen = []
    wd.register_hook(lambda g_: seen.append(1) if g_ is not None else None)           # records a weight gradient handed to autograd
    y.backward(dy.to(dev))
    assert bool(seen) == (not in_place)                    # in place: the finishing kernel added into wd.grad, the engine got nothing
    tol = 5e-5 if x3_bwd else 2e-5

    def close(got, ref, name, t):
        err = (got.double().cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
        assert err < t, (name, err)
    close(y.detach(), yr.detach(), "y", 2e-5)
    close(xd.grad, xr.grad, "dx", tol)
    close(bd.grad, br.grad, "db", tol)
    close(wd.grad, wr.grad + (prior.double() if in_place else 0.0), "dw", tol)


def test_packed_weights_follow_the_optimizer_epoch(dev):
    """The re-laid-out copies are cached per weights epoch: after weights_changed() (the model calls it after every optimizer step) the next
    forward must see the new values; without it the cache is allowed to serve the old ones."""
    from scl_amd import hipnn
    x = torch.randn(1, 6, 6, 4, device=dev)
    w = torch.nn.Parameter(torch.randn(8, 4, 3, 3, device=dev))
    hipnn.weights_changed()
    y0 = hipnn.conv2d(x, w, None, (1, 1), (1, 1)).detach().clone()
    with torch.no_grad():
        w.data.view(-1)[0] += 1.0                                   # a write torch's version counter does not see through .data
    hipnn.weights_changed()
    y1 = hipnn.conv2d(x, w, None, (1, 1), (1, 1)).detach()
    ref = F.conv2d(x.permute(0, 3, 1, 2).cpu().double(), w.detach().cpu().double(), None, padding=1).permute(0, 2, 3, 1)
    assert (y1.cpu().double() - ref).abs().max().item() < 1e-4
    assert (y1 - y0).abs().max().item() > 1e-3


def test_conv_entry_points_refuse_bad_arguments(dev):
    from scl_amd import lib, ops
    w = torch.zeros(8, 4, 3, 3, device=dev)
    f = torch.zeros(8, 9 * 4, device=dev)
    b = torch.zeros(4, 9 * 8, device=dev)
    with pytest.raises(lib.SclError):
        ops.conv_pack_weights(w, f, b, 8, 4, 3, 3, 2, 8)            # Cp < Ci
    with pytest.raises(lib.SclError):
        ops.conv_wgrad_finish(f, w, 0, 8, 4, 3, 3, 4)               # no slabs


@pytest.mark.parametrize("in_place", [False, True])
def test_batch_norm_matches_torch_and_accumulates_in_place(dev, in_place):
    """hipnn.batch_norm + ReLU in train mode against torch's BatchNorm2d on the CPU (fp64); with grad_in_place the finishing kernel of
    scl_bn_bwd adds dgamma / dbeta into the attached .grad tensors (earlier contents kept) and hands autograd nothing."""
    from scl_amd import hipnn
    g = torch.Generator().manual_seed(5)
    C = 32      # scl_bn_fwd takes channel counts that divide 256 (or 512)
    x = torch.randn(3, 7, 9, C, generator=g)
    ref_bn = torch.nn.BatchNorm2d(C).double()
    with torch.no_grad():
        ref_bn.weight.copy_(torch.randn(C, generator=g)); ref_bn.bias.copy_(torch.randn(C, generator=g))
    bn = torch.nn.BatchNorm2d(C).to(dev)
    with torch.no_grad():
        bn.weight.copy_(ref_bn.weight.float()); bn.bias.copy_(ref_bn.bias.float())
    xr = x.double().requires_grad_(True)
    yr = torch.relu(ref_bn(xr.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
    dy = torch.randn(yr.shape, generator=g)
    yr.backward(dy.double())
    prior_w, prior_b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    if in_place:
        bn.weight.grad, bn.bias.grad = prior_w.to(dev).clone(), prior_b.to(dev).clone()
    xd = x.to(dev).requires_grad_(True)
    y = hipnn.batch_norm(xd, bn, hipnn.ACT_RELU, grad_in_place=in_place)
    seen = []
    bn.weight.register_hook(lambda g_: seen.append(1) if g_ is not None else None)
    y.backward(dy.to(dev))
    assert bool(seen) == (not in_place)
    add_w, add_b = (prior_w.double(), prior_b.double()) if in_place else (0.0, 0.0)
    for got, ref, name in ((y.detach(), yr.detach(), "y"), (xd.grad, xr.grad, "dx"), (bn.weight.grad, ref_bn.weight.grad + add_w, "dgamma"),
                           (bn.bias.grad, ref_bn.bias.grad + add_b, "dbeta")):
        err = (got.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-5, (name, err)
    assert (bn.running_mean.cpu().double() - ref_bn.running_mean).abs().max().item() < 1e-5


def test_conv2d_residual_is_added_in_the_epilogue_and_passes_the_gradient_through(dev):
    """y = conv(x) + residual (resnet.py:60-70's `out += shortcut`) as the GEMM's f32 residual operand; d(residual) = dy."""
    from scl_amd import hipnn
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 10, 12, 8, generator=g).to(dev).requires_grad_(True)
    w = torch.nn.Parameter((torch.randn(16, 8, 3, 3, generator=g) * 0.2).to(dev))
    r = torch.randn(2, 10, 12, 16, generator=g).to(dev).requires_grad_(True)
    dy = torch.randn(2, 10, 12, 16, generator=g).to(dev)
    hipnn.weights_changed()
    y = hipnn.conv2d(x, w, None, (1, 1), (1, 1), residual=r)
    y.backward(dy)
    gx, gw = x.grad.clone(), w.grad.clone()
    x.grad = None; w.grad = None
    y2 = hipnn.conv2d(x, w, None, (1, 1), (1, 1)) + r.detach()
    y2.backward(dy)
    assert (y - y2).abs().max().item() < 1e-5 * y2.abs().max().item()
    assert torch.equal(r.grad, dy) and torch.equal(gx, x.grad) and torch.equal(gw, w.grad)


def test_padded_input_pool_survives_a_dying_older_graph(dev):
    """Lifetime of the pooled zero-bordered input map (round-5 advisory): step 0's graph is still alive (main.py's run_epoch keeps `out` /
    the loss) when step 1's forward A re-acquires the entry step 0's backward released; dropping the old graph must not free the entry under
    A, a further forward B of the same convolution must get its own map, and A's backward must still see A's input (weight gradient against
    float64).  A second backward through an already-released node is refused instead of reading someone else's map."""
    import gc
    from scl_amd import hipnn
    g = torch.Generator().manual_seed(11)
    w = torch.nn.Parameter((torch.randn(16, 8, 3, 3, generator=g) * 0.2).to(dev))
    hipnn.weights_changed()

    def fwd(seed):
        x = torch.randn(2, 12, 10, 8, generator=torch.Generator().manual_seed(seed))
        return x, hipnn.conv2d(x.to(dev).requires_grad_(True), w, None, (1, 1), (1, 1))

    _, y0 = fwd(1)
    dy = torch.randn(y0.shape, generator=g)
    y0.backward(dy.to(dev), retain_graph=True)      # released by the backward; the graph (and its finaliser) stays alive through y0
    xa, ya = fwd(2)                                  # takes the entry step 0 released
    ent_a = ya.grad_fn.xp_ent
    assert ent_a is y0.grad_fn.xp_ent and ent_a[1] and ya.grad_fn.xp_gen == ent_a[2] != y0.grad_fn.xp_gen
    with pytest.raises(RuntimeError, match="later forward"):
        y0.backward(dy.to(dev))                      # step 0's node no longer owns the map
    del y0
    gc.collect()                                     # step 0's finaliser fires: wrong generation, the entry stays A's
    assert ent_a[1]
    xb, yb = fwd(3)
    assert yb.grad_fn.xp_ent is not ent_a            # B got its own map, A's input is intact
    w.grad = None
    ya.backward(dy.to(dev))
    wr = w.detach().double().cpu().requires_grad_(True)
    F.conv2d(xa.double().permute(0, 3, 1, 2), wr, None, padding=1).permute(0, 2, 3, 1).backward(dy.double())
    err = (w.grad.double().cpu() - wr.grad).abs().max().item() / wr.grad.abs().max().item()
    assert err < 2e-5, err
    assert not ent_a[1] and yb.grad_fn.xp_ent[1]     # A released its own entry, B's is still held
