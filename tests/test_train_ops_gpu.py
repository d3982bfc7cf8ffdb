"""GPU parity of the training-path operators (C ABI through anatomix_amd.model.train_ops) against torch autograd on CPU
(float64 on the SAME rounded operands)."""
import pytest
import torch
import torch.nn.functional as F

from _util import rel_l2
from anatomix_amd.model import train_ops as T

pytestmark = pytest.mark.gpu

DT = {"bf16": torch.bfloat16, "f16": torch.float16}
ULP = {"bf16": 2.0 ** -8, "f16": 2.0 ** -11}


def cl(x, dt, device):            # NCDHW float -> NDHWC 16-bit on the device
    return x.permute(0, 2, 3, 4, 1).contiguous().to(dt).to(device)


def ncdhw(x):                     # NDHWC device tensor -> NCDHW double on the host
    return x.detach().cpu().double().permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 16, 8, 12, 16), (2, 32, 16, 16, 16), (1, 256, 2, 2, 2), (3, 64, 5, 6, 7)])
def test_bn_train_forward_and_backward(device, prec, shape):
    dt = DT[prec]
    g = torch.Generator().manual_seed(1)
    n, c = shape[0], shape[1]
    x = (torch.randn(shape, generator=g) * (0.2 + torch.rand(c, generator=g))[None, :, None, None, None]
         + 2.0 * torch.randn(c, generator=g)[None, :, None, None, None])
    gamma, beta = 0.5 + torch.rand(c, generator=g), 0.3 * torch.randn(c, generator=g)
    rm0, rv0 = torch.randn(c, generator=g), 0.5 + torch.rand(c, generator=g)
    dy = torch.randn(shape, generator=g)
    xq = x.to(dt)
    rm, rv = rm0.clone().to(device), rv0.clone().to(device)
    y, mean, rstd = T.bn_train_forward(cl(x, dt, device), gamma.to(device), beta.to(device), 1e-5, "relu", running_mean=rm,
                                       running_var=rv)
    # reference: double, on the rounded input
    xr = xq.double().requires_grad_(True)
    rmr, rvr = rm0.double().clone(), rv0.double().clone()
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = F.relu(F.batch_norm(xr, rmr, rvr, gr, br, True, 0.1, 1e-5))
    assert rel_l2(ncdhw(y), yr.detach()) < ULP[prec]
    torch.testing.assert_close(mean.cpu().double(), xq.double().mean((0, 2, 3, 4)), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rm.cpu().double(), rmr, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(rv.cpu().double(), rvr, rtol=1e-4, atol=1e-5)
    # backward with the kernel's own stored (rounded) y as the activation mask, as the step does
    dyq = dy.to(dt)
    yr.backward(dyq.double())
    framed, dgamma, dbeta = T.bn_act_backward(cl(dy, dt, device), y, cl(x, dt, device), mean, rstd, gamma.to(device), "relu")
    assert framed.shape == (n, shape[2] + 4, shape[3] + 4, shape[4] + 4, c)
    dx = ncdhw(T.interior(framed))
    assert rel_l2(dx, xr.grad) < 4 * ULP[prec], rel_l2(dx, xr.grad)
    fz = framed.clone()
    T.interior(fz).zero_()
    assert not fz.any()                                            # the frame stays zero
    torch.testing.assert_close(dgamma.cpu().double(), gr.grad, rtol=2e-3, atol=2e-3 * gr.grad.abs().max().item())
    torch.testing.assert_close(dbeta.cpu().double(), br.grad, rtol=2e-3, atol=2e-3 * br.grad.abs().max().item())
    # the variant that does not read y (amx_bn_act_backward_recompute): the sign act' needs comes from a x + b -- the same numbers
    fr2, dg2, db2 = T.bn_act_backward(cl(dy, dt, device), None, cl(x, dt, device), mean, rstd, gamma.to(device), "relu",
                                      beta=beta.to(device), recompute=True)
    assert torch.equal(fr2, framed) and torch.equal(dg2, dgamma) and torch.equal(db2, dbeta)


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,size", [(16, 16, (8, 16, 32)), (32, 16, (6, 10, 12)), (64, 128, (4, 4, 4)), (1, 16, (8, 8, 40)),
                                            (16, 32, (2, 2, 2)), (128, 64, (8, 8, 8)),
                                            # wide rows: two x tiles of the 8 x 64 geometry, ragged in y (amx_wgrad.hip)
                                            (16, 16, (5, 8, 64)), (32, 32, (3, 4, 128)), (16, 32, (9, 12, 64)), (1, 16, (4, 8, 128))])
def test_conv_dgrad_and_wgrad(device, prec, cin, cout, size):
    dt = DT[prec]
    g = torch.Generator().manual_seed(2)
    n = 2
    cin_pad = (cin + 15) // 16 * 16
    x = torch.randn(n, cin, *size, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    dy = torch.randn(n, cout, *size, generator=g)
    xq, dyq, wq = x.to(dt).double().requires_grad_(True), dy.to(dt).double(), w.to(dt).double().requires_grad_(True)
    yr = F.conv3d(F.pad(xq, (1,) * 6, mode="reflect"), wq)
    yr.backward(dyq)
    # device side
    xd = torch.zeros((n, *size, cin_pad), dtype=dt, device=device)
    xd[..., :cin] = cl(x, dt, device)
    framed = T.new_framed(n, *size, cout, dt, device)
    T.interior(framed).copy_(cl(dy, dt, device))
    # forward sanity (same entry point the step uses)
    y = T.conv_forward(xd, None, w.to(device))
    assert rel_l2(ncdhw(y), yr.detach()) < 2 * ULP[prec]
    din = T.conv_dgrad(framed, w.to(device))
    assert din.shape[-1] == cin_pad
    assert rel_l2(ncdhw(din)[:, :cin], xq.grad) < 2 * ULP[prec], rel_l2(ncdhw(din)[:, :cin], xq.grad)
    if cin_pad != cin:
        assert not din[..., cin:].any()
    dw = T.conv_wgrad(framed, xd, None, cin, cout)
    assert dw.shape == (cout, cin, 3, 3, 3)
    assert rel_l2(dw.cpu().double(), wq.grad) < 1e-5, rel_l2(dw.cpu().double(), wq.grad)      # fp32 accumulate of exact products
    assert torch.equal(dw, T.conv_wgrad(framed, xd, None, cin, cout))                          # fixed summation order
    base = torch.randn_like(dw)                                                                # accumulate: added to what is there
    acc = T.conv_wgrad(framed, xd, None, cin, cout, out=base.clone(), accumulate=True)
    assert rel_l2(acc.cpu().double(), base.cpu().double() + wq.grad) < 1e-5


@pytest.mark.parametrize("prec", ["bf16"])
@pytest.mark.parametrize("c0,c1,cout,size", [(16, 32, 16, (8, 8, 32)), (32, 64, 32, (4, 8, 12)), (128, 256, 128, (4, 4, 4)),
                                              (16, 32, 16, (6, 8, 64)), (32, 64, 32, (4, 4, 128))])
def test_wgrad_and_dgrad_of_upsample_concat_conv(device, prec, c0, c1, cout, size):
    dt = DT[prec]
    g = torch.Generator().manual_seed(3)
    n = 2
    lo = tuple(s // 2 for s in size)
    skip, low = torch.randn(n, c0, *size, generator=g), torch.randn(n, c1, *lo, generator=g)
    w = torch.randn(cout, c0 + c1, 3, 3, 3, generator=g) / (27 * (c0 + c1)) ** 0.5
    dy = torch.randn(n, cout, *size, generator=g)
    sq, lq = skip.to(dt).double().requires_grad_(True), low.to(dt).double().requires_grad_(True)
    wq = w.to(dt).double().requires_grad_(True)
    cat = torch.cat((sq, F.interpolate(lq, scale_factor=2, mode="nearest")), 1)
    yr = F.conv3d(F.pad(cat, (1,) * 6, mode="reflect"), wq)
    yr.backward(dy.to(dt).double())
    framed = T.new_framed(n, *size, cout, dt, device)
    T.interior(framed).copy_(cl(dy, dt, device))
    sd, ld = cl(skip, dt, device), cl(low, dt, device)
    y = T.conv_forward(sd, ld, w.to(device))
    assert rel_l2(ncdhw(y), yr.detach()) < 2 * ULP[prec]
    dw = T.conv_wgrad(framed, sd, ld, c0 + c1, cout)
    assert rel_l2(dw.cpu().double(), wq.grad) < 1e-5
    dcat = ncdhw(T.conv_dgrad(framed, w.to(device)))
    assert rel_l2(dcat[:, :c0], sq.grad) < 2 * ULP[prec]
    # adjoint of the nearest upsample = sum over the 8 children (host arithmetic in the test; the step uses torch ops too)
    dl = dcat[:, c0:].reshape(n, c1, lo[0], 2, lo[1], 2, lo[2], 2).sum((3, 5, 7))
    assert rel_l2(dl, lq.grad) < 2 * ULP[prec]


@pytest.mark.parametrize("prec", ["bf16", "f16"])
def test_pool2_max_backward_first_max_tie_rule(device, prec):
    dt = DT[prec]
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 16, 8, 8, 8, generator=g).relu()            # many exact ties at 0
    x[0, :, :2, :2, :2] = 1.5                                       # and a non-zero tie
    dp = torch.randn(2, 16, 4, 4, 4, generator=g)
    xq = x.to(dt).double().requires_grad_(True)
    F.max_pool3d(xq, 2).backward(dp.to(dt).double())
    xd = cl(x, dt, device)
    p = T.pool2_max(xd)
    assert torch.equal(ncdhw(p), F.max_pool3d(x.to(dt).double(), 2))
    din = T.pool2_max_backward(cl(dp, dt, device), xd)
    assert torch.equal(ncdhw(din), xq.grad)
    acc = torch.ones_like(xd)
    T.pool2_max_backward(cl(dp, dt, device), xd, accumulate_into=acc)
    assert rel_l2(ncdhw(acc), xq.grad + 1.0) < ULP[prec]


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(2, 16, 4, 6, 8), (1, 8, 2, 2, 2), (1, 24, 7, 5, 3)])
def test_trilinear_upsample_and_its_adjoint(device, prec, shape):
    """nn.Upsample(2, 'trilinear') forward and the adjoint kernel against torch (autograd) on the same rounded operands."""
    dt = torch.float16 if prec == "f16" else torch.bfloat16
    torch.manual_seed(sum(shape))
    n, c, d, h, w = shape
    x = torch.randn(shape).to(dt).double().requires_grad_(True)
    g = torch.randn(n, c, 2 * d, 2 * h, 2 * w).to(dt).double()
    y = F.interpolate(x, scale_factor=2, mode="trilinear")
    y.backward(g)
    yd = T.upsample2_trilinear(cl(x.detach().float(), dt, device))
    gd = T.upsample2_trilinear_backward(cl(g.float(), dt, device))
    tol = 2e-3 if prec == "f16" else 1.2e-2                   # one rounding of the 16-bit result
    assert rel_l2(ncdhw(yd), y.detach()) < tol
    assert rel_l2(ncdhw(gd), x.grad) < tol
    # adjoint identity <U x, g> == <x, U^T g> on the kernels' own outputs
    lhs = (ncdhw(yd) * g).sum().item()
    rhs = (x.detach() * ncdhw(gd)).sum().item()
    assert abs(lhs - rhs) < 2e-2 * (abs(lhs) + abs(rhs)) + 1e-2


def test_avg_pool_forward(device):
    dt = torch.float16
    x = torch.randn(2, 16, 8, 4, 6).to(dt)
    y = T.pool2(cl(x.float(), dt, device), 1)
    assert rel_l2(ncdhw(y), F.avg_pool3d(x.double(), 2)) < 1e-3


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_layout_export_and_strided_accumulating_import(device, prec):
    dt = DT[prec]
    torch.manual_seed(3)
    x = torch.randn(2, 24, 5, 6, 7).to(dt)
    xd = cl(x.float(), dt, device)
    out = T.export_ncdhw(xd)
    assert out.dtype == torch.float32 and torch.equal(out.cpu(), x.float())
    g = torch.randn(2, 24, 5, 6, 7, device=device)
    framed = T.new_framed(2, 5, 6, 7, 32, dt, device)                       # wider voxel pitch than the source: channels 24..31 stay 0
    T.import_ncdhw(g, T.interior(framed))
    want = g.permute(0, 2, 3, 4, 1).to(dt)
    assert torch.equal(T.interior(framed)[..., :24], want) and T.interior(framed)[..., 24:].abs().max().item() == 0
    assert framed[:, :2].abs().max().item() == 0 and framed[:, :, :, -2:].abs().max().item() == 0   # the frame is untouched
    T.import_ncdhw(g, T.interior(framed), accumulate=True)
    assert torch.equal(T.interior(framed)[..., :24], (want.float() + g.permute(0, 2, 3, 4, 1)).to(dt))


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_upcat_split_backward(device, prec):
    """Adjoint of cat(skip, nearest_up2(low)) in one pass against the torch formulation."""
    dt = DT[prec]
    torch.manual_seed(4)
    n, c0, c1, d, h, w = 2, 16, 40, 6, 4, 10
    dcat = torch.randn(n, d, h, w, c0 + c1).to(dt).to(device)
    dskip, dlow = T.upcat_split_backward(dcat, c0, c1)
    assert torch.equal(dskip, dcat[..., :c0])
    want = dcat[..., c0:].reshape(n, d // 2, 2, h // 2, 2, w // 2, 2, c1).float().sum((2, 4, 6)).to(dt)
    assert (dlow.float() - want.float()).abs().max().item() <= ULP[prec] * 4 * want.float().abs().max().item()
    prev = torch.randn(n, d, h, w, c0).to(dt).to(device)
    acc, _ = T.upcat_split_backward(dcat, c0, c1, skip_into=prev.clone())
    assert torch.equal(acc, (prev.float() + dcat[..., :c0].float()).to(dt))


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("size", [(6, 4, 10), (2, 2, 2), (4, 8, 2), (16, 12, 20)])
def test_upcat_split_backward_framed_is_pad_fold_then_split(device, prec, size):
    """amx_upcat_split_backward_framed = amx_pad_fold followed by amx_upcat_split_backward, without the folded tensor in between:
    the skip part is bit-identical (same sums, same rounding), the low-resolution part skips one rounding (fp32 over fold + children)."""
    dt = DT[prec]
    torch.manual_seed(sum(size))
    n, c0, c1 = 2, 16, 24
    d, h, w = size
    g = torch.randn(n, d + 4, h + 4, w + 4, c0 + c1).to(dt).to(device)
    folded = T.pad_fold(g)
    dskip_ref, dlow_ref = T.upcat_split_backward(folded, c0, c1)
    dskip, dlow = T.upcat_split_backward_framed(g, c0, c1)
    assert torch.equal(dskip, dskip_ref)
    # the float64 value of fold + child sum
    gd = g.double()
    ref = _fold64(gd, d, h, w)[..., c0:].reshape(n, d // 2, 2, h // 2, 2, w // 2, 2, c1).sum((2, 4, 6))
    err = (dlow.double() - ref).abs().max().item()
    assert err <= ULP[prec] * 2 * ref.abs().max().item(), err
    assert (dlow.float() - dlow_ref.float()).abs().max().item() <= ULP[prec] * 8 * ref.abs().max().item()
    _, low_only = T.upcat_split_backward_framed(g[..., c0:].contiguous(), 0, c1)       # no skip part at all
    assert torch.equal(low_only, dlow)
    prev = torch.randn(n, d, h, w, c0).to(dt).to(device)
    acc, _ = T.upcat_split_backward_framed(g, c0, c1, skip_into=prev.clone())
    want = prev.double() + _fold64(gd, d, h, w)[..., :c0]            # one rounding of prev + fold (the two-pass form rounds the fold first)
    assert (acc.double() - want).abs().max().item() <= ULP[prec] * want.abs().max().item()


def _fold64(gd, d, h, w):
    """Adjoint of reflect padding by one voxel, in float64, on a [N, D+4, H+4, W+4, C] tensor: padded coordinate j sits at framed
    index j + 2; voxel 1 collects j = -1, voxel L-2 collects j = L."""
    out = gd
    for ax, L in ((1, d), (2, h), (3, w)):
        core = out.narrow(ax, 2, L).clone()
        core.narrow(ax, 1, 1).add_(out.narrow(ax, 1, 1))
        core.narrow(ax, L - 2, 1).add_(out.narrow(ax, L + 2, 1))
        out = core
    return out


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,size", [(16, 16, (8, 8, 32)), (16, 16, (16, 24, 64)), (32, 32, (8, 12, 32)), (16, 32, (8, 8, 40)),
                                           (16, 16, (9, 10, 35)), (32, 16, (8, 8, 32))])
def test_direct_data_gradient_equals_the_framed_route(device, prec, cin, cout, size):
    """conv_dgrad_direct = the forward kernel on the interior of the zero-framed gradient (halo read from the frame) + the folded shell
    terms of the reflect adjoint, against (a) autograd of the fp64 reflect-padded conv on the same rounded operands and (b) the
    framed-domain route (conv on (n+4)^3 + pad_fold) it replaces.  Shapes: every z-march class (16 -> 16, 32 -> 32, and the two halves
    of the split concat gradient), partial tiles, and one the direct route does not take (asks ``dgrad_direct_supported``)."""
    dt = DT[prec]
    g = torch.Generator().manual_seed(7)
    n = 2
    x = torch.randn(n, cin, *size, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    dy = torch.randn(n, cout, *size, generator=g)
    xq, dyq, wq = x.to(dt).double().requires_grad_(True), dy.to(dt).double(), w.to(dt).double()
    F.conv3d(F.pad(xq, (1,) * 6, mode="reflect"), wq).backward(dyq)
    framed = T.new_framed(n, *size, cout, dt, device)
    T.interior(framed).copy_(cl(dy, dt, device))
    wd = w.to(device)
    ok = T.dgrad_direct_supported(framed, wd)
    assert ok == (cin in (16, 32) and cout in (16, 32) and not (cout == 32 and cin == 16) and size[2] >= 32 and size[0] >= 8 and size[1] >= 8)
    if not ok:
        return
    old = T.pad_fold(T.conv_dgrad_framed(framed, wd))
    new = T.conv_dgrad_direct(framed, wd)
    assert new.shape == old.shape
    assert not framed[:, :2].any() and not framed[:, -2:].any() and not framed[:, :, :2].any() and not framed[..., :2, :].any()
    e_new, e_old = rel_l2(ncdhw(new)[:, :cin], xq.grad), rel_l2(ncdhw(old)[:, :cin], xq.grad)
    assert e_new < 2 * ULP[prec], (e_new, e_old)
    # the near-face voxels are where the two routes differ in arithmetic: check them on their own
    face = torch.zeros(size, dtype=torch.bool)
    for a, s in enumerate(size):
        idx = [slice(None)] * 3
        for v in (1, s - 2):
            idx[a] = v
            face[tuple(idx)] = True
    ref = xq.grad[:, :, face]
    assert rel_l2(ncdhw(new)[:, :cin][:, :, face], ref) < 3 * ULP[prec]
    assert torch.equal(T.conv_dgrad_direct(framed, wd), new)          # deterministic


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("cin", [1, 2, 16])
def test_import_input_pads_to_sixteen_channels(device, prec, cin):
    """amx_import_input: fp32 NCDHW -> 16-bit channels-last with 16 stored channels, bit for bit what zero-fill + cast + copy give."""
    dt = DT[prec]
    x = torch.randn(2, cin, 6, 10, 20, generator=torch.Generator().manual_seed(3)).to(device) * 3
    got = T.import_input(x, dt)
    want = torch.zeros(2, 6, 10, 20, 16, dtype=dt, device=device)
    want[..., :cin] = x.permute(0, 2, 3, 4, 1).to(dt)
    assert got.shape == want.shape and got.dtype == dt and torch.equal(got, want)


@pytest.mark.parametrize("prec", ["bf16", "f16"])
@pytest.mark.parametrize("size,cin,cout,views,patches", [((8, 12, 16), 16, 16, 2, 64), ((6, 6, 6), 16, 5, 1, 200), ((16, 16, 32), 9, 16, 2, 512)])
def test_sampled_output_conv_backward_equals_the_dense_route(device, prec, size, cin, cout, views, patches):
    """amx_conv3d_backward_sampled (the output conv tapped at sampled voxels only) against the dense route it replaces: rows
    scattered into a zero gradient volume, dense weight gradient, dense data gradient + reflect fold.  Same rounding points; the sums
    run in another order (weight gradient: fp32 noise) and the data gradient is rounded once where the dense route rounds the padded
    domain and the fold separately (voxels next to a face; neighbourhoods of many samples overlap at these small sizes on purpose)."""
    dt = DT[prec]
    g = torch.Generator().manual_seed(sum(size) + cin + cout)
    d, h, w = size
    nvox = d * h * w
    flat = torch.randperm(nvox, generator=g)[:patches]
    coords = torch.stack([flat // (h * w), (flat // w) % h, flat % w], 1).to(device)
    x0 = torch.zeros((views, d, h, w, 16), dtype=dt, device=device)
    x0[..., :cin] = torch.randn(views, d, h, w, cin, generator=g).to(dt).to(device)
    wt = (torch.randn(cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5).to(device)
    rows = torch.randn(views, patches, cout, generator=g).to(device)
    dw, din = T.conv_backward_sampled(rows, coords, x0, wt, cin)
    dw2, din2 = T.conv_backward_sampled(rows, coords, x0, wt, cin)
    assert torch.equal(dw, dw2) and torch.equal(din, din2)                    # fixed order, benign duplicate writes
    cpad = (cout + 15) // 16 * 16
    fr = T.new_framed(views, d, h, w, cpad, dt, device)
    T.scatter_rows(rows, coords, T.interior(fr), accumulate=True)
    dw_d = T.conv_wgrad(fr, x0, None, cin, cpad)[:cout]
    din_d = T.pad_fold(T.conv_dgrad_framed(fr, torch.cat([wt, wt.new_zeros(cpad - cout, cin, 3, 3, 3)]) if cpad != cout else wt))
    assert rel_l2(dw.cpu().double(), dw_d.cpu().double()) < 1e-6
    a, b = din[..., :cin].float().cpu(), din_d[..., :cin].float().cpu()
    assert rel_l2(a.double(), b.double()) < 2 * ULP[prec]
    assert not din[..., cin:].any()
    zero_d = (b == 0).all(-1)
    assert torch.equal((a == 0).all(-1) | ~zero_d, torch.ones_like(zero_d))    # nothing written where the dense gradient is zero
