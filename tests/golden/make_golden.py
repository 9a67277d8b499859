#!/usr/bin/env python3
"""Generate the golden vectors in this directory by running the REAL reference.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
The reference is pure Python; four third-party imports it makes at module import time are absent
here and are not touched by the hot path's arithmetic, so they are stubbed.
Nothing of the reference is written into this repo: the outputs are DATA (inputs, parameters,
expected outputs) stored as .npz.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Fixtures:
  g1_squeeze_split.npz   squeeze/unsqueeze, split/cat
  g2_actnorm.npz         data-dependent init; fwd/rev with logdet in {None, tensor}
  g2_actnorm_bv.npz      data-dependent init with batch_variance=True (one pooled log-scale), scale 1 and 3
  g3_invconv.npz         C in {12,24,48,96}, non-orthogonal W: fwd, dlogdet, rev
  g4_coupling_net.npz    f(): 3x3 -> relu -> 1x1 -> relu -> zeros-3x3, all params random
  g5_flowstep.npz        {invconv,reverse,shuffle} x {additive,affine}: fwd z/logdet, rev x/logdet
  g6_split2d.npz         fwd logp; rev with injected eps for eps_std in {None, 0, 0.7}
  g7_glow_tiny.npz       Glow 16x16x3 L=2 K=2 hidden 32 (affine+invconv, additive+reverse):
                         noise, z, nll, decode with injected eps, data-dependent init
  g7_glow_tiny_grads.npz d mean(nll) / d theta for every parameter + d / d x of the two G7 models, from the reference's own backward
                         (F4 shim: misc.ops.split_channel returning clones)
  g7_glow_tiny_learn_top.npz the same model with a learned top prior (ablation.learn_top): z, nll, every gradient incl. learn_top.bias / logs
  g9_inferer.npz + g9_reference_snapshot.pth
                         a snapshot WRITTEN BY the reference (misc/util.py save_model, after one Adam step) and what the
                         reference's Inferer (network/inferer.py) computes from it: encode, attribute deltaz (with the data
                         loader order it used), make_interpolation_vector
  g8_glow_celeba64.npz   celeba.json-sized model (64x64x3 L3 K32 w512 affine), B=2, weights from the
                         oracle's seeded procedure (too big to commit): digests only
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

import numpy as np
import torch


def _install_stubs():
    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        __setattr__ = __setitem__

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("cv2")
    tv = mod("torchvision")
    tr = mod("torchvision.transforms", transforms=types.SimpleNamespace())
    ut = mod("torchvision.utils", make_grid=lambda *a, **k: None)
    tv.transforms, tv.utils = tr, ut
    mod("tensorboardX", SummaryWriter=object)
    mod("easydict", EasyDict=EasyDict)
    return EasyDict


EasyDict = _install_stubs()
sys.path.insert(0, REF)
os.chdir(REF)
from network import module as rmod  # noqa: E402
from network import model as rmodel  # noqa: E402
from misc import ops as rops  # noqa: E402

sys.path.insert(0, REPO)
from oracle import glow_oracle as O  # noqa: E402  (only for the seeded weight procedure of G8)


def npd(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **npd(d))
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(d)} arrays")


def randomize_(m, g, std=0.2):
    """Randomise every parameter of a reference module in place (zero-init layers included)."""
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("invconv.weight") or name == "weight" and p.dim() == 2:
                c = p.shape[0]
                q = np.linalg.qr(np.random.randn(c, c))[0].astype("float32")
                p.copy_(torch.from_numpy(q) + 0.1 * torch.randn(c, c, generator=g))
            elif name.endswith("logs"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            elif name.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            elif "f.4." in name or "conv2d_zeros" in name or name == "weight" and getattr(m, "logscale_factor", None):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            else:
                p.copy_(torch.randn(p.shape, generator=g) * std)
    for sub in m.modules():
        if isinstance(sub, rmod.ActNorm):
            sub.bias_inited = True
            sub.logs_inited = True


def g1(g):
    x = torch.randn(2, 3, 8, 8, generator=g)
    sq = rmod.Squeeze2d.squeeze(x.clone(), 2)
    x2 = torch.randn(2, 8, 4, 6, generator=g)
    un = rmod.Squeeze2d.unsqueeze(x2.clone(), 2)
    t = torch.randn(2, 6, 4, 4, generator=g)
    s1, s2 = rops.split_channel(t, "simple")
    c1, c2 = rops.split_channel(t, "cross")
    save("g1_squeeze_split.npz", dict(x=x, squeezed=sq, x2=x2, unsqueezed=un, t=t, simple_a=s1, simple_b=s2,
                                      cross_a=c1, cross_b=c2, cat=rops.cat_channel(s1, s2),
                                      rsum=rops.reduce_sum(t.clone(), dim=[1, 2, 3]),
                                      rmean=rops.reduce_mean(t.clone(), dim=[0, 2, 3], keepdim=True)))


def g2(g):
    out = {}
    x = torch.randn(4, 12, 8, 8, generator=g) * 1.7 + 0.3
    an = rmod.ActNorm(12, scale=1.0)
    an.train()
    y, _ = an(x.clone())
    out.update(init_x=x, init_bias=an.bias.data.clone(), init_logs=an.logs.data.clone(), init_y=y)
    an3 = rmod.ActNorm(12, scale=3.0)
    an3.train()
    an3(x.clone())
    out.update(init3_bias=an3.bias.data.clone(), init3_logs=an3.logs.data.clone())
    an = rmod.ActNorm(12)
    randomize_(an, g)
    ld = torch.randn(4, generator=g)
    yf, ldf = an(x.clone(), ld.clone(), reverse=False)
    yr, ldr = an(x.clone(), ld.clone(), reverse=True)
    yn, ldn = an(x.clone(), None, reverse=False)
    assert ldn is None
    out.update(x=x, bias=an.bias, logs=an.logs, logdet=ld, fwd_y=yf, fwd_logdet=ldf, rev_y=yr, rev_logdet=ldr, fwd_y_nold=yn)
    save("g2_actnorm.npz", out)


def g2_bv():
    """ActNorm(batch_variance=True), network/module.py:109-110 (own generator: the other fixtures' draws stay what they were)."""
    g = torch.Generator().manual_seed(4321)
    out = {}
    x = torch.randn(4, 12, 8, 8, generator=g) * torch.linspace(0.5, 3.0, 12).view(1, 12, 1, 1) + 0.3
    for name, scale in (("bv", 1.0), ("bv3", 3.0)):
        an = rmod.ActNorm(12, scale=scale, batch_variance=True)
        an.train()
        ld = torch.randn(4, generator=g)
        y, ldo = an(x.clone(), ld.clone())
        out.update({f"{name}_bias": an.bias.data.clone(), f"{name}_logs": an.logs.data.clone(), f"{name}_y": y,
                    f"{name}_logdet_in": ld, f"{name}_logdet": ldo})
    out["x"] = x
    save("g2_actnorm_bv.npz", out)


def g3(g):
    out = {}
    for c in (12, 24, 48, 96):
        inv = rmod.Invertible1x1Conv(c)
        randomize_(inv, g)
        x = torch.randn(2, c, 4, 4, generator=g)
        ld = torch.randn(2, generator=g)
        zf, ldf = inv(x.clone(), ld.clone(), reverse=False)
        zr, ldr = inv(x.clone(), ld.clone(), reverse=True)
        out.update({f"c{c}_w": inv.weight, f"c{c}_x": x, f"c{c}_logdet": ld, f"c{c}_fwd_z": zf, f"c{c}_fwd_logdet": ldf,
                    f"c{c}_rev_z": zr, f"c{c}_rev_logdet": ldr})
    save("g3_invconv.npz", out)


def g4(g):
    net = rmod.f(6, 32, 12)
    randomize_(net, g)
    x = torch.randn(4, 6, 8, 8, generator=g)
    with torch.no_grad():
        y = net(x.clone())
    out = {"x": x, "y": y}
    out.update({"p." + k: v for k, v in net.state_dict().items()})
    # a Conv2d / Conv2dZeros pair of odd shape (reference test_module.py:31-48 uses 16 -> 5)
    cv = rmod.Conv2d(16, 5)
    randomize_(cv, g)
    cz = rmod.Conv2dZeros(16, 5)
    randomize_(cz, g)
    c1 = rmod.Conv2d(16, 7, kernel_size=1)
    randomize_(c1, g)
    x2 = torch.randn(2, 16, 4, 4, generator=g)
    with torch.no_grad():
        out.update(x2=x2, conv_y=cv(x2.clone()), convz_y=cz(x2.clone()), conv1_y=c1(x2.clone()))
    out.update({"conv." + k: v for k, v in cv.state_dict().items()})
    out.update({"convz." + k: v for k, v in cz.state_dict().items()})
    out.update({"conv1." + k: v for k, v in c1.state_dict().items()})
    save("g4_coupling_net.npz", out)


def g5(g):
    out = {}
    for perm in ("invconv", "reverse", "shuffle"):
        for coup in ("additive", "affine"):
            tag = f"{perm}_{coup}"
            st = rmodel.FlowStep(12, 32, permutation=perm, coupling=coup)
            randomize_(st, g)
            x = torch.randn(4, 12, 8, 8, generator=g)
            ld = torch.randn(4, generator=g)
            with torch.no_grad():
                z, ldz = st(x.clone(), ld.clone(), reverse=False)
                xr, ldx = st(x.clone(), ld.clone(), reverse=True)
            out.update({f"{tag}.x": x, f"{tag}.logdet": ld, f"{tag}.fwd_z": z, f"{tag}.fwd_logdet": ldz,
                        f"{tag}.rev_x": xr, f"{tag}.rev_logdet": ldx})
            out.update({f"{tag}.p.{k}": v for k, v in st.state_dict().items()})
            if perm != "invconv":
                pm = getattr(st, perm)
                out[f"{tag}.indices"] = pm.indices.copy()
                out[f"{tag}.indices_inverse"] = pm.indices_inverse.copy()
    save("g5_flowstep.npz", out)


class EpsTap:
    """Replace GaussianDiag.eps by a recorder (reference module.py:408-421)."""

    def __init__(self):
        self.draws, self.stds = [], []
        self.orig = rmod.GaussianDiag.eps

    def __enter__(self):
        tap = self

        def eps(shape_tensor, eps_std=None):
            e = tap.orig(shape_tensor, eps_std)
            tap.draws.append(e.clone())
            tap.stds.append(eps_std)
            return e

        rmod.GaussianDiag.eps = staticmethod(eps)
        return self

    def __exit__(self, *a):
        rmod.GaussianDiag.eps = staticmethod(self.orig)


def g6(g):
    sp = rmod.Split2d(12)
    randomize_(sp, g)
    x = torch.randn(4, 12, 8, 8, generator=g)
    ld = torch.randn(4, generator=g)
    out = {"x": x, "logdet": ld}
    out.update({"p." + k: v for k, v in sp.state_dict().items()})
    with torch.no_grad():
        z1, ldf = sp(x.clone(), ld.clone(), reverse=False)
        out.update(fwd_z1=z1, fwd_logdet=ldf)
        for tag, std in (("none", None), ("zero", 0), ("p7", 0.7)):
            with EpsTap() as tap:
                torch.manual_seed(77)
                xr, _ = sp(z1.clone(), 0., reverse=True, eps_std=std)
            out[f"rev_{tag}_eps"] = tap.draws[0]
            out[f"rev_{tag}_x"] = xr
    save("g6_split2d.npz", out)


def tiny_hps(coupling, permutation, batch=4):
    return EasyDict(dict(
        model=dict(image_shape=[16, 16, 3], hidden_channels=32, K=2, L=2, actnorm_scale=1.0, n_bits_x=8, weight_y=0.0),
        ablation=dict(learn_top=False, y_condition=False, lu_decomposition=False, flow_permutation=permutation,
                      flow_coupling=coupling),
        optim=dict(num_batch_train=batch), dataset=dict(num_classes=1), device=dict(graph=["cpu"])))


def run_glow(glow, x, seed):
    """Forward with recoverable dequantisation noise (reference model.py:421)."""
    torch.manual_seed(seed)
    z, nll, _ = glow(x=x.clone(), reverse=False)
    torch.manual_seed(seed)
    noise = torch.nn.init.uniform_(torch.empty(*x.shape), 0, 1. / 2 ** glow.hps.model.n_bits_x)
    return z, nll, noise


def g7(g):
    out = {}
    for coup, perm in (("affine", "invconv"), ("additive", "reverse")):
        tag = f"{coup}_{perm}"
        np.random.seed(5)
        glow = rmodel.Glow(tiny_hps(coup, perm))
        x = torch.rand(4, 3, 16, 16, generator=g)
        # (a) data-dependent init pass on fresh weights with non-zero tails
        with torch.no_grad():
            for n, p in glow.named_parameters():
                if "f.4." in n or "conv2d_zeros" in n:
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)
        pre = {k: v.clone() for k, v in glow.state_dict().items()}
        glow.train()
        with torch.no_grad():
            z0, nll0, noise0 = run_glow(glow, x, 11)
        post = {k: v.clone() for k, v in glow.state_dict().items()}
        out.update({f"{tag}.x": x, f"{tag}.init_noise": noise0, f"{tag}.init_z": z0, f"{tag}.init_nll": nll0})
        out.update({f"{tag}.pre.{k}": v for k, v in pre.items()})
        out.update({f"{tag}.post.{k}": v for k, v in post.items()})
        # (b) eval forward with fully randomised weights
        randomize_(glow, g, std=0.1)
        with torch.no_grad():
            glow.h_top.zero_()
        glow.eval()
        sd = {k: v.clone() for k, v in glow.state_dict().items()}
        with torch.no_grad():
            z, nll, noise = run_glow(glow, x, 12)
            with EpsTap() as tap:
                torch.manual_seed(13)
                xr = glow(z=z.clone(), eps_std=0.6, reverse=True)
        out.update({f"{tag}.noise": noise, f"{tag}.z": z, f"{tag}.nll": nll, f"{tag}.dec_x": xr})
        for j, e in enumerate(tap.draws):
            out[f"{tag}.dec_eps{j}"] = e
        out.update({f"{tag}.sd.{k}": v for k, v in sd.items()})
        if perm != "invconv":
            for i, layer in enumerate(glow.flow.layers):
                if hasattr(layer, perm):
                    out[f"{tag}.indices.{i}"] = getattr(layer, perm).indices.copy()
                    out[f"{tag}.indices_inverse.{i}"] = getattr(layer, perm).indices_inverse.copy()
    save("g7_glow_tiny.npz", out)


def g7_grads():
    """SURVEY 8c G7, second half: d mean(nll) / d theta for EVERY parameter and d mean(nll) / d x, recorded from the reference's own
    backward.  Inputs are the committed G7 fixture (eval state_dict `sd`, batch `x`, dequantisation noise `noise`), so the other
    fixtures' random streams are untouched.  The reference's backward trips over its in-place coupling ops (`z2 += ...` on a VIEW
    made by misc.ops.split_channel, network/model.py:105-113): the F4 shim of SURVEY 8c -- split_channel returning CLONES of the
    two halves -- changes no value and lets autograd run."""
    fx = dict(np.load(os.path.join(HERE, "g7_glow_tiny.npz")))
    orig_split = rops.split_channel
    shim = lambda t, s="simple": tuple(a.clone() for a in orig_split(t, s))
    rops.split_channel = shim
    out = {}
    try:
        for coup, perm in (("affine", "invconv"), ("additive", "reverse")):
            tag = f"{coup}_{perm}"
            np.random.seed(5)
            glow = rmodel.Glow(tiny_hps(coup, perm))
            sd = {k[len(tag) + 4:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith(tag + ".sd.")}
            glow.load_state_dict(sd)
            if perm != "invconv":           # the fixed channel permutations are attributes, not parameters
                for i, layer in enumerate(glow.flow.layers):
                    if hasattr(layer, perm):
                        getattr(layer, perm).indices = fx[f"{tag}.indices.{i}"].copy()
                        getattr(layer, perm).indices_inverse = fx[f"{tag}.indices_inverse.{i}"].copy()
            glow.set_actnorm_inited()
            glow.eval()
            x = torch.from_numpy(fx[f"{tag}.x"]).clone().requires_grad_(True)
            z, nll, noise = run_glow(glow, x, 12)          # same seed as G7 (b): the same noise
            assert np.array_equal(noise.numpy(), fx[f"{tag}.noise"]) and np.allclose(nll.detach().numpy(), fx[f"{tag}.nll"], atol=1e-6)
            loss = rmodel.Glow.generative_loss(nll)
            loss.backward()
            out[f"{tag}.loss"] = loss.detach()
            out[f"{tag}.dx"] = x.grad.detach()
            n_none = 0
            for name, p_ in glow.named_parameters():
                if p_.grad is None:
                    n_none += 1          # h_top: detached by the reference (model.py:372)
                    continue
                out[f"{tag}.grad.{name}"] = p_.grad.detach()
            assert n_none == 1, n_none
    finally:
        rops.split_channel = orig_split
    save("g7_glow_tiny_grads.npz", out)


def g7_learn_top():
    """Glow with a LEARNED top prior (ablation.learn_top, network/model.py:340-343, 375-376): forward nll / z and the gradients of
    mean(nll) for every parameter (F4 shim as in g7_grads) on the tiny affine / invconv model with randomised parameters --
    learn_top.bias / .logs included (its weight sees an all-zero input: zero gradient)."""
    orig_split = rops.split_channel
    rops.split_channel = lambda t, s="simple": tuple(a.clone() for a in orig_split(t, s))
    out = {}
    try:
        g = torch.Generator().manual_seed(4321)
        np.random.seed(7)
        hps = tiny_hps("affine", "invconv")
        hps.ablation.learn_top = True
        glow = rmodel.Glow(hps)
        randomize_(glow, g, std=0.1)
        with torch.no_grad():
            glow.h_top.zero_()
            glow.learn_top.bias.copy_(torch.randn(glow.learn_top.bias.shape, generator=g) * 0.3)
            glow.learn_top.logs.copy_(torch.randn(glow.learn_top.logs.shape, generator=g) * 0.1)
        glow.set_actnorm_inited()
        glow.eval()
        sd = {k: v.clone() for k, v in glow.state_dict().items()}
        x = torch.rand(4, 3, 16, 16, generator=g).requires_grad_(True)
        z, nll, noise = run_glow(glow, x, 31)
        loss = rmodel.Glow.generative_loss(nll)
        loss.backward()
        out.update({"x": x.detach(), "noise": noise, "z": z.detach(), "nll": nll.detach(), "loss": loss.detach(), "dx": x.grad.detach()})
        out.update({f"sd.{k}": v for k, v in sd.items()})
        for name, p_ in glow.named_parameters():
            if p_.grad is not None:
                out[f"grad.{name}"] = p_.grad.detach()
        assert "grad.learn_top.bias" in out and "grad.learn_top.logs" in out and float(out["grad.learn_top.bias"].abs().max()) > 0
    finally:
        rops.split_channel = orig_split
    save("g7_glow_tiny_learn_top.npz", out)


def g8():
    """celeba.json-sized model driven by the ORACLE's seeded weight procedure: commit digests only."""
    cfg = O.default_cfg(batch=2)
    hps = EasyDict(dict(
        model=dict(image_shape=[64, 64, 3], hidden_channels=512, K=32, L=3, actnorm_scale=1.0, n_bits_x=8, weight_y=0.0),
        ablation=dict(learn_top=False, y_condition=False, lu_decomposition=False, flow_permutation="invconv",
                      flow_coupling="affine"),
        optim=dict(num_batch_train=2), dataset=dict(num_classes=40), device=dict(graph=["cpu"])))
    glow = rmodel.Glow(hps)
    sd = O.seeded_state_dict(cfg, seed=2384)
    glow.load_state_dict(sd)
    x = torch.rand(2, 3, 64, 64, generator=torch.Generator().manual_seed(2384))
    glow.train()  # data-dependent init on this batch, as trainer.py:112-115
    with torch.no_grad():
        z0, nll0, noise0 = run_glow(glow, x, 21)
    glow.eval()
    with torch.no_grad():
        z, nll, noise = run_glow(glow, x, 22)
        with EpsTap() as tap:
            torch.manual_seed(23)
            xr = glow(z=z.clone(), eps_std=0.7, reverse=True)
    post = glow.state_dict()
    out = dict(seed=2384, eps_std=0.7, x=x, init_noise=noise0, noise=noise,
               w_digest=torch.stack([sd["flow.layers.50.f.2.weight"].double().sum(),
                                     sd["flow.layers.100.f.4.weight"].double().sum(),
                                     sd["flow.layers.1.invconv.weight"].double().sum()]),
               init_nll=nll0, init_z_corner=z0[:, :, 0, 0], nll=nll, z_corner=z[:, :, 0, 0],
               z_sum=z.double().sum(), z_sumsq=(z.double() ** 2).sum(),
               dec_x_corner=xr[:, :, :4, :4], dec_x_sum=xr.double().sum(),
               an_bias_1=post["flow.layers.1.actnorm.bias"], an_logs_1=post["flow.layers.1.actnorm.logs"],
               an_bias_last=post["flow.layers.100.actnorm.bias"], an_logs_last=post["flow.layers.100.actnorm.logs"],
               f2_logs_50=post["flow.layers.50.f.2.actnorm.logs"])
    for j, e in enumerate(tap.draws):
        out[f"dec_eps{j}"] = e
    save("g8_glow_celeba64.npz", out)


def g9(g):
    """Builder / Inferer row: reference-written snapshot + reference Inferer outputs.  n_bits_x = 24 makes the dequantisation
    noise (<= 6e-8) irrelevant, so the outputs do not depend on a random stream the HIP path cannot replay."""
    import shutil, tempfile
    from misc import util as rutil
    from network import inferer as rinf
    hps = tiny_hps("affine", "invconv", batch=4)
    hps.model.n_bits_x = 24
    hps.dataset = EasyDict(num_classes=3, num_workers=0)
    np.random.seed(9)
    glow = rmodel.Glow(hps)
    randomize_(glow, g, std=0.1)
    with torch.no_grad():
        glow.h_top.zero_()
    xs = torch.rand(12, 3, 16, 16, generator=g)
    ys = (torch.rand(12, 3, generator=g) > 0.5).float()
    # one optimiser step so that the snapshot carries a real Adam state
    opt = torch.optim.Adam(glow.parameters(), lr=1e-4, betas=(0.9, 0.9999), eps=1e-8)
    # (the reference's own backward trips over its in-place coupling ops under this torch version: synthetic gradients)
    for p_ in glow.parameters():
        p_.grad = torch.randn(p_.shape, generator=g) * 1e-2
    opt.step()
    for p_ in glow.parameters():
        p_.grad = None
    with torch.no_grad():
        glow.h_top.zero_()
    glow.eval()
    tmp = tempfile.mkdtemp()
    rutil.save_model(tmp, 7, glow, opt, 1.5, is_best=True)
    shutil.copy(os.path.join(tmp, rutil.get_model_name(7)), os.path.join(HERE, "g9_reference_snapshot.pth"))
    shutil.rmtree(tmp)

    inf = rinf.Inferer(hps, glow, devices=["cpu"], data_device="cpu")
    with torch.no_grad():
        z_all = torch.cat([glow(x=xs[i:i + 4].clone())[0] for i in range(0, 12, 4)])
        nll_all = torch.cat([glow(x=xs[i:i + 4].clone())[1] for i in range(0, 12, 4)])
    z_enc = inf.encode(xs[5].clone())

    order = []

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 12

        def __getitem__(self, i):
            order.append(i)
            return {"x": xs[i].clone(), "y_onehot": ys[i].clone()}

    # The accumulation `attrs_z_pos[cls] += z[i]` (numpy array += torch tensor, inferer.py:139) raises under numpy 2; the loop is
    # run unchanged on a graph wrapper that hands z over as a numpy array.
    class NumpyZ:
        def __init__(self, graph):
            self.graph, self.h_top, self.flow = graph, graph.h_top, graph.flow

        def eval(self):
            return self

        def __call__(self, x):
            z, nll, yl = self.graph(x)
            return z.numpy(), nll, yl

    inf2 = rinf.Inferer(hps, NumpyZ(glow), devices=["cpu"], data_device="cpu")
    torch.manual_seed(77)
    deltaz = inf2.compute_attribute_delta(DS())
    out = dict(xs=xs, ys=ys, z_all=z_all, nll_all=nll_all, z_enc=z_enc, enc_index=5, deltaz=deltaz,
               order=np.asarray(order), loader_seed=77, step=7, seconds=1.5,
               interp=rutil.make_interpolation_vector(3, step=0.5), model_name=np.frombuffer(rutil.get_model_name(7).encode(), dtype=np.uint8),
               best_name=np.frombuffer(rutil.get_best_model_name().encode(), dtype=np.uint8))
    save("g9_inferer.npz", out)
    print("g9 snapshot:", os.path.getsize(os.path.join(HERE, "g9_reference_snapshot.pth")) // 1024, "KiB; loader order", order)


if __name__ == "__main__":
    torch.set_num_threads(8)
    np.random.seed(1234)
    g = torch.Generator().manual_seed(1234)
    if os.environ.get("ONLY") == "g9":
        g9(torch.Generator().manual_seed(99))
        sys.exit(0)
    if os.environ.get("ONLY") == "g2_bv":
        g2_bv()
        sys.exit(0)
    if os.environ.get("ONLY") == "g7_grads":
        g7_grads()
        sys.exit(0)
    if os.environ.get("ONLY") == "g7_learn_top":
        g7_learn_top()
        sys.exit(0)
    g1(g); g2(g); g3(g); g4(g); g5(g); g6(g); g7(g)
    g2_bv()
    g7_grads()
    g7_learn_top()
    g8()
    g9(torch.Generator().manual_seed(99))
    leftovers = [os.path.join(r, d) for r, ds, _ in os.walk(REF) for d in ds if d == "__pycache__"]
    assert not leftovers, f"bytecode written into the reference tree: {leftovers}"
