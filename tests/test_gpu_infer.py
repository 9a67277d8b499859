"""GPU tests of the callers either side of the flow path: a snapshot WRITTEN BY THE REFERENCE loads
through this package's Builder and reproduces the reference's outputs; the Inferer against the reference's own results
(tests/golden/g9_inferer.npz, made by tests/golden/make_golden.py from the real reference) and against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import pytorch_glow_amd as G  # noqa: E402
from pytorch_glow_amd.misc import util  # noqa: E402
from pytorch_glow_amd.network import Builder, Inferer  # noqa: E402
from oracle import glow_oracle as O  # noqa: E402
from conftest import GOLDEN, load_golden  # noqa: E402
from test_host import _g9_hps  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def built():
    hps = _g9_hps()
    hps.general.pre_trained = os.path.join(GOLDEN, "g9_reference_snapshot.pth")
    state = Builder(hps).build(training=False)
    return hps, state, load_golden("g9_inferer")


def test_builder_loads_the_reference_snapshot_and_reproduces_its_outputs(built):
    hps, state, g = built
    assert state["step"] == 7 and state["optimizer"] is None and state["devices"] == [0]
    glow = state["graph"].eval()
    assert glow.h_top.device.type == "cuda"
    xs = g["xs"].to(DEV)
    for i in range(0, 12, 4):            # n_bits_x = 24: the dequantisation noise (<= 6e-8) is below fp32 resolution here
        z, nll, _ = glow(xs[i:i + 4])
        assert (z.cpu() - g["z_all"][i:i + 4]).abs().max().item() < 2e-5
        assert (nll.cpu() - g["nll_all"][i:i + 4]).abs().max().item() < 2e-5


def test_builder_training_state(built, tmp_path):
    hps, _, _ = built
    hps2 = _g9_hps()
    hps2.general.pre_trained = hps.general.pre_trained
    hps2.general.result_dir = str(tmp_path)
    st = Builder(hps2).build(training=True)
    assert isinstance(st["optimizer"], torch.optim.Adam) and len(st["optimizer"].state) > 0     # Adam state of the snapshot
    assert st["scheduler"](global_step=0) == pytest.approx(1e-4 / 10)
    assert os.path.basename(st["result_subdir"]) == "000-g9"
    hps3 = _g9_hps()
    with pytest.raises(RuntimeError, match="No pre-trained model"):
        Builder(hps3).build(training=False)


def test_inferer_encode_and_attribute_delta_match_the_reference(built):
    hps, state, g = built
    inf = Inferer(hps, state["graph"], state["devices"], state["data_device"])
    assert inf.batch_size == 4 and inf.num_classes == 3
    z = inf.encode(g["xs"][int(g["enc_index"])])
    assert z.shape == (48 // 4 * 1, 4, 4) or z.dim() == 3
    assert (z.cpu() - g["z_enc"]).abs().max().item() < 2e-5

    xs, ys = g["xs"], g["ys"]
    data = [{"x": xs[i], "y_onehot": ys[i]} for i in range(12)]
    torch.manual_seed(int(g["loader_seed"]))      # the data loader's shuffle is the first consumer of the default generator
    dz_ref_mode = inf.compute_attribute_delta(data, samples_per_batch="reference", num_workers=0)
    assert np.abs(dz_ref_mode - g["deltaz"].numpy()).max() < 2e-5, "the reference's loop (2 samples per batch) as written"

    dz = inf.compute_attribute_delta(data, shuffle=False, num_workers=0)
    want = O.attribute_delta(g["z_all"].numpy(), ys.numpy(), batch_size=4)
    assert np.abs(dz - want).max() < 2e-5, "every sample, against the oracle's restatement"


def test_inferer_decode_sample_and_attribute_manipulation(built):
    hps, state, g = built
    glow = state["graph"]
    inf = Inferer(hps, glow, state["devices"], state["data_device"])
    z = inf.encode(g["xs"][0])
    torch.manual_seed(5)
    img = inf.decode(z)
    torch.manual_seed(5)
    direct = glow(z=util.make_batch(z, 4), y_onehot=None, reverse=True)[0]
    assert img.shape == (3, 16, 16) and torch.equal(img, direct)
    torch.manual_seed(6)
    s = inf.sample(z=None, y_onehot=None, eps_std=0.5)
    assert s.shape == (4, 3, 16, 16) and torch.isfinite(s).all()
    deltaz = g["deltaz"].numpy()
    interp = [0.5, 0.0, -1.0]
    torch.manual_seed(7)
    out = inf.apply_attribute_delta(g["xs"][0], deltaz, interp)
    torch.manual_seed(7)
    z0 = inf.encode(g["xs"][0])
    zi = z0 + sum(torch.as_tensor(deltaz[c], dtype=torch.float32, device=DEV) * interp[c] for c in range(3))
    want = inf.decode(zi)
    assert (out - want).abs().max().item() < 1e-5
    with pytest.raises(AssertionError):
        inf.apply_attribute_delta(g["xs"][0], deltaz, [0.5])
