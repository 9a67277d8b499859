"""`Inferer` of the reference (network/inferer.py:10-188): the inverse-path application -- sampling, encode / decode of a
single image, and attribute manipulation in latent space -- on the HIP flow path.

Differences, all deliberate:
* `compute_attribute_delta` accumulates on the device (two matrix products per batch instead of a Python loop over
  samples x classes and a device->host copy per image) and, in a multi-process run, ends with ONE all-reduce of the
  (classes, C, H, W) sums and the counts -- images are independent, so ranks just see different batches.
* The reference's accumulation loop runs `for i in range(len(batch))` where `batch` is the DICT of the data loader
  (inferer.py:131): it visits len({'x', 'y_onehot'}) = 2 samples of every batch, not the batch.  `samples_per_batch=None`
  (default) uses every sample -- what the method documents; `samples_per_batch="reference"` reproduces the loop as written
  (the parity tests check both against the oracle).
* `sample` returns the batch; building an image grid is torchvision's job (`make_grid`), outside the flow path."""
import numpy as np
import torch
import torch.distributed as dist

from ..misc import util


class Inferer:
    def __init__(self, hps, graph, devices, data_device):
        self.hps = hps
        self.graph = graph
        self.graph.eval()
        self.devices = devices
        self.data_device = data_device
        self.batch_size = self.graph.h_top.shape[0]
        self.num_classes = self.hps.dataset.num_classes
        self.y_condition = self.hps.ablation.y_condition
        self.device = self.graph.h_top.device

    def sample(self, z, y_onehot, eps_std=0.5):
        """Images drawn from the model (z=None: the top latent is sampled too), inferer.py:41-62."""
        with torch.no_grad():
            return self.graph(z=z, y_onehot=y_onehot, eps_std=eps_std, reverse=True)

    def encode(self, img):
        """Latent of ONE image (C,H,W tensor; a batch is accepted and its first latent returned), inferer.py:64-85."""
        with torch.no_grad():
            if not torch.is_tensor(img):
                raise TypeError("encode takes a tensor; image decoding (cv2 / PIL) is outside this package")
            if len(img.shape) == 3:
                img = util.make_batch(img, self.batch_size)
            z, _, _ = self.graph(img.to(self.device))
            return z[0, :, :, :]

    def decode(self, z):
        """Image of ONE latent, inferer.py:87-102."""
        with torch.no_grad():
            if len(z.shape) == 3:
                z = util.make_batch(z, self.batch_size)
            return self.graph(z=z.to(self.device), y_onehot=None, reverse=True)[0, :, :, :]

    def compute_attribute_delta(self, dataset, samples_per_batch=None, shuffle=True, num_workers=None, world=1, _exact=False):
        """deltaz[c] = mean latent of the images with attribute c - mean latent of those without (inferer.py:104-153).
        `dataset` yields dicts with 'x' (C,H,W) and 'y_onehot' (classes,).

        Range check without a host sync per batch: the encodes run unchecked on the product kernels, a device flag collects
        "some nll was not finite", and it is read once with the sums at the end -- only then (an input beyond the fp16 pairs'
        range) the pass is repeated with the plans on the exact-fp32 family."""
        from torch.utils.data import DataLoader
        shape = tuple(self.graph.flow.output_shapes[-1][1:])
        dim = int(np.prod(shape))
        pos = torch.zeros(self.num_classes, dim, dtype=torch.float64, device=self.device)
        neg = torch.zeros_like(pos)
        n_pos = torch.zeros(self.num_classes, dtype=torch.float64, device=self.device)
        n_neg = torch.zeros_like(n_pos)
        bad = torch.zeros((), dtype=torch.bool, device=self.device)
        loader = DataLoader(dataset, batch_size=self.batch_size, shuffle=shuffle, drop_last=True,
                            num_workers=self.hps.dataset.num_workers if num_workers is None else num_workers)
        with torch.no_grad():
            for batch in loader:
                assert 'y_onehot' in batch.keys(), 'Compute attribute deltaz needs "y_onehot" in batch data'
                x = batch['x'].to(self.device)
                y = batch['y_onehot'].to(self.device)
                if hasattr(self.graph.flow, "plan_for"):
                    plan = self.graph.flow.plan_for(x)
                    fam = plan.family
                    if _exact:
                        plan.set_family(plan.FAMILY_EXACT_FP32)
                    try:
                        z, nll, _ = self.graph.normal_flow(x, None, safe=False)
                    finally:
                        plan.set_family(fam)
                    bad |= ~torch.isfinite(nll).all()
                else:                                               # (a graph without flow plans: the bookkeeping tests' stand-in)
                    z, _, _ = self.graph(x)
                take = len(batch) if samples_per_batch == "reference" else (samples_per_batch or x.shape[0])
                zf = z[:take].reshape(take, dim).double()
                has = (y[:take] > 0).double()                       # (take, classes)
                pos += has.t() @ zf
                neg += (1.0 - has).t() @ zf
                n_pos += has.sum(0)
                n_neg += (1.0 - has).sum(0)
        if world > 1:                                               # ranks saw different batches: one exchange at the end
            flat = torch.cat([pos.reshape(-1), neg.reshape(-1), n_pos, n_neg, bad.double().reshape(1)])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            k = self.num_classes * dim
            pos, neg = flat[:k].view(self.num_classes, dim), flat[k:2 * k].view(self.num_classes, dim)
            n_pos, n_neg = flat[2 * k:2 * k + self.num_classes], flat[2 * k + self.num_classes:2 * k + 2 * self.num_classes]
            bad = flat[-1] > 0
        if getattr(self.graph, "range_check", True) and not _exact and bool(bad):
            return self.compute_attribute_delta(dataset, samples_per_batch, shuffle, num_workers, world, _exact=True)
        delta = pos / n_pos.clamp(min=1.0)[:, None] - neg / n_neg.clamp(min=1.0)[:, None]
        return delta.view(self.num_classes, *shape).cpu().numpy()

    def apply_attribute_delta(self, img, deltaz, interpolation):
        """decode(encode(img) + sum_c interpolation[c] * deltaz[c]), inferer.py:155-188."""
        if isinstance(deltaz, np.ndarray):
            deltaz = torch.as_tensor(deltaz, dtype=torch.float32)
        assert len(interpolation) == self.num_classes
        assert deltaz.shape == torch.Size([self.num_classes, *self.graph.flow.output_shapes[-1][1:]])
        z = self.encode(img)
        coef = torch.as_tensor(np.asarray(interpolation, dtype=np.float32), device=self.device)
        z_interpolated = z + (deltaz.to(self.device) * coef.view(-1, 1, 1, 1)).sum(0)
        return self.decode(z_interpolated)
