"""FlowPlan: Python handle of a `glowhip_plan` (include/glowhip.h) built from live nn.Modules.

A plan is the MI355X-native replacement of the reference's Python loop over ``FlowModel.layers``
(network/model.py:263-294): the layer list is static, so the whole stack (Squeeze2d / FlowStep / Split2d)
becomes ONE C call that issues a fixed kernel sequence on torch's current HIP stream.  torch only owns
the memory (parameters, workspace, packed-parameter buffer).
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import LayerDesc, check, lib, ptr, stream_ptr


def _param_tensors(layer) -> List[torch.Tensor]:
    return [p for p in layer.parameters()]


class FlowPlan:
    """One executable stack of flow layers for a fixed input (C, H, W) on one device."""

    def __init__(self, layers: Sequence[torch.nn.Module], in_chw, device: torch.device):
        self.layers = list(layers)
        self.in_chw = tuple(int(v) for v in in_chw)
        self.device = torch.device(device)
        self._keep = []  # tensors whose addresses the C plan holds
        descs = (LayerDesc * len(self.layers))()
        c, h, w = self.in_chw
        for i, layer in enumerate(self.layers):
            d = descs[i]
            d.C, d.H, d.W = c, h, w
            kind = layer.glowhip_kind
            d.kind = kind
            if kind == _lib.LAYER_SQUEEZE:
                if layer.factor != 2:
                    raise _lib.GlowHipError("flow plans support Squeeze2d(factor=2) only")
                c, h, w = c * 4, h // 2, w // 2
            elif kind == _lib.LAYER_FLOWSTEP:
                self._fill_flowstep(d, layer)
            else:
                self._fill_split(d, layer)
                c = c // 2
        self.out_chw = (c, h, w)
        self._params = [p for layer in self.layers for p in _param_tensors(layer)]
        for p in self._params:
            if p.device != self.device:
                raise _lib.GlowHipError(f"parameter on {p.device}, plan on {self.device}")
        self._ptr_sig = self._pointer_signature()
        handle = lib().glowhip_plan_create(descs, len(self.layers))
        if not handle:
            raise _lib.GlowHipError("glowhip_plan_create: " + lib().glowhip_last_error().decode())
        self._h = ctypes.c_void_p(handle)
        self._descs = descs
        self.packed_bytes = int(lib().glowhip_plan_packed_bytes(self._h))
        self.packed = torch.empty(max(self.packed_bytes, 256), dtype=torch.uint8, device=self.device)
        self._packed_version = None
        self._ws: Optional[torch.Tensor] = None
        self.n_split = sum(1 for l in self.layers if l.glowhip_kind == _lib.LAYER_SPLIT2D)

    # ------------------------------------------------------------------ construction helpers
    def _dev(self, t: torch.Tensor) -> int:
        if not t.is_cuda:
            raise _lib.GlowHipError("flow plans need parameters on a HIP device (module.cuda() first)")
        if not t.is_contiguous() or t.dtype != torch.float32:
            raise _lib.GlowHipError("parameters must be contiguous fp32")
        self._keep.append(t)
        return t.data_ptr()

    def _fill_flowstep(self, d: LayerDesc, step) -> None:
        d.hidden = step.hidden_channels
        d.coupling = _lib.COUPLING_AFFINE if step.coupling == 'affine' else _lib.COUPLING_ADDITIVE
        d.an_bias, d.an_logs = self._dev(step.actnorm.bias), self._dev(step.actnorm.logs)
        if step.permutation == 'invconv':
            d.permutation = _lib.PERM_INVCONV
            d.invconv_w = self._dev(step.invconv.weight)
        else:
            d.permutation = _lib.PERM_GATHER
            perm = getattr(step, step.permutation)
            idx, inv = perm.device_tables(self.device)
            self._keep += [idx, inv]
            d.perm_idx, d.perm_idx_inv = idx.data_ptr(), inv.data_ptr()
        f0, f2, f4 = step.f[0], step.f[2], step.f[4]
        d.f0_w, d.f0_an_bias, d.f0_an_logs = self._dev(f0.weight), self._dev(f0.actnorm.bias), self._dev(f0.actnorm.logs)
        d.f2_w, d.f2_an_bias, d.f2_an_logs = self._dev(f2.weight), self._dev(f2.actnorm.bias), self._dev(f2.actnorm.logs)
        d.f4_w, d.f4_bias, d.f4_logs = self._dev(f4.weight), self._dev(f4.bias), self._dev(f4.logs)

    def _fill_split(self, d: LayerDesc, split) -> None:
        cz = split.conv2d_zeros
        d.f4_w, d.f4_bias, d.f4_logs = self._dev(cz.weight), self._dev(cz.bias), self._dev(cz.logs)

    def _pointer_signature(self):
        return tuple(p.data_ptr() for p in self._params)

    def _version_signature(self):
        return tuple(p._version for p in self._params)

    def still_valid(self) -> bool:
        """False when a parameter was re-allocated (module moved / parameter replaced): rebuild the plan."""
        return self._pointer_signature() == self._ptr_sig

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().glowhip_plan_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------ execution
    def _workspace(self, n: int) -> torch.Tensor:
        need = int(lib().glowhip_plan_workspace_bytes(self._h, n))
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    PACK_INFERENCE, PACK_TRAINING, PACK_INVERSE = 1, 2, 4   # glowhip.h GLOWHIP_PACK_*
    FAMILY_AUTO, FAMILY_EXACT_FP32 = 0, 1                   # glowhip.h GLOWHIP_FAMILY_*

    @property
    def family(self) -> int:
        return int(lib().glowhip_plan_get_family(self._h))

    def set_family(self, family: int) -> None:
        """Kernel family of this plan's coupling networks (glowhip_plan_set_family): per plan, no process-wide state.  The
        exact-fp32 family reads the fp32 MFMA weight images; they are packed on demand by the next call (ensure_packed)."""
        check(lib().glowhip_plan_set_family(self._h, int(family)))

    def _family_use(self, use: int) -> int:
        return use | (self.PACK_TRAINING if self.family == self.FAMILY_EXACT_FP32 else 0)

    def status(self, n: int, result: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(n,) int32 device tensor: sticky non-finite flags of the call that last used this plan's workspace (bits 0-2) |
        8 where `result` (that call's output tensor) holds a non-finite element.  Asynchronous: no host sync."""
        st = torch.empty(n, dtype=torch.int32, device=self.device)
        if n:
            ws = self._workspace(n)
            elems = 0 if result is None else result[0].numel()
            check(lib().glowhip_plan_status(self._h, ptr(ws), ws.numel(), n, ptr(result), elems, ptr(st), stream_ptr(self.device)))
        return st

    def pack(self, use: int = 7, merge: bool = True) -> None:
        """Refresh what is derived from the parameters (exp(3 logs), LU, and the weight images `use` asks for: the inference
        kernels' and/or the training kernels')."""
        version = self._version_signature()
        if merge and version == self._packed_version:
            use |= getattr(self, "_packed_use", 0)       # same parameters: the images packed earlier stay valid, add to them
        check(lib().glowhip_plan_pack_for(self._h, ptr(self.packed), self.packed.numel(), use, stream_ptr(self.device)))
        self._packed_version = version
        self._packed_use = use
        # (a pack for the OTHER kernel family rebuilds the plan's host-side job tables with other contents and sizes -- and the copy
        # nodes of a CAPTURED pack read them through the addresses of capture time: a replay of the training step's graph after a
        # re-run on the exact-fp32 family faulted on a host address.  Whoever replays a graph with this plan's pack inside compares
        # this count -- training.GraphedTrainStep, GraphedForward.  Packs of the same family, whatever their image set, keep their
        # tables in place: tests/test_gpu_fused.py test_captured_forward_survives_packs_of_other_use_masks.)
        key = self.family
        if key != getattr(self, "_pack_key", None):
            self._pack_key = key
            self._pack_epoch = getattr(self, "_pack_epoch", 0) + 1

    def ensure_packed(self, force: bool = False, use: int = 1) -> None:
        """Re-derive the packed data when the parameters changed since the last pack (or `force`), or when images `use` asks
        for have not been packed for the current parameter version."""
        use = self._family_use(use)
        have = getattr(self, "_packed_use", 0)
        stale = self._packed_version != self._version_signature()
        if force or stale or (use & ~have):
            # forced = the caller declares the parameters changed (behind the version counters' back): nothing packed earlier counts
            self.pack(use if (force or stale) else (use & ~have), merge=not force)

    def pack_sync(self) -> None:
        """Host-wait for the side-stream part of the last pack (glowhip_plan_pack_sync): before capturing pack-free calls."""
        check(lib().glowhip_plan_pack_sync(self._h))

    def invalidate(self) -> None:
        self._packed_version = None

    def timing(self, enable: bool) -> None:
        """Record HIP events around every coupling-path kernel launch of subsequent encode/decode calls."""
        check(lib().glowhip_plan_timing_enable(self._h, int(enable)))

    def timing_read(self, max_records: int = 1 << 16):
        """[(kind, layer, mfma, ms)] in launch order since the last read (synchronises with the events)."""
        buf = (_lib.TimingRecord * max_records)()
        n = ctypes.c_int(0)
        check(lib().glowhip_plan_timing_read(self._h, buf, max_records, ctypes.byref(n)))
        return [(buf[i].kind, buf[i].layer, buf[i].mfma, buf[i].ms) for i in range(n.value)]

    def set_dequant_rng(self, seed, enable: bool = True) -> int:
        """In-kernel dequantisation noise for glow_forward calls without a noise tensor; returns the next call number."""
        nxt = ctypes.c_ulonglong(0)
        check(lib().glowhip_plan_set_dequant_rng(self._h, int(seed) & (2 ** 64 - 1), -1 if enable is None else int(bool(enable)), ctypes.byref(nxt)))
        return int(nxt.value)

    def set_dequant_stream(self, seed, call) -> None:
        """In-kernel dequantisation noise on, the next forward without a noise tensor draws with (seed, call)."""
        check(lib().glowhip_plan_set_dequant_stream(self._h, int(seed) & (2 ** 64 - 1), int(call)))

    def dequant_noise(self, shape, seed, call, n_bits) -> torch.Tensor:
        """The draw the kernel makes for call number `call` under `seed`, as a tensor shaped like x."""
        out = torch.empty(tuple(shape), dtype=torch.float32, device=self.device)
        check(lib().glowhip_dequant_noise(ptr(out), out.numel(), int(seed) & (2 ** 64 - 1), int(call), int(n_bits), stream_ptr(self.device)))
        return out

    def describe(self, n: int = 0) -> str:
        """Kernel selection per layer; with a batch size ``n`` the choices that depend on the grid size are resolved as a
        call with that batch resolves them (n = 0: what the shapes support)."""
        buf = ctypes.create_string_buffer(1 << 16)
        check(lib().glowhip_plan_describe_for(self._h, int(n), buf, len(buf)))
        return buf.value.decode()

    def launch_counts(self, reset: bool = False) -> dict:
        """{kernel family: launches} recorded by the C executor as it launched them (run-time evidence, not a prediction)."""
        buf = ctypes.create_string_buffer(1 << 14)
        check(lib().glowhip_plan_launch_counts(self._h, buf, len(buf), int(reset)))
        return {k: int(v) for k, v in (line.split("=") for line in buf.value.decode().splitlines() if line)}

    def encode(self, x, noise=None, logdet=None, want_logdet=True, repack=False):
        n = x.shape[0]
        assert tuple(x.shape[1:]) == self.in_chw, (x.shape, self.in_chw)
        self.ensure_packed(repack)
        z = torch.empty((n,) + self.out_chw, dtype=torch.float32, device=self.device)
        ld_out = torch.empty(n, dtype=torch.float32, device=self.device) if want_logdet else None
        if n == 0:
            return z, ld_out
        ws = self._workspace(n)
        check(lib().glowhip_plan_encode(self._h, ptr(self.packed), ptr(x), ptr(noise), ptr(logdet), ptr(z), ptr(ld_out), n,
                                        ptr(ws), ws.numel(), stream_ptr(self.device)))
        return z, ld_out

    def decode(self, z, eps: Sequence[torch.Tensor], logdet=None, want_logdet=False, repack=False):
        n = z.shape[0]
        assert tuple(z.shape[1:]) == self.out_chw, (z.shape, self.out_chw)
        assert len(eps) >= self.n_split
        self.ensure_packed(repack, use=self.PACK_INFERENCE | self.PACK_INVERSE)
        x = torch.empty((n,) + self.in_chw, dtype=torch.float32, device=self.device)
        ld_out = torch.empty(n, dtype=torch.float32, device=self.device) if want_logdet else None
        if n == 0:
            return x, ld_out
        ws = self._workspace(n)
        arr = (ctypes.c_void_p * max(len(eps), 1))(*[e.data_ptr() for e in eps])
        check(lib().glowhip_plan_decode(self._h, ptr(self.packed), ptr(z), arr, len(eps), ptr(logdet), ptr(x), ptr(ld_out),
                                        n, ptr(ws), ws.numel(), stream_ptr(self.device)))
        return x, ld_out

    def glow_forward(self, x, noise, prior_mean, prior_logs, prior_stride, n_bits, repack=False, out=None):
        n = x.shape[0]
        self.ensure_packed(repack)
        if out is None:
            z = torch.empty((n,) + self.out_chw, dtype=torch.float32, device=self.device)
            nll = torch.empty(n, dtype=torch.float32, device=self.device)
            obj = torch.empty(n, dtype=torch.float32, device=self.device)
        else:
            z, nll, obj = out
        if n == 0:
            return z, nll, obj
        ws = self._workspace(n)
        if x.dtype == torch.uint8:   # 8-bit pixels straight from the data loader: converted inside the leading squeeze
            check(lib().glowhip_glow_forward_u8(self._h, ptr(self.packed), ptr(x), 255.0, ptr(noise), ptr(prior_mean),
                                                ptr(prior_logs), prior_stride, n_bits, ptr(z), ptr(nll), ptr(obj), n, ptr(ws),
                                                ws.numel(), stream_ptr(self.device)))
            return z, nll, obj
        check(lib().glowhip_glow_forward(self._h, ptr(self.packed), ptr(x), ptr(noise), ptr(prior_mean), ptr(prior_logs),
                                         prior_stride, n_bits, ptr(z), ptr(nll), ptr(obj), n, ptr(ws), ws.numel(),
                                         stream_ptr(self.device)))
        return z, nll, obj

    # ------------------------------------------------------------------ training step
    def _grad_fields(self):
        """[(layer index, LayerGrads field, parameter)] for every parameter the C backward produces a gradient for (cached: the
        plan is rebuilt when a parameter is replaced, PlanCache.get / still_valid)."""
        cached = getattr(self, "_fields", None)
        if cached is not None:
            return cached
        out = []
        for i, layer in enumerate(self.layers):
            kind = layer.glowhip_kind
            if kind == _lib.LAYER_FLOWSTEP:
                out += [(i, "an_bias", layer.actnorm.bias), (i, "an_logs", layer.actnorm.logs)]
                if layer.permutation == 'invconv':
                    out.append((i, "invconv_w", layer.invconv.weight))
                f0, f2, f4 = layer.f[0], layer.f[2], layer.f[4]
                out += [(i, "f0_w", f0.weight), (i, "f0_an_bias", f0.actnorm.bias), (i, "f0_an_logs", f0.actnorm.logs),
                        (i, "f2_w", f2.weight), (i, "f2_an_bias", f2.actnorm.bias), (i, "f2_an_logs", f2.actnorm.logs),
                        (i, "f4_w", f4.weight), (i, "f4_bias", f4.bias), (i, "f4_logs", f4.logs)]
            elif kind == _lib.LAYER_SPLIT2D:
                cz = layer.conv2d_zeros
                out += [(i, "f4_w", cz.weight), (i, "f4_bias", cz.bias), (i, "f4_logs", cz.logs)]
        self._fields = out
        self._trainable = [p for _, _, p in out]
        return out

    def trainable_parameters(self):
        self._grad_fields()
        return self._trainable

    def glow_forward_train(self, x, noise, prior_mean, prior_logs, prior_stride, n_bits, force_pack=False):
        """Forward that records the activation tape; returns (z, nll, tape).  ``force_pack``: re-derive the weight images whatever the
        version counters say (a captured training step: every replay follows an update the counters of capture time know nothing of)."""
        n = x.shape[0]
        self.ensure_packed(force_pack, use=self.PACK_TRAINING)   # re-packs when the weights changed since the last pack (optimizer step)
        z = torch.empty((n,) + self.out_chw, dtype=torch.float32, device=self.device)
        nll = torch.empty(n, dtype=torch.float32, device=self.device)
        tape = torch.empty(int(lib().glowhip_plan_tape_bytes(self._h, n)), dtype=torch.uint8, device=self.device)
        ws = self._train_workspace(n)
        check(lib().glowhip_glow_forward_train(self._h, ptr(self.packed), ptr(x), ptr(noise), ptr(prior_mean),
                                               ptr(prior_logs), prior_stride, n_bits, ptr(z), ptr(nll), None, n, ptr(tape),
                                               tape.numel(), ptr(ws), ws.numel(), stream_ptr(self.device)))
        self._tape_version = self._packed_version           # glow_backward must see the same weights (and their images)
        return z, nll, tape

    def _train_workspace(self, n):
        need = int(lib().glowhip_plan_train_workspace_bytes(self._h, n))
        ws = getattr(self, "_tws", None)
        if ws is None or ws.numel() < need:
            self._tws = None
            self._tws = ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return ws

    def glow_backward(self, x, tape, nll_grad, z_grad, prior_mean, prior_logs, prior_stride, want_grad_x=False, persistent=False):
        """Parameter gradients (list aligned with trainable_parameters()) and optionally dL/dx.
        ``persistent``: the flat gradient buckets, the views into them, the pointer table handed to the C sweep and the
        gradient-ready events are created ONCE per plan and reused by every call (the sweep writes every gradient in full) -- the
        per-step host cost of ~1 000 tensor views and ~1 000 ctypes assignments disappears; the caller must have consumed the
        previous call's gradients (training.TrainLoop: the optimiser step of step N is enqueued before the sweep of step N + 1)."""
        n = x.shape[0]
        if (self._version_signature() != getattr(self, "_tape_version", None) or self._packed_version != self._tape_version
                or not (getattr(self, "_packed_use", 0) & self.PACK_TRAINING)):
            raise _lib.GlowHipError("glow_backward: the parameters (or their packed images) changed between the training forward "
                                    "and its backward -- an in-place update or optimizer step in between would mix weight versions")
        fields = self._grad_fields()
        layout = self._bucket_layout()
        # Gradients live in one flat buffer per bucket (the convolution weights of one LEVEL; everything small in a last one):
        # a data-parallel run all-reduces each bucket in place -- no concatenation, no copy back -- and starts with a level's
        # bucket as soon as the sweep has left that level (gradient-ready marks), while the lower levels are still being swept.
        cached = getattr(self, "_pgrad", None) if persistent else None
        if cached is None:
            flats = [torch.empty(nel, dtype=torch.float32, device=self.device) for nel in layout["sizes"]]
            grads = [flats[b][off:off + p.numel()].view_as(p) for (_, _, p), (b, off) in zip(fields, layout["slots"])]
            arr = (_lib.LayerGrads * len(self.layers))()
            for (i, name, _), gt in zip(fields, grads):
                setattr(arr[i], name, gt.data_ptr())
            events = [torch.cuda.Event() for _ in layout["marks"]]
            for ev in events:
                ev.record()                  # (creates the handle; the C sweep records it again where it belongs)
            marks = (ctypes.c_int32 * max(len(events), 1))(*layout["marks"])
            handles = (ctypes.c_void_p * max(len(events), 1))(*[ev.cuda_event for ev in events])
            if persistent:
                self._pgrad = (flats, grads, arr, events, marks, handles)
        else:
            flats, grads, arr, events, marks, handles = cached
        if not persistent:
            # the sweep rewrites the plan's host-side gradient job tables with THIS call's temporary buffers, and the copy nodes of a
            # captured training step read those tables at replay time: training.GraphedTrainStep sees this epoch move and captures
            # again after an eager step.  (A captured inference forward reads none of them: it keeps to _pack_epoch.)
            self._grad_table_epoch = getattr(self, "_grad_table_epoch", 0) + 1
        gx = torch.empty_like(x) if want_grad_x else None
        ws = self._train_workspace(n)
        check(lib().glowhip_plan_backward_marks(self._h, marks, handles, len(events)))
        try:
            check(lib().glowhip_glow_backward(self._h, ptr(self.packed), ptr(x), ptr(tape), tape.numel(), ptr(nll_grad),
                                              ptr(z_grad), ptr(prior_mean), ptr(prior_logs), prior_stride, arr, ptr(gx), n,
                                              ptr(ws), ws.numel(), stream_ptr(self.device)))
        finally:
            check(lib().glowhip_plan_backward_marks(self._h, None, None, 0))
        done = torch.cuda.Event()
        done.record()
        # in sweep order: (flat bucket, event after which it is final); the small bucket is final at the end of the call
        self.last_grad_buckets = [(flats[b], events[k]) for k, b in enumerate(layout["mark_bucket"])] + [(flats[-1], done)]
        return grads, gx

    def _bucket_layout(self):
        lay = getattr(self, "_bucket_lay", None)
        if lay is not None:
            return lay
        fields = self._grad_fields()
        level_of, starts, cur = [], [], -1
        for i, layer in enumerate(self.layers):
            if layer.glowhip_kind == _lib.LAYER_SQUEEZE or cur < 0:
                cur += 1
                starts.append(i)
            level_of.append(cur)
        nlev = cur + 1
        sizes = [0] * (nlev + 1)                                   # one bucket per level + the small one (last)
        slots = []
        for i, name, p in fields:
            b = level_of[i] if name in ("f0_w", "f2_w", "f4_w") else nlev
            slots.append((b, sizes[b]))
            sizes[b] += (p.numel() + 63) // 64 * 64                # 256-byte aligned slots
        order = [l for l in range(nlev - 1, -1, -1) if sizes[l] > 0]   # sweep order: deepest level first
        lay = dict(sizes=[max(v, 64) for v in sizes], slots=slots, marks=[starts[l] for l in order], mark_bucket=order)
        self._bucket_lay = lay
        return lay

    def actnorm_init(self, x, noise, actnorm_scale: float) -> None:
        """Data-dependent init of every ActNorm in the plan from batch x (writes the parameters in place)."""
        n = x.shape[0]
        ws = self._workspace(n)
        check(lib().glowhip_plan_actnorm_init(self._h, ptr(self.packed), self.packed.numel(), ptr(x), ptr(noise),
                                              float(actnorm_scale), n, ptr(ws), ws.numel(), stream_ptr(self.device)))
        self._packed_version = self._version_signature()   # the init pass ends with a pack of the inference kernels' data
        self._packed_use = self.PACK_INFERENCE


class PlanCache:
    """Per-module cache of FlowPlans keyed by (input CHW, device); rebuilt when parameters move."""

    def __init__(self):
        self._plans = {}

    def get(self, layers, in_chw, device) -> FlowPlan:
        key = (tuple(in_chw), str(device))
        plan = self._plans.get(key)
        if plan is None or not plan.still_valid():
            plan = FlowPlan(layers, in_chw, device)
            self._plans[key] = plan
        return plan

    def clear(self):
        self._plans.clear()

    def invalidate(self):
        for p in self._plans.values():
            p.invalidate()
