"""`Builder` of the reference (network/builder.py:9-113): devices, `Glow(hps)`, warm start from a snapshot, optimiser and
learning-rate schedule -- the glue that lets the reference's `train.py` / `infer.py` construct this package's model.
One process drives ONE GPU here (`hps.device.graph` lists the device of this rank); DataParallel is replaced by
`pytorch_glow_amd.parallel` / `training.TrainLoop`."""
import os

from . import model
from .. import training as _training
from ..misc import lr_scheduler, util


def _snapshot_to_resume(general):
    """What `hps.general` asks to warm-start from: an existing file path, a step number, or None (builder.py:66-71)."""
    if not general.warm_start:
        return None
    if os.path.exists(general.pre_trained):
        return general.pre_trained
    if general.resume_step not in ('', 'best', 'latest'):
        return int(general.resume_step)
    return None


class Builder:
    optimizer_dict = _training.OPTIMIZERS              # builder.py:10-13
    lr_scheduler_dict = lr_scheduler.SCHEDULES         # builder.py:14-20

    def __init__(self, hps):
        self.hps = hps

    def _result_subdir(self, training):
        g = self.hps.general
        found = util.locate_result_subdir(g.result_dir, g.resume_run_id) if g.warm_start and g.resume_run_id != "" else None
        if found is None and training:
            found = util.create_result_subdir(g.result_dir, desc=self.hps.profile, profile=self.hps)
        return found

    def build(self, training=True):
        """The reference's result dict: step, graph, optimizer, scheduler, devices, data_device, result_subdir."""
        hps = self.hps
        devices = util.get_devices(hps.device.graph)
        data_device = util.get_devices(hps.device.data)[0]
        if 'cpu' in devices:   # the reference would run its CPU graph here; this package has no CPU flow path
            raise RuntimeError("the flow path needs a HIP device; hps.device.graph = %r has none usable" % (hps.device.graph,))
        out = dict(step=0, graph=model.Glow(hps), optimizer=None, scheduler=None, devices=devices, data_device=data_device,
                   result_subdir=self._result_subdir(training))

        snapshot, state = _snapshot_to_resume(hps.general), None
        if snapshot is not None:
            state = util.load_model(out['result_subdir'], snapshot, out['graph'], device='cpu')
            out['step'] = state['step']
        if hps.general.warm_start and state is None and not training:
            raise RuntimeError('No pre-trained model for inference')

        out['graph'] = out['graph'].to('cuda:{}'.format(devices[0]))
        print('[Builder] Use {} for model running and {} for data loading'.format(devices[0], data_device))
        if training:   # optimiser AFTER the move to the device, Adam state of the snapshot restored (builder.py:86-104)
            out['optimizer'] = _training.build_optimizer(hps, out['graph'].parameters())
            if state is not None:
                out['optimizer'].load_state_dict(state['optimizer'])
            out['scheduler'] = _training.build_scheduler(hps)
        return out
