"""Learning-rate schedulers by the reference's names (scheduler/__init__.py:8-29, scheduler/schedulers.py).

Only what the trainer's checkpoint needs is here: every scheduler is a stock ``torch.optim.lr_scheduler``
object (so ``state_dict()`` / ``load_state_dict()`` round-trip through ``_save_checkpoints`` /
``_load_checkpoints``, trainer/trainers.py:174-195).  ``ConstantLR`` and ``PolynomialLR`` are the two the
reference defines itself; they are expressed as ``LambdaLR`` factors with the reference's formula
(scheduler/schedulers.py:8-31: the polynomial factor applies only on epochs that are multiples of both
``decay_iter`` and ``max_iter``, every other epoch keeps the base rate).
"""
from __future__ import annotations

import functools
import warnings

from torch.optim import lr_scheduler as _S


def _constant_factor(epoch):
    return 1.0


def _poly_factor(epoch, max_iter, decay_iter, gamma):
    if epoch % decay_iter or epoch % max_iter:
        return 1.0
    return (1 - epoch / float(max_iter)) ** gamma


def ConstantLR(optimizer, last_epoch=-1):
    return _S.LambdaLR(optimizer, _constant_factor, last_epoch=last_epoch)


def PolynomialLR(optimizer, max_iter, decay_iter=1, gamma=0.9, last_epoch=-1):
    f = functools.partial(_poly_factor, max_iter=max_iter, decay_iter=decay_iter, gamma=gamma)
    return _S.LambdaLR(optimizer, f, last_epoch=last_epoch)


key2scheduler = {
    'ConstantLR': ConstantLR,
    'PolynomialLR': PolynomialLR,
    'StepLR': _S.StepLR,
    'MultiStepLR': _S.MultiStepLR,
    'CosineAnnealingLR': _S.CosineAnnealingLR,
    'ExponentialLR': _S.ExponentialLR,
}


def get_scheduler(optimizer, scheduler_dict):
    """scheduler/__init__.py:18-29: ``None`` means a constant rate; otherwise ``{'name': ..., **params}``."""
    if scheduler_dict is None:
        print('Using no lr scheduler')
        return ConstantLR(optimizer)
    params = dict(scheduler_dict)
    s_type = params.pop('name')
    if s_type not in key2scheduler:
        raise NotImplementedError('Scheduler {} not implemented'.format(s_type))
    print('Using scheduler: {} with {} params'.format(s_type, params))
    return key2scheduler[s_type](optimizer, **params)


def tick(scheduler, epoch):
    """The reference's per-epoch tick, ``scheduler.step(epoch=epoch_idx)`` (trainer/trainers.py:209): the rate is a
    closed-form function of the epoch number, so a run resumed in epoch e (checkpoint: last_epoch = e, fresh
    optimizer -- the reference does not restore it either, trainer/trainers.py:185-195) continues with epoch e's
    rate instead of ticking one epoch ahead.  torch keeps this form of ``step`` behind a deprecation warning."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        scheduler.step(epoch)
