"""Train-time callbacks of the thin trainer, with the class surface of the reference's callbacks.py
(tf2.5/scripts/callbacks.py): ``WeightsSaver`` (CB:44-75), ``ReduceLR_Schedule`` (CB:79-101), ``PolyLR_Schedule``
(CB:105-119) and ``ResumeTraining`` (CB:195-215).  Keras-style hooks (``on_epoch_begin/on_epoch_end``) driven by
``LoadableModel.fit``.

Checkpoints are ``model_weights_NNN.npz`` (Keras tensor layouts under App. E names + the constructor config; h5py is not
in the image) instead of ``model_weights_NNN.h5``.  The reference's harness bugs are not reproduced:
``ResumeTraining`` overwrites its own ``weights_dir`` argument (CB:196) -- here it scans the directory it was given;
the FROC validation callbacks it imports do not exist in the reference and are out of scope.
"""
from __future__ import annotations

import os
import re
from typing import Optional, Sequence, Tuple

_PREFIX = "model_weights"
_EXT = ".npz"


def weights_path(weights_dir: str, epoch: int, prefix: str = _PREFIX) -> str:
    """``<dir>/model_weights_NNN.npz`` (CB:55-56: ``'_%03d'``)."""
    return os.path.join(weights_dir, prefix + "_%03d" % epoch + _EXT)


class Callback:
    model = None

    def set_model(self, model):
        self.model = model

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass


class WeightsSaver(Callback):
    """Export weights every N epochs (CB:44-75): at the end of an epoch, with e = own epoch counter (starts at
    ``init_epoch``), save when ``(e+1) % N == 0 and e != 0 and (e+1) >= min_epoch`` to ``model_weights_%03d`` of e+1;
    with ``weights_overwrite`` the file of N epochs earlier is removed afterwards."""

    def __init__(self, model, min_epoch, weights_num_epochs, weights_dir, init_epoch=0, weights_overwrite=True, rank: int = 0):
        self.model = model
        self.N = int(weights_num_epochs)
        self.M = int(min_epoch)
        self.D = weights_dir
        self.O = bool(weights_overwrite)
        self.epoch = int(init_epoch)
        self.rank = int(rank)                  # data-parallel runs: replicas are identical, rank 0 writes
        self.saved = []

    def on_epoch_end(self, epoch, logs=None):
        if ((self.epoch + 1) % self.N == 0) and (self.epoch != 0) and ((self.epoch + 1) >= self.M):
            if self.rank == 0:
                name = weights_path(self.D, self.epoch + 1)
                os.makedirs(self.D, exist_ok=True)
                tmp = name + ".tmp" + _EXT
                self.model.save(tmp)                      # complete file first, then an atomic rename: a reader (resume)
                os.replace(tmp, name)                     # never sees a torn checkpoint
                self.saved.append(name)
                print('Model Weights Saved: ', name, flush=True)
                if self.O:
                    old = weights_path(self.D, (self.epoch + 1) - self.N)
                    if os.path.exists(old):
                        os.remove(old)
        self.epoch += 1


class ReduceLR_Schedule(Callback):
    """Piecewise-constant learning rate at four epoch points (CB:79-101)."""

    def __init__(self, lr_rates: Sequence[float], epoch_points: Sequence[int]):
        self.lr_rates = list(lr_rates)
        self.epoch_points = list(epoch_points)

    def on_epoch_begin(self, epoch, logs=None):
        assert len(self.epoch_points) == len(self.lr_rates)
        e, pts = epoch + 1, self.epoch_points
        if e in pts:
            new_lr = self.lr_rates[max(i for i, p in enumerate(pts) if e >= p)]
            self.model.optimizer.lr = new_lr
            print('\nEpoch %03d: ReduceLR_Schedule reducing learning rate to %s.' % (e, new_lr), flush=True)


class PolyLR_Schedule(Callback):
    """nn-U-Net polynomial decay (CB:105-119): lr = initial * (1 - epoch/max_epochs)**exponent at every epoch begin."""

    def __init__(self, initial_lr, exponent, max_epochs):
        self.initial_lr, self.exponent, self.max_epochs = float(initial_lr), float(exponent), int(max_epochs)

    def on_epoch_begin(self, epoch, logs=None):
        new_lr = self.initial_lr * (1 - epoch / self.max_epochs) ** self.exponent
        self.model.optimizer.lr = new_lr
        print('\nEpoch %03d: PolyLR_Schedule reducing learning rate to %s.' % (epoch + 1, new_lr), flush=True)


def latest_checkpoint(weights_dir: str, prefix: str = _PREFIX) -> Tuple[Optional[str], int]:
    """(path, epoch) of the highest-numbered ``<prefix>_NNN.npz`` in ``weights_dir`` (CB:199-203), or (None, 0)."""
    best, best_path = 0, None
    if os.path.isdir(weights_dir):
        pat = re.compile(re.escape(prefix) + r"_(\d+)" + re.escape(_EXT) + r"$")
        for f in os.listdir(weights_dir):
            m = pat.match(f)
            if m and int(m.group(1)) > best:
                best, best_path = int(m.group(1)), os.path.join(weights_dir, f)
    return best_path, best


def ResumeTraining(model, weights_dir, resume=True, prefix=_PREFIX):
    """Load the newest checkpoint of ``weights_dir`` and return ``(model, init_epoch)`` (CB:195-215).  The model is
    re-created from the constructor config stored with the weights (``M1.load``, modelio.py:97-117), like the reference;
    without a checkpoint the given model is returned with ``init_epoch = 0``."""
    init_epoch = 0
    if resume:
        path, init_epoch = latest_checkpoint(weights_dir, prefix)
        if path is not None:
            print('Loading Model Weights...', flush=True)
            dev = next(model.parameters()).device
            dtype = getattr(model, "compute_dtype", None)
            model = type(model).load(path=path).to(dev)
            if dtype is not None and hasattr(model, "set_compute_dtype"):
                model.set_compute_dtype(dtype)
            print('Complete: ', path, flush=True)
    if init_epoch == 0:
        print('Begin Training @ Epoch ', init_epoch, flush=True)
    else:
        print('Resume Training @ Epoch ', init_epoch, flush=True)
    return model, init_epoch
