"""Model I/O helpers with the surface of the reference's ``modelio.py`` (tf2.5/scripts/model/unets/
modelio.py:20-135): ``store_config_args``, ``ModelConfig``, ``LoadableModel`` (+ ``ReferenceContainer``).

The reference's ``LoadableModel`` is a ``tf.keras.Model``; here it is a ``torch.nn.Module`` that offers the
Keras-Model methods the reference's callers use (train_model.py:210-215,231,253-259; callbacks.py:62,99,
209): ``.layers``, ``.compile``, ``.fit``, ``.optimizer``, ``.inputs``, ``.save_weights/.load_weights``,
``get_config/from_config/load``.  Weights are stored as ``.npz`` in Keras tensor layouts (h5py is not
available in the build image).
"""
from __future__ import annotations

import functools
import inspect
import json
import time
from typing import Any, Dict, Iterable, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .. import initializers as _init


def store_config_args(func):
    """Class-method decorator that saves every constructor argument in ``self.config`` (modelio.py:20-55).
    Uses inspect.signature (the reference's getargspec was removed in Python 3.11)."""
    sig = inspect.signature(func)
    names = [n for n in sig.parameters][1:]
    defaults = {n: p.default for n, p in sig.parameters.items() if p.default is not inspect.Parameter.empty}

    @functools.wraps(func)
    def wrapper(self, *args, **kwargs):
        retval = func(self, *args, **kwargs)
        params = dict(defaults)
        for attr, val in zip(names, args):
            params[attr] = val
        params.update(kwargs)
        self.config = ModelConfig(params)
        return retval
    return wrapper


class ModelConfig:
    """Container so that the framework does not try to track the config (modelio.py:58-65)."""

    def __init__(self, params):
        self.params = params


def _jsonable(v):
    if hasattr(v, "get_config") and not isinstance(v, type):
        return v.get_config()
    if isinstance(v, (tuple, list)):
        return [_jsonable(x) for x in v]
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    return v


def _unjson(v):
    if isinstance(v, dict) and "class_name" in v:
        return _init.deserialize(v)
    if isinstance(v, list):
        return tuple(_unjson(x) for x in v)
    return v


class _OptimizerHandle:
    """What callbacks touch: ``model.optimizer.lr`` (callbacks.py:99,117,179)."""

    def __init__(self, opt):
        self._opt = opt

    @property
    def lr(self):
        return self._opt.lr

    @lr.setter
    def lr(self, v):
        self._opt.lr = v

    def __getattr__(self, k):
        return getattr(self._opt, k)


class LoadableModel(nn.Module):
    """Base class for models that can be re-created from their own saved constructor arguments
    (modelio.py:68-117)."""

    def __init__(self, name: str = "model"):
        super().__init__()
        self.name = name
        self._compiled = None
        self.stop_training = False

    # ---- config (modelio.py:81-95) ----
    def get_config(self):
        if not hasattr(self, "config"):
            raise RuntimeError('models that inherit from LoadableModel must decorate the constructor with @store_config_args')
        return self.config.params

    @classmethod
    def from_config(cls, config, custom_objects=None):
        return cls(**config)

    # ---- weights: .npz with Keras tensor layouts and the stable names of SURVEY App. E ----
    _NAME_MAP = (("m1_model.", ""), ("m1_stage1.", "stage1."), ("m1_stage2.", "stage2."))

    @classmethod
    def _export_name(cls, k: str) -> str:
        """state_dict key -> App. E name: ``{prior|posterior|core}.{layer}.{sub}.{kernel|bias|gamma|beta}``,
        ``stitch.logits.*``; cascaded models prefix ``stage1.`` / ``stage2.``."""
        for a, b in cls._NAME_MAP:
            if k.startswith(a):
                return b + k[len(a):]
        return k

    def get_weights_dict(self) -> Dict[str, np.ndarray]:
        """Every weight tensor in its Keras layout (Conv3D kernel (kd,kh,kw,Cin,Cout), Conv3DTranspose kernel
        (kd,kh,kw,Cout,Cin), bias (Cout,), InstanceNormalization gamma/beta (C,)) under its App. E name."""
        return {self._export_name(k): v.detach().float().cpu().numpy() for k, v in self.state_dict().items()
                if k != "rng_state"}

    def save_weights(self, path: str):
        meta = json.dumps({"class_name": type(self).__name__, "config": {k: _jsonable(v) for k, v in self.get_config().items()}})
        np.savez(path, __model_config__=np.frombuffer(meta.encode("utf-8"), dtype=np.uint8), **self.get_weights_dict())

    def save(self, path: str):
        """tf.keras.models.save_model(model, path) equivalent used by WeightsSaver (callbacks.py:62)."""
        self.save_weights(path)

    def load_weights(self, path: str, by_name: bool = False):
        with np.load(path) as f:
            sd = {self._export_name(k): torch.from_numpy(np.array(f[k])) for k in f.files if k != "__model_config__"}
        own = {self._export_name(k): v for k, v in self.state_dict().items() if k != "rng_state"}
        if not by_name:
            missing = sorted(set(own) - set(sd)); extra = sorted(set(sd) - set(own))
            if missing or extra:
                raise RuntimeError(f"weight file does not match model: missing={missing[:5]} unexpected={extra[:5]}")
        with torch.no_grad():
            for k, v in sd.items():
                if k in own:
                    if tuple(v.shape) != tuple(own[k].shape):
                        raise RuntimeError(f"weight file does not match model: {k} has shape {tuple(v.shape)}, "
                                           f"the layer expects {tuple(own[k].shape)} (Keras layouts, SURVEY App. E)")
                    own[k].copy_(v.to(own[k].device, own[k].dtype))
        from ..hip import ops
        ops.invalidate_panels()

    @classmethod
    def load(cls, path, by_name=False):
        """Re-instantiate from the stored constructor config, then load the weights (modelio.py:97-117)."""
        with np.load(path) as f:
            meta = json.loads(bytes(np.array(f["__model_config__"])).decode("utf-8"))
        config = {k: _unjson(v) for k, v in meta["config"].items()}
        model = cls(**config)
        model.load_weights(path, by_name=by_name)
        return model

    class ReferenceContainer:
        """Holds pointers to tensors/sub-graphs without registering them again (modelio.py:119-135)."""

        def __init__(self):
            pass

    # ---- Keras-Model surface used by the reference's callers ----
    @property
    def layers(self) -> List[nn.Module]:
        """Leaf layers in construction order (train_model.py:210-215 freezes ``layers[:k]``)."""
        return [m for m in self.modules() if len(list(m.children())) == 0 and len(list(m.parameters(recurse=False))) > 0]

    @property
    def optimizer(self):
        return None if self._compiled is None else _OptimizerHandle(self._compiled["optimizer"])

    def compile(self, optimizer=None, loss=None, loss_weights=None, **_):
        """model.compile(optimizer, loss=[...], loss_weights=[...]) (train_model.py:231)."""
        losses = list(loss) if isinstance(loss, (list, tuple)) else [loss]
        weights = list(loss_weights) if loss_weights is not None else [1.0] * len(losses)
        if hasattr(optimizer, "bind"):
            optimizer.bind(self)
        self._compiled = {"optimizer": optimizer, "losses": losses, "weights": weights}

    def train_step(self, x, y) -> Dict[str, float]:
        raise NotImplementedError

    def fit(self, x: Iterable = None, epochs: int = 1, steps_per_epoch: Optional[int] = None, initial_epoch: int = 0,
            verbose: int = 2, callbacks: Optional[list] = None, use_multiprocessing: bool = False, **_):
        """model.fit(x=dataset, epochs, steps_per_epoch, initial_epoch, verbose, callbacks) (train_model.py:253-259).
        ``x`` yields (inputs, targets) batches; callbacks get Keras-style hooks."""
        if self._compiled is None:
            raise RuntimeError("call compile() before fit()")
        callbacks = callbacks or []
        history: Dict[str, List[float]] = {}
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
            elif not hasattr(cb, "model") or cb.model is None:
                cb.model = self
        it = iter(x)
        for epoch in range(initial_epoch, epochs):
            for cb in callbacks:
                getattr(cb, "on_epoch_begin", lambda *a, **k: None)(epoch, {})
            t0, agg, n = time.time(), {}, 0
            while steps_per_epoch is None or n < steps_per_epoch:
                try:
                    bx, by = next(it)
                except StopIteration:
                    if steps_per_epoch is None:
                        break
                    it = iter(x)
                    bx, by = next(it)
                logs = self.train_step(bx, by)
                for k, v in logs.items():
                    agg[k] = agg.get(k, 0.0) + float(v)
                n += 1
            logs = {k: v / max(n, 1) for k, v in agg.items()}
            for k, v in logs.items():
                history.setdefault(k, []).append(v)
            if verbose:
                print(f"Epoch {epoch + 1}/{epochs} - {time.time() - t0:.1f}s - " + " - ".join(f"{k}: {v:.6f}" for k, v in logs.items()), flush=True)
            for cb in callbacks:
                getattr(cb, "on_epoch_end", lambda *a, **k: None)(epoch, logs)
            if steps_per_epoch is None:
                it = iter(x)
            if self.stop_training:
                break

        class History:
            pass
        h = History(); h.history = history
        return h
