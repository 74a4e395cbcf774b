"""Value objects standing in for the tf.keras initializer / regularizer arguments of M1 (reference
networks.py:45-48; semantics: SURVEY.md App. B-7).  They are plain Python so that ``get_config`` stays
serialisable."""
from __future__ import annotations

import math

import torch


class Initializer:
    def __call__(self, shape, generator: torch.Generator) -> torch.Tensor:  # pragma: no cover - interface
        raise NotImplementedError

    def get_config(self):
        return {"class_name": type(self).__name__, "config": dict(self.__dict__)}


class Orthogonal(Initializer):
    """tf.keras.initializers.Orthogonal(gain): flatten to (prod(shape[:-1]), shape[-1]); QR of a N(0,1) matrix
    of shape (max,min); q*sign(diag r); transpose if rows<cols; reshape; *gain."""

    def __init__(self, gain: float = 1.0, seed=None):
        self.gain, self.seed = float(gain), seed

    def __call__(self, shape, generator):
        rows = int(math.prod(shape[:-1])); cols = int(shape[-1])
        a = torch.randn((max(rows, cols), min(rows, cols)), generator=generator, dtype=torch.float64)
        q, r = torch.linalg.qr(a)
        q = q * torch.sign(torch.diagonal(r))
        if rows < cols:
            q = q.t()
        return (self.gain * q.contiguous().reshape(shape)).to(torch.float32).contiguous()


class TruncatedNormal(Initializer):
    """tf.keras.initializers.TruncatedNormal(mean, stddev): values beyond 2 stddev are re-drawn."""

    def __init__(self, mean: float = 0.0, stddev: float = 0.05, seed=None):
        self.mean, self.stddev, self.seed = float(mean), float(stddev), seed

    def __call__(self, shape, generator):
        t = torch.randn(tuple(shape), generator=generator, dtype=torch.float32)
        for _ in range(64):
            bad = t.abs() > 2.0
            if not bool(bad.any()):
                break
            t = torch.where(bad, torch.randn(tuple(shape), generator=generator, dtype=torch.float32), t)
        return t.clamp_(-2.0, 2.0) * self.stddev + self.mean


class GlorotUniform(Initializer):
    """Keras default kernel initializer (conv6/conv7, network_blocks.py:45-46): U(+-sqrt(6/(fan_in+fan_out))),
    fans include the kernel volume."""

    def __init__(self, seed=None):
        self.seed = seed

    def __call__(self, shape, generator):
        rf = int(math.prod(shape[:-2])) if len(shape) > 2 else 1
        fan_in, fan_out = rf * int(shape[-2]), rf * int(shape[-1])
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(tuple(shape), generator=generator, dtype=torch.float32) * 2.0 - 1.0) * lim


class Zeros(Initializer):
    def __call__(self, shape, generator):
        return torch.zeros(tuple(shape), dtype=torch.float32)


class Ones(Initializer):
    def __call__(self, shape, generator):
        return torch.ones(tuple(shape), dtype=torch.float32)


class L2:
    """tf.keras.regularizers.l2(l2): adds l2*sum(w^2) to the loss."""

    def __init__(self, l2: float = 0.01):
        self.l2 = float(l2)

    def __call__(self, w: torch.Tensor) -> torch.Tensor:
        return self.l2 * (w.float() ** 2).sum()

    def get_config(self):
        return {"class_name": "L2", "config": {"l2": self.l2}}


def l2(l2: float = 0.01) -> L2:  # noqa: A001 - mirrors tf.keras.regularizers.l2
    return L2(l2)


def deserialize(obj):
    """Inverse of get_config() for initializer / regularizer objects (used by LoadableModel.load)."""
    if isinstance(obj, dict) and "class_name" in obj:
        cls = {"Orthogonal": Orthogonal, "TruncatedNormal": TruncatedNormal, "GlorotUniform": GlorotUniform,
               "Zeros": Zeros, "Ones": Ones, "L2": L2}[obj["class_name"]]
        return cls(**obj.get("config", {}))
    return obj
