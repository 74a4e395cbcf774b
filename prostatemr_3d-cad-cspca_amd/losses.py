"""Loss functions with the class surface of the reference's losses.py (tf2.5/scripts/model/losses.py):
``Focal`` (L:20-49) and ``EvidenceLowerBound`` (L:52-63).  On the GPU ``Focal.loss`` is one fused HIP pass over the softmax
heads (hip.ops.focal_loss: m1_focal_fwd / m1_focal_bwd); host tensors (unit tests of the host logic) take the
plain expression ``FL`` below, which is also what the fused kernel is tested against.
``SoftDicePlusBoundarySurface`` (L:66-130) needs a CPU scipy distance transform per batch and is not on the
train-step metric path: out of scope.
"""
from __future__ import annotations

import torch

K_EPSILON = 1e-7   # tf.keras.backend.epsilon()


class Focal:
    """[1] T.Y. Lin et al. (2017), "Focal Loss for Dense Object Detection".
    Requires 'y_pred': softmax prediction, 'y_true': one-hot label."""

    def __init__(self, alpha=[0.25, 0.75], gamma=2.00):
        self.alpha = alpha
        self.gamma = gamma
        self._cw = {}        # device -> class-weight tensor (created once: no H2D copy inside a captured step)

    def FL(self, y_true, y_pred):
        """L:32-41: renormalise -> clip [eps, 1-eps] -> -y*log p -> * y(1-p)^gamma -> * alpha -> sum_{DHWC} -> mean_b."""
        key = (y_pred.device, tuple(float(a) for a in self.alpha))
        if key not in self._cw:
            self._cw = {key: torch.as_tensor(self.alpha, dtype=torch.float32, device=y_pred.device)}
        class_weights = self._cw[key]
        y_true = y_true.to(torch.float32)
        y_pred = y_pred / y_pred.sum(dim=-1, keepdim=True)
        y_pred = torch.clamp(y_pred, K_EPSILON, 1 - K_EPSILON)
        ce = y_true * -torch.log(y_pred)
        gamma_weight = y_true * torch.pow(1.0 - y_pred, self.gamma)
        fl = class_weights * (gamma_weight * ce)
        return fl.sum(dim=(1, 2, 3, 4)).mean(dim=0)

    def loss(self, y_true, y_pred):
        """L:43-49: mean over the y_pred.shape[-1]//y_true.shape[-1] prediction heads (deep supervision)."""
        if y_pred.is_cuda:
            from .hip import ops
            return ops.focal_loss(y_true, y_pred, self.alpha, self.gamma)
        c = int(y_true.shape[-1])
        n = int(y_pred.shape[-1]) // c
        elems = [self.FL(y_true, y_pred[..., c * i:c * (i + 1)]) for i in range(n)]
        return torch.stack(elems).mean()


class EvidenceLowerBound:
    """Dummy wrapper: the KL is computed inside the model and passed via y_pred (L:52-63)."""

    def __init__(self, beta=1.00):
        self.beta = beta

    def loss(self, y_true, y_pred):
        return self.beta * y_pred.sum()
