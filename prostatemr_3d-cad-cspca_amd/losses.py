"""Loss functions with the class surface of the reference's losses.py (tf2.5/scripts/model/losses.py):
``Focal`` (L:20-49) and ``EvidenceLowerBound`` (L:52-63).  ``Focal.loss`` is one fused HIP pass over the softmax heads
(hip.ops.focal_loss: m1_focal_fwd / m1_focal_bwd); like every op of the package it raises on host tensors (the
plain expression it is tested against lives in the oracle, oracle/m1_oracle.py focal_loss).
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

    @staticmethod
    def _gpu(y_pred):
        if not y_pred.is_cuda:
            raise RuntimeError("Focal loss runs on the HIP extension only: move the tensors to a GPU device "
                               "(no CPU fallback exists in this package)")

    def FL(self, y_true, y_pred):
        """L:32-41 for ONE head: renormalise -> clip [eps, 1-eps] -> -y*log p -> * y(1-p)^gamma -> * alpha -> sum_{DHWC} -> mean_b
        (the fused kernel m1_focal_fwd / m1_focal_bwd; head.hip)."""
        self._gpu(y_pred)
        from .hip import ops
        return ops.focal_loss(y_true, y_pred[..., :int(y_true.shape[-1])].contiguous(), self.alpha, self.gamma)

    def loss(self, y_true, y_pred):
        """L:43-49: mean over the y_pred.shape[-1]//y_true.shape[-1] prediction heads (deep supervision), one fused launch."""
        self._gpu(y_pred)
        from .hip import ops
        return ops.focal_loss(y_true, y_pred, self.alpha, self.gamma)


class EvidenceLowerBound:
    """Dummy wrapper: the KL is computed inside the model and passed via y_pred (L:52-63)."""

    def __init__(self, beta=1.00):
        self.beta = beta

    def loss(self, y_true, y_pred):
        return self.beta * y_pred.sum()
