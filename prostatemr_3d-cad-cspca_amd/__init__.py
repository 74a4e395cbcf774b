"""MI355X-native M1 (hierarchical probabilistic 3D U-Net) hot path.

Drop-in for ``tf2.5/scripts/model`` of DIAGNijmegen/prostateMR_3D-CAD-csPCa for ONE path: the
forward/backward of ``unets.networks.M1`` (reference networks.py / network_blocks.py), plus the loss and
optimiser plumbing a train step needs.  Host side: Python on PyTorch-ROCm (memory, streams, autograd tape,
torch.distributed/RCCL).  Compute: hand-written HIP kernels for gfx950 behind the C ABI of include/m1hip.h
(``libm1hip.so``).  There is no CPU or eager fallback.

The directory name contains a hyphen; import it with ``importlib.import_module("prostatemr_3d-cad-cspca_amd")``
or through the reference-shaped alias package ``model`` at the repo root (``import model.unets as unets``).
"""
from . import hip            # noqa: F401
from . import initializers   # noqa: F401
from . import losses         # noqa: F401
from . import optim          # noqa: F401
from . import unets          # noqa: F401
from . import ddp            # noqa: F401

__all__ = ["hip", "initializers", "losses", "optim", "unets", "ddp"]
