"""Reference-shaped import path: ``import model.unets as unets; import model.losses as losses`` (as in the
reference's train_model.py:19-21) resolves to the MI355X-native package ``prostatemr_3d-cad-cspca_amd``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("prostatemr_3d-cad-cspca_amd")
unets = _pkg.unets
losses = _pkg.losses
initializers = _pkg.initializers
optim = _pkg.optim
sys.modules[__name__ + ".unets"] = unets
sys.modules[__name__ + ".unets.networks"] = unets.networks
sys.modules[__name__ + ".unets.network_blocks"] = unets.network_blocks
sys.modules[__name__ + ".unets.modelio"] = unets.modelio
sys.modules[__name__ + ".losses"] = losses
