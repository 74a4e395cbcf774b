from . import networks          # noqa: F401
from . import network_blocks    # noqa: F401
from . import modelio           # noqa: F401
