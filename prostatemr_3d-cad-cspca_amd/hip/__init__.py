from . import lib    # noqa: F401
from . import ops    # noqa: F401
from . import graphs # noqa: F401
