"""`iopath` stand-in: utils/pytorch3d_load_obj.py:47 needs PathManager.open only."""
from . import common  # noqa: F401
