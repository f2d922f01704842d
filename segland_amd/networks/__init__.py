from . import pspnet_pop  # noqa: F401  (drivers resolve `networks.<model>.GFSS_Model` like the reference does)
from . import swin_pop    # noqa: F401
