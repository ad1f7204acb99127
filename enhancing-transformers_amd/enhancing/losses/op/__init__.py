from . import conv2d_gradfix  # noqa: F401
from . import conv_nhwc  # noqa: F401
from .fused_act import FusedLeakyReLU, fused_leaky_relu  # noqa: F401
from .upfirdn2d import upfirdn2d  # noqa: F401
