from .causal_conv import CausalConv1d, CausalConvTranspose1d  # noqa: F401
from .conv import Conv2d, ConvTranspose1d  # noqa: F401
from .pqmf import *  # noqa: F401,F403
from .residual_block import *  # noqa: F401,F403
from .residual_stack import *  # noqa: F401,F403
from .upsample import ConvInUpsampleNetwork, Stretch2d, UpsampleNetwork  # noqa: F401
from .tade_res_block import TADELayer, TADEResBlock  # noqa: F401
