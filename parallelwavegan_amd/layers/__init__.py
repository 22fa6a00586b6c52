"""The reference's ``parallel_wavegan.layers`` namespace (layers/__init__.py:1-8: star imports, later modules win):
``Conv1d`` / ``Conv1d1x1`` are the residual block's (kaiming-normal weight, zero bias), ``Conv2d`` is the
upsampler's (1 / prod(kernel_size)).  The general convolution modules live in ``layers.conv``."""
from .causal_conv import CausalConv1d, CausalConvTranspose1d  # noqa: F401
from .conv import ConvTranspose1d  # noqa: F401
from .pqmf import *  # noqa: F401,F403
from .residual_stack import *  # noqa: F401,F403
from .residual_block import *  # noqa: F401,F403
from .residual_block import Conv1d, Conv1d1x1  # noqa: F401
from .upsample import Conv2d, ConvInUpsampleNetwork, Stretch2d, UpsampleNetwork  # noqa: F401
from .tade_res_block import TADELayer, TADEResBlock  # noqa: F401
