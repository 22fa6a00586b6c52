from .conv import Conv1d, ConvTranspose1d  # noqa: F401
from .residual_block import *  # noqa: F401,F403
