from .hifigan import *  # noqa: F401,F403
from .melgan import *  # noqa: F401,F403
from .parallel_wavegan import *  # noqa: F401,F403
from .style_melgan import *  # noqa: F401,F403
from .uhifigan import *  # noqa: F401,F403
