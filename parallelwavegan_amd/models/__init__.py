from .hifigan import *  # noqa: F401,F403
