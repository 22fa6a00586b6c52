from .adversarial_loss import *  # noqa: F401,F403
from .feat_match_loss import *  # noqa: F401,F403
from .mel_loss import *  # noqa: F401,F403
from .stft_loss import *  # noqa: F401,F403
