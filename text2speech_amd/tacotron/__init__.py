from .tacotron import Tacotron  # noqa: F401
from .loss_function import Tacotron2Loss  # noqa: F401
