from .tacotron import Tacotron  # noqa: F401
