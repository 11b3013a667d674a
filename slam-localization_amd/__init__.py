"""slam-localization_amd: MI355X-native sigma-point Kalman hot path (Usckf / Msckf predict + update).

The directory name carries a hyphen (it mirrors the reference repository's name), so import it
through the loader in the repo root:  `from slkpkg import slk`.
"""
from . import slk  # noqa: F401
from .slk import Msckf, Usckf, SlkError, load_library, device_count  # noqa: F401
