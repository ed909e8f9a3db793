"""alphagomoku_amd — MI355X-native self-play MCTS + policy/value evaluation engine (hot path of AlphaGomoku).

Python here is plumbing only (ctypes binding of the C ABI in include/agx.h, synthetic inputs, bench driver).
The product is the HIP library built from alphagomoku_amd/csrc/.
"""
from ._lib import lib, AgxError, check  # noqa: F401
