"""MI355X-native batched TETRA pi/4-DQPSK demodulator (host-side Python surface).

The product is the C-ABI shared library built from csrc/ (include/tetra_demod.h); this package
only builds it (build.py), binds it with ctypes (binding.py) and generates synthetic input
(synth.py).  The directory name contains '-', so import it through the repo-root shim
`tetra_amd.py` (importlib by path) rather than with a plain import statement.
"""
from . import build, binding, chan_binding, scan_binding, lmac_binding, bsync_binding, rx_binding, synth, synth_gpu, shard  # noqa: F401
from .binding import Demodulator, TetraDemodError, load_library  # noqa: F401
from .chan_binding import Channeliser, Resampler  # noqa: F401
from .rx_binding import RxChain  # noqa: F401
