"""roundtrip_amd — MI355X-native importance-sampling density evaluation for Roundtrip.

Drop-in for the IS path of the reference (`estimate_py_with_IS`, /root/reference/main_density_est.py:266;
`estimate_px_with_IS`, /root/reference/src/pyroundtrip/roundtrip.py:229).  All arithmetic runs in
librt_is.so (hand-written gfx950 HIP kernels behind the C ABI of include/rt_is.h).
"""
__version__ = '0.1.0'
from .nets import BaseFullyConnectedNet, DenseNet, Generator
from .roundtrip import Roundtrip, RoundtripModel

__all__ = ['Roundtrip', 'RoundtripModel', 'BaseFullyConnectedNet', 'Generator', 'DenseNet']
