"""pygpa_amd: MI355X-native geometric phase analysis hot path.

Drop-in modules (same names as the reference package):
    pygpa_amd.geometric_phase_analysis, pygpa_amd.cuGPA, pygpa_amd.phase_unwrap,
    pygpa_amd.property_extract, pygpa_amd.mathtools
plus pygpa_amd.distributed (tile sharding over the GPUs of a node).
The HIP library is loaded lazily on the first call (pygpa_amd._lib.load()).

``pinned_empty(shape, dtype)`` returns a NumPy array in page-locked host memory: images kept in such
arrays (and results received through ``out=``) cross PCIe at about twice the rate of pageable memory.
"""
__version__ = '0.1.0'


def pinned_empty(shape, dtype=None):
    import numpy as np
    from ._lib import pinned_empty as _pe
    return _pe(shape, np.float64 if dtype is None else dtype)
