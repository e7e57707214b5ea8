"""pygpa_amd: MI355X-native geometric phase analysis hot path.

Drop-in modules (same names as the reference package):
    pygpa_amd.geometric_phase_analysis, pygpa_amd.cuGPA, pygpa_amd.phase_unwrap
The HIP library is loaded lazily on the first call (pygpa_amd._lib.load()).
"""
__version__ = '0.1.0'
