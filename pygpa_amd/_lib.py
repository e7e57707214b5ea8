"""ctypes binding of libgpa_hip.so (C ABI declared in include/gpa_hip.h).

The library is the product: if it cannot be loaded, or no GPU is visible, every
entry point raises -- there is no CPU fallback.
"""
import atexit
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPA_HIP_LIB selects another build of the same library (performance experiments, tools/variant.sh)
LIB_PATH = os.environ.get('GPA_HIP_LIB') or os.path.join(_HERE, 'libgpa_hip.so')

GPA_F32, GPA_F64 = 0, 1
_DTYPES = {GPA_F32: (np.float32, np.complex64), GPA_F64: (np.float64, np.complex128)}

# every symbol of include/gpa_hip.h: name -> (restype, argtypes)
_vp, _i, _d, _sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
_dp, _ip, _fp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_float)
SIGNATURES = {
    'gpa_version': (_i, []),
    'gpa_last_error': (C.c_char_p, []),
    'gpa_device_count': (_i, []),
    'gpa_set_option': (_i, [C.c_char_p, C.c_char_p]),
    'gpa_plan_create': (_vp, [_i, _i, _i, _i, _i]),
    'gpa_plan_destroy': (None, [_vp]),
    'gpa_plan_sync': (_i, [_vp]),
    'gpa_plan_workspace_bytes': (_sz, [_vp]),
    'gpa_plan_stream': (_vp, [_vp]),
    'gpa_plan_fft_len': (_i, [_vp, _i]),
    'gpa_lockin_batch': (_i, [_vp, _vp, _vp, _i, _d, _vp]),
    'gpa_lockin_batch_dev': (_i, [_vp, _vp, _vp, _i, _d, _vp]),
    'gpa_sweep': (_i, [_vp, _vp, _vp, _vp, _i, _d, _vp, _vp, _vp]),
    'gpa_sweep_dev': (_i, [_vp, _vp, _vp, _vp, _i, _d, _vp, _vp, _vp]),
    'gpa_sweep_grad': (_i, [_vp, _vp, _vp, _vp, _i, _d, _i, _vp, _vp, _vp]),
    'gpa_sweep_grad_dev': (_i, [_vp, _vp, _vp, _vp, _i, _d, _i, _vp, _vp, _vp]),
    'gpa_sweep_gated': (_i, [_vp, _vp, _vp, _vp, _i, _d, _vp, _vp, _vp]),
    'gpa_reconstruct_prediff': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp]),
    'gpa_invert_u': (_i, [_vp, _vp, _i, _i, _vp]),
    'gpa_invert_u_mode': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'gpa_invert_u_mode_dev': (_i, [_vp, _vp, _d, _i, _i, _i, _i, _vp, _i, _vp]),
    'gpa_undistort_image_dev': (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    'gpa_undistort_image_scaled_dev': (_i, [_vp, _vp, _vp, _d, _vp, _i, _vp, _vp]),
    'gpa_reconstruct_grad': (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    'gpa_reconstruct_grad_dev': (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    'gpa_weighted_lstsq': (_i, [_vp, _vp, _vp, _vp, _i, _vp]),
    'gpa_unwrap_prediff': (_i, [_vp, _vp, _vp, _vp, _i, _d, _i, _vp, _vp]),
    'gpa_unwrap_prediff_dev': (_i, [_vp, _vp, _vp, _vp, _i, _d, _i, _vp, _vp]),
    'gpa_unwrap_prediff_enqueue_dev': (_i, [_vp, _vp, _vp, _vp, _i, _d, _i, _vp]),
    'gpa_unwrap_finish': (_i, [_vp, _vp]),
    'gpa_unwrap': (_i, [_vp, _vp, _vp, _i, _d, _i, _vp, _vp]),
    'gpa_extract_displacement_field': (_i, [_vp, _vp, _vp, _i, _vp, _i, _d, _i, _i, _vp, _vp, _vp, _vp]),
    'gpa_extract_displacement_field_dev': (_i, [_vp, _vp, _vp, _i, _vp, _i, _d, _i, _i, _vp, _vp, _vp, _vp]),
    'gpa_extract_displacement_field_async': (_i, [_vp, _vp, _vp, _i, _vp, _i, _d, _i, _i, _vp, _vp, _vp]),
    'gpa_last_iters': (_i, [_vp, _vp]),
    'gpa_extract_displacement_field_batch_dev': (_i, [_vp, _vp, _i, _vp, _i, _vp, _i, _d, _i, _i, _vp, _vp]),
    'gpa_last_batch_iters': (_i, [_vp, _i, _vp]),
    'gpa_supports_batch': (_i, [_vp]),
    'gpa_extract_gradients': (_i, [_vp, _vp, _vp, _i, _vp, _i, _d, _i, _vp, _vp, _vp]),
    'gpa_mean_dev': (_i, [_vp, _vp, _sz, _dp]),
    'gpa_tile_gradients_dev': (_i, [_vp, _vp, _sz, _i, _i, _d, _vp, _i, _vp, _i, _d, _i, _i, _i, _i, _i,
                                    _vp, _sz, _sz, _vp, _sz, _sz, _vp, _sz]),
    'gpa_tile_sums_dev': (_i, [_vp, _vp, _sz, _sz, _vp, _i, _i, _vp]),
    'gpa_tile_set_mean_dev': (_i, [_vp, _vp, _d]),
    'gpa_tile_gradients_meandev_dev': (_i, [_vp, _vp, _sz, _i, _i, _vp, _i, _vp, _i, _d, _i, _i, _i, _i, _i,
                                            _vp, _sz, _sz, _vp, _sz, _sz, _vp, _sz, _sz]),
    'gpa_stitch_tiles_dev': (_i, [_vp, _vp, _sz, _sz, _sz, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'gpa_plan_wait_stream': (_i, [_vp, _vp]),
    'gpa_stream_wait_plan': (_i, [_vp, _vp]),
    'gpa_invert_u_overlap': (_i, [_vp, _vp, _i, _i, _vp]),
    'gpa_undistort_image': (_i, [_vp, _vp, _vp, _vp]),
    'gpa_phasegradient2J': (_i, [_vp, _vp, _i, _vp, _vp, _d, _vp, _vp]),
    'gpa_phasegradient2J_dev': (_i, [_vp, _vp, _i, _vp, _vp, _d, _vp, _vp]),
    'gpa_lockin_weights_dev': (_i, [_vp, _vp, _i, _vp]),
    'gpa_props_from_jac': (_i, [_i, _i, _sz, _vp, _i, _d, _d, _i, _vp]),
    'gpa_props_from_jac_dev': (_i, [_i, _i, _sz, _vp, _i, _d, _d, _i, _vp, _vp]),
    'gpa_fit_plane': (_i, [_vp, _vp, _i, _d, _dp, _ip]),
    'gpa_fit_plane_dev': (_i, [_vp, _vp, _i, _d, _dp, _ip]),
    'gpa_per_dft': (_i, [_vp, _vp, _vp]),
    'gpa_per_dft_dev': (_i, [_vp, _vp, _vp]),
    'gpa_per': (_i, [_vp, _vp, _i, _vp, _vp]),
    'gpa_gaussian_deconvolve': (_i, [_vp, _vp, _i, _d, _d, _vp]),
    'gpa_gaussian_deconvolve_dev': (_i, [_vp, _vp, _i, _d, _d, _vp]),
    'gpa_find_peaks': (_i, [_vp, _vp, _d, _d, _d, _i, _vp, _vp, _ip, _vp]),
    'gpa_find_peaks_dev': (_i, [_vp, _vp, _d, _d, _d, _i, _vp, _vp, _ip, _vp]),
    'gpa_find_peaks_again': (_i, [_vp, _d, _i, _vp, _vp, _ip]),
    'gpa_timer_start': (_i, [_vp]),
    'gpa_timer_stop': (_i, [_vp, _vp]),
    'gpa_set_profiling': (_i, [_vp, _i]),
    'gpa_last_stage_ms': (_i, [_vp, _vp]),
    'gpa_last_kernel_profile': (_i, [_vp, C.c_char_p, _sz]),
    'gpa_download_async': (_i, [_vp, _vp, _vp, _sz, _i]),
    'gpa_download_wait': (_i, [_vp, _i]),
}

_lib = None
_lock = threading.Lock()


class GPAError(RuntimeError):
    pass


def load():
    """Load libgpa_hip.so (once) and declare every prototype.  Raises GPAError if the
    library is missing -- build it with ``python -m pygpa_amd.build``."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise GPAError('HIP extension %s is missing; build it with `python -m pygpa_amd.build`. '
                           'pygpa_amd has no CPU fallback.' % LIB_PATH)
        try:
            lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        except OSError as e:
            raise GPAError('cannot load %s: %s' % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def last_error():
    msg = load().gpa_last_error()
    return msg.decode() if msg else ''


def check(code, what):
    if code != 0:
        raise GPAError('%s failed (%d): %s' % (what, code, last_error()))


_OPTION_VALUES = {}     # what this process has set through set_option (name -> str); a name that is absent follows the
#                         library's start-up value, i.e. the GPA_<NAME> environment variable read once


def set_option(name, value):
    """gpa_set_option: a diagnostic / test switch of the library ('NO_LAT', 'COLSOLVE', 'F32_EPS_FLOOR', ...; the
    names of INTEGRATION.md).  value None clears it.  The library reads the GPA_<NAME> environment variables once, when
    it is first used; afterwards only this call changes a switch."""
    check(load().gpa_set_option(str(name).encode(), None if value is None else str(value).encode()), 'gpa_set_option')
    _OPTION_VALUES[str(name)] = None if value is None else str(value)


def get_option(name):
    """the value a switch currently has as far as this process can know: what set_option last gave it, else the GPA_<NAME>
    environment variable the library read at start-up (None: unset)"""
    name = str(name)
    if name in _OPTION_VALUES:
        return _OPTION_VALUES[name]
    return os.environ.get('GPA_' + name)


class options:
    """with options(NO_LAT=1, COLSOLVE='tri'): ...  -- switches set for the block; afterwards every one of them is back at the
    value it had before (an outer options() block's, the environment's, or unset), not simply cleared"""

    def __init__(self, **kw):
        self.kw = kw
        self.prev = {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.prev[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            set_option(k, self.prev.get(k))
        return False


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Plan:
    """One device + stream + workspace for images of a fixed shape (wraps gpa_plan)."""

    def __init__(self, shape, max_batch, dtype=np.float64, device=0):
        self.lib = load()
        self.shape = (int(shape[0]), int(shape[1]))
        self.code = GPA_F32 if np.dtype(dtype) == np.float32 else GPA_F64
        self.rdtype, self.cdtype = _DTYPES[self.code]
        self.max_batch = int(max_batch)
        self.device = int(device)
        if self.lib.gpa_device_count() <= 0:
            raise GPAError('no MI355X/HIP device visible; pygpa_amd has no CPU fallback')
        self.handle = self.lib.gpa_plan_create(self.device, self.shape[0], self.shape[1], self.max_batch, self.code)
        if not self.handle:
            raise GPAError('gpa_plan_create failed: ' + last_error())

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.gpa_plan_destroy(self.handle)
            self.handle = None
        for name in ('_d_img', '_d_u'):
            buf = getattr(self, name, None)
            if buf is not None:
                buf.free()
                setattr(self, name, None)
        self._stage = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- helpers -----------------------------------------------------------
    def _img(self, image):
        image = np.ascontiguousarray(image, dtype=self.rdtype)
        if image.shape != self.shape:
            raise ValueError('image shape %s does not match the plan %s' % (image.shape, self.shape))
        return image

    @property
    def workspace_bytes(self):
        return self.lib.gpa_plan_workspace_bytes(self.handle)

    def fft_len(self, axis):
        return self.lib.gpa_plan_fft_len(self.handle, axis)

    def sync(self):
        check(self.lib.gpa_plan_sync(self.handle), 'gpa_plan_sync')

    # ---- host-pointer entry points ------------------------------------------
    def lockin_batch(self, image, kvecs, sigma):
        image = self._img(image)
        kvecs = _f64(kvecs).reshape(-1, 2)
        out = np.empty((len(kvecs),) + self.shape, dtype=self.cdtype)
        check(self.lib.gpa_lockin_batch(self.handle, _ptr(image), _ptr(kvecs), len(kvecs), float(sigma), _ptr(out)),
              'gpa_lockin_batch')
        return out

    def sweep(self, image, kref, klist, sigma, want_kidx=True, want_grad=False, grad_mode=0):
        image = self._img(image)
        kref = _f64(kref).reshape(2)
        klist = _f64(klist).reshape(-1, 2)
        lockin = np.empty(self.shape, dtype=self.cdtype)
        kidx = np.empty(self.shape, dtype=np.int32) if want_kidx else None
        grad = np.empty(self.shape + (2,), dtype=self.rdtype) if want_grad else None
        if want_grad and grad_mode:
            check(self.lib.gpa_sweep_grad(self.handle, _ptr(image), _ptr(kref), _ptr(klist), len(klist), float(sigma),
                                          int(grad_mode), _ptr(lockin), _ptr(kidx), _ptr(grad)), 'gpa_sweep_grad')
        else:
            check(self.lib.gpa_sweep(self.handle, _ptr(image), _ptr(kref), _ptr(klist), len(klist), float(sigma),
                                     _ptr(lockin), _ptr(kidx), _ptr(grad)), 'gpa_sweep')
        return lockin, kidx, grad

    def sweep_gated(self, image, kref, klist, sigma, gate):
        """wfr4's selection rule; gate: (K, K) bool, gate[j, k] = candidate k may replace the kept candidate j"""
        image = self._img(image)
        kref = _f64(kref).reshape(2)
        klist = _f64(klist).reshape(-1, 2)
        gate = np.ascontiguousarray(gate, dtype=np.uint8)
        if gate.shape != (len(klist), len(klist)):
            raise ValueError('gate must be (K, K)')
        lockin = np.empty(self.shape, dtype=self.cdtype)
        kidx = np.empty(self.shape, dtype=np.int32)
        check(self.lib.gpa_sweep_gated(self.handle, _ptr(image), _ptr(kref), _ptr(klist), len(klist), float(sigma),
                                       _ptr(gate), _ptr(lockin), _ptr(kidx)), 'gpa_sweep_gated')
        return lockin, kidx

    def reconstruct_prediff(self, grads, weights, kvecs):
        kvecs = _f64(kvecs).reshape(-1, 2)
        grads = np.ascontiguousarray(grads, dtype=self.rdtype)
        weights = np.ascontiguousarray(weights, dtype=self.rdtype)
        if grads.shape != (len(kvecs),) + self.shape + (2,) or weights.shape != grads.shape[:-1]:
            raise ValueError('grads must be (P,) + plan shape + (2,), weights (P,) + plan shape')
        n0, n1 = self.shape
        dudx = np.empty((2, n0, n1 - 1), dtype=self.rdtype)
        dudy = np.empty((2, n0 - 1, n1), dtype=self.rdtype)
        wnorm = np.empty((n0, n1), dtype=self.rdtype)
        check(self.lib.gpa_reconstruct_prediff(self.handle, _ptr(grads), _ptr(weights), _ptr(kvecs), len(kvecs),
                                               _ptr(dudx), _ptr(dudy), _ptr(wnorm)), 'gpa_reconstruct_prediff')
        return dudx, dudy, wnorm

    _WARP_MODES = {'nearest': 0, 'constant': 1}

    def invert_u(self, us, iters=35, edge=0, mode='nearest'):
        us = np.ascontiguousarray(us, dtype=self.rdtype)
        if us.shape != (2,) + self.shape:
            raise ValueError('us must have shape (2,) + plan shape')
        if mode not in self._WARP_MODES:
            raise NotImplementedError("mode must be 'nearest' or 'constant'")
        out = np.empty((2,) + self.shape, dtype=self.rdtype)
        check(self.lib.gpa_invert_u_mode(self.handle, _ptr(us), int(iters), int(edge), 0, self._WARP_MODES[mode], _ptr(out)),
              'gpa_invert_u_mode')
        return out

    def reconstruct_grad(self, lockins, kvecs, mask_border):
        lockins = np.ascontiguousarray(lockins, dtype=self.cdtype)
        kvecs = _f64(kvecs).reshape(-1, 2)
        n0, n1 = self.shape
        dudx = np.empty((2, n0, n1 - 1), dtype=self.rdtype)
        dudy = np.empty((2, n0 - 1, n1), dtype=self.rdtype)
        wnorm = np.empty((n0, n1), dtype=self.rdtype)
        check(self.lib.gpa_reconstruct_grad(self.handle, _ptr(lockins), _ptr(kvecs), len(kvecs), int(mask_border),
                                            _ptr(dudx), _ptr(dudy), _ptr(wnorm)), 'gpa_reconstruct_grad')
        return dudx, dudy, wnorm

    def weighted_lstsq(self, b, weights, kvecs):
        b = np.ascontiguousarray(b, dtype=self.rdtype)
        weights = np.ascontiguousarray(weights, dtype=self.rdtype)
        kvecs = _f64(kvecs).reshape(-1, 2)
        if b.shape != (len(kvecs),) + self.shape or weights.shape != b.shape:
            raise ValueError('b / weights must have shape (P,) + plan shape')
        out = np.empty((2,) + self.shape, dtype=self.rdtype)
        check(self.lib.gpa_weighted_lstsq(self.handle, _ptr(b), _ptr(weights), _ptr(kvecs), len(kvecs), _ptr(out)),
              'gpa_weighted_lstsq')
        return out

    def invert_u_overlap(self, us, iters=35, edge=0, mode='nearest'):
        us = np.ascontiguousarray(us, dtype=self.rdtype)
        if us.shape != (2,) + self.shape:
            raise ValueError('us must have shape (2,) + plan shape')
        if mode not in self._WARP_MODES:
            raise NotImplementedError("mode must be 'nearest' or 'constant'")
        out = np.empty((2, self.shape[0] + 2 * edge, self.shape[1] + 2 * edge), dtype=self.rdtype)
        check(self.lib.gpa_invert_u_mode(self.handle, _ptr(us), int(iters), int(edge), 1, self._WARP_MODES[mode], _ptr(out)),
              'gpa_invert_u_mode')
        return out

    def undistort_image(self, deformed, u):
        deformed = self._img(deformed)
        u = np.ascontiguousarray(u, dtype=self.rdtype)
        if u.shape != (2,) + self.shape:
            raise ValueError('u must have shape (2,) + plan shape')
        out = np.empty(self.shape, dtype=self.rdtype)
        check(self.lib.gpa_undistort_image(self.handle, _ptr(deformed), _ptr(u), _ptr(out)), 'gpa_undistort_image')
        return out

    @staticmethod
    def _rects(rects):
        """None, one (r0, c0, h, w) or a list of them -> (ctypes int array or None, count)"""
        if rects is None:
            return None, 0
        flat = np.asarray(rects, dtype=np.int64).reshape(-1)
        if flat.size % 4:
            raise ValueError('rects: (r0, c0, h, w) per window')
        return (C.c_int * flat.size)(*[int(v) for v in flat]), flat.size // 4

    def invert_u_dev(self, u_ptr, out_ptr, scale=1.0, iters=35, edge=0, overlap=True, mode='nearest', rects=None):
        """invert_u_overlap / invert_u of scale * u on device pointers, enqueued on the plan's stream (no host sync);
        rects = (r0, c0, h, w) or a list of such windows: only those parts of the output grid are computed"""
        arr, n = self._rects(rects)
        check(self.lib.gpa_invert_u_mode_dev(self.handle, _ptr(int(u_ptr)), float(scale), int(iters), int(edge), int(bool(overlap)),
                                             self._WARP_MODES[mode], arr, n, _ptr(int(out_ptr))), 'gpa_invert_u_mode_dev')

    def undistort_image_dev(self, deformed_ptr, u_ptr, out_ptr, uinv_ptr=None, rects=None, scale=1.0):
        """undistort_image(deformed, scale * u) on device pointers, enqueued on the plan's stream (no host sync); uinv_ptr
        (2 x n0 x n1) receives u_inv = invert_u_overlap(-scale * u); rects = (r0, c0, h, w) or a list: only those windows of the
        outputs are computed.  scale = -1: u is the field as extract_displacement_field returns it (minus the displacement)"""
        arr, n = self._rects(rects)
        check(self.lib.gpa_undistort_image_scaled_dev(self.handle, _ptr(int(deformed_ptr)), _ptr(int(u_ptr)), float(scale), arr, n,
                                                      _ptr(None if uinv_ptr is None else int(uinv_ptr)), _ptr(int(out_ptr))),
              'gpa_undistort_image_dev')

    def phasegradient2J(self, kvecs, grads, weights, nmperpixel, dks=None):
        kvecs = _f64(kvecs).reshape(-1, 2)
        grads = np.ascontiguousarray(grads, dtype=self.rdtype)
        weights = np.ascontiguousarray(weights, dtype=self.rdtype)
        if grads.shape != (len(kvecs),) + self.shape + (2,) or weights.shape != grads.shape[:-1]:
            raise ValueError('grads must be (P,) + plan shape + (2,), weights (P,) + plan shape')
        dks = None if dks is None else _f64(dks).reshape(len(kvecs), 2)
        J = np.empty(self.shape + (2, 2), dtype=self.rdtype)
        check(self.lib.gpa_phasegradient2J(self.handle, _ptr(kvecs), len(kvecs), _ptr(grads), _ptr(weights),
                                           float(nmperpixel), _ptr(dks), _ptr(J)), 'gpa_phasegradient2J')
        return J

    # ---- device-pointer forms of the rows after the path (f-2, f-4): nothing leaves HBM -----------------------------
    def sweep_grad_dev(self, image_ptr, kref, klist, sigma, lockin_ptr, grad_ptr, kidx_ptr=None, grad_mode=0):
        """wfr2_grad_opt of one peak on device pointers: lockin (n0 x n1 complex), grad (n0 x n1 x 2), optional kidx"""
        kref = _f64(kref).reshape(2)
        klist = _f64(klist).reshape(-1, 2)
        check(self.lib.gpa_sweep_grad_dev(self.handle, _ptr(int(image_ptr)), _ptr(kref), _ptr(klist), len(klist), float(sigma),
                                          int(grad_mode), _ptr(int(lockin_ptr)),
                                          _ptr(None if kidx_ptr is None else int(kidx_ptr)), _ptr(int(grad_ptr))),
              'gpa_sweep_grad_dev')

    def lockin_weights_dev(self, lockins_ptr, P, weights_ptr):
        """np.abs of P lock-ins on the device"""
        check(self.lib.gpa_lockin_weights_dev(self.handle, _ptr(int(lockins_ptr)), int(P), _ptr(int(weights_ptr))),
              'gpa_lockin_weights_dev')

    def phasegradient2J_dev(self, kvecs, grads_ptr, weights_ptr, nmperpixel, J_ptr, dks=None):
        """phasegradient2J on device pointers: grads (P x n0 x n1 x 2), weights (P x n0 x n1) -> J (n0 x n1 x 2 x 2)"""
        kvecs = _f64(kvecs).reshape(-1, 2)
        dks = None if dks is None else _f64(dks).reshape(len(kvecs), 2)
        check(self.lib.gpa_phasegradient2J_dev(self.handle, _ptr(kvecs), len(kvecs), _ptr(int(grads_ptr)),
                                               _ptr(int(weights_ptr)), float(nmperpixel), _ptr(dks), _ptr(int(J_ptr))),
              'gpa_phasegradient2J_dev')

    def props_from_jac_dev(self, jac_ptr, props_ptr, add_identity=False, refangle=0., refscale=1., diff=False):
        """props_from_Jac on device pointers (n0 x n1 x 2 x 2 -> 4 x n0 x n1), on the plan's stream"""
        check(self.lib.gpa_props_from_jac_dev(self.device, self.code, self.shape[0] * self.shape[1], _ptr(int(jac_ptr)),
                                              int(bool(add_identity)), float(refangle), float(refscale), int(bool(diff)),
                                              _ptr(int(props_ptr)), _ptr(self.stream())), 'gpa_props_from_jac_dev')

    def fit_plane_dev(self, image_ptr, max_iter=200, tol=1e-12):
        coef = (C.c_double * 3)()
        iters = C.c_int(0)
        check(self.lib.gpa_fit_plane_dev(self.handle, _ptr(int(image_ptr)), int(max_iter), float(tol), coef, C.byref(iters)),
              'gpa_fit_plane_dev')
        return np.array(coef[:]), iters.value

    def fit_plane(self, image, max_iter=200, tol=1e-12):
        image = self._img(image)
        coef = (C.c_double * 3)()
        iters = C.c_int(0)
        check(self.lib.gpa_fit_plane(self.handle, _ptr(image), int(max_iter), float(tol), coef, C.byref(iters)),
              'gpa_fit_plane')
        return np.array(coef[:]), iters.value

    def find_peaks(self, image, sigma, dog_sigma, threshold_rel, want_smooth=False, max_out=4096):
        """peak_local_max candidates of the smoothed spectrum: (coords (n, 2) int, values (n,)) sorted like
        skimage (highest first, raster order among equals) [, smooth]."""
        image = self._img(image)
        smooth = np.empty(self.shape, dtype=self.rdtype) if want_smooth else None
        tried_full = False
        while True:
            coords = np.empty((max_out, 2), dtype=np.int32)
            vals = np.empty(max_out, dtype=self.rdtype)
            count = C.c_int(0)
            check(self.lib.gpa_find_peaks(self.handle, _ptr(image), float(sigma), float(dog_sigma), float(threshold_rel),
                                          int(max_out), _ptr(coords), _ptr(vals), C.byref(count), _ptr(smooth)),
                  'gpa_find_peaks')
            if count.value <= max_out or tried_full:
                break
            max_out, tried_full = count.value, True   # (the library caps max_out at its buffer size)
        n = min(count.value, max_out)
        coords, vals = coords[:n], vals[:n]
        order = np.lexsort((coords[:, 1], coords[:, 0], -vals))
        out = (coords[order].astype(np.intp), vals[order])
        return out + (smooth,) if want_smooth else out

    def find_peaks_dev(self, image_ptr, sigma, dog_sigma, threshold_rel, smooth_ptr=None, max_out=4096):
        """find_peaks on a device image (not modified); the candidate list comes back to the host, the smoothed spectrum
        (optional) stays on the device"""
        coords = np.empty((max_out, 2), dtype=np.int32)
        vals = np.empty(max_out, dtype=self.rdtype)
        count = C.c_int(0)
        check(self.lib.gpa_find_peaks_dev(self.handle, _ptr(int(image_ptr)), float(sigma), float(dog_sigma),
                                          float(threshold_rel), int(max_out), _ptr(coords), _ptr(vals), C.byref(count),
                                          _ptr(None if smooth_ptr is None else int(smooth_ptr))), 'gpa_find_peaks_dev')
        n = min(count.value, max_out)
        coords, vals = coords[:n], vals[:n]
        order = np.lexsort((coords[:, 1], coords[:, 0], -vals))
        return coords[order].astype(np.intp), vals[order]

    def find_peaks_again(self, threshold_rel, max_out=4096):
        """the candidates of the smoothed spectrum of the LAST find_peaks / find_peaks_dev call at another threshold"""
        tried_full = False
        while True:
            coords = np.empty((max_out, 2), dtype=np.int32)
            vals = np.empty(max_out, dtype=self.rdtype)
            count = C.c_int(0)
            check(self.lib.gpa_find_peaks_again(self.handle, float(threshold_rel), int(max_out), _ptr(coords), _ptr(vals),
                                                C.byref(count)), 'gpa_find_peaks_again')
            if count.value <= max_out or tried_full:
                break
            max_out, tried_full = count.value, True
        n = min(count.value, max_out)
        coords, vals = coords[:n], vals[:n]
        order = np.lexsort((coords[:, 1], coords[:, 0], -vals))
        return coords[order].astype(np.intp), vals[order]

    def per_dft_dev(self, image_ptr, out_ptr):
        """per_dft on device pointers (n0 x n1 reals in, n0 x n1 complex out), enqueued on the plan's stream"""
        check(self.lib.gpa_per_dft_dev(self.handle, _ptr(int(image_ptr)), _ptr(int(out_ptr))), 'gpa_per_dft_dev')

    def gaussian_deconvolve_dev(self, data_ptr, dr, sigma, balance, out_ptr):
        """gaussian_deconvolve of one m0 x m1 device field (this plan: the padded shape); out_ptr may equal data_ptr"""
        check(self.lib.gpa_gaussian_deconvolve_dev(self.handle, _ptr(int(data_ptr)), int(dr), float(sigma), float(balance),
                                                   _ptr(int(out_ptr))), 'gpa_gaussian_deconvolve_dev')

    def gaussian_deconvolve(self, data, dr, sigma, balance):
        """one m0 x m1 field; this plan has the padded shape (m0 + 4 dr, m1 + 4 dr)"""
        data = np.ascontiguousarray(data, dtype=self.rdtype)
        if data.shape != (self.shape[0] - 4 * dr, self.shape[1] - 4 * dr):
            raise ValueError('data shape %s does not match the padded plan %s' % (data.shape, self.shape))
        out = np.empty(data.shape, dtype=self.rdtype)
        check(self.lib.gpa_gaussian_deconvolve(self.handle, _ptr(data), int(dr), float(sigma), float(balance), _ptr(out)),
              'gpa_gaussian_deconvolve')
        return out

    def per_dft(self, image):
        image = self._img(image)
        out = np.empty(self.shape, dtype=self.cdtype)
        check(self.lib.gpa_per_dft(self.handle, _ptr(image), _ptr(out)), 'gpa_per_dft')
        return out

    def per(self, image, inverse_dft=True):
        """moisan2011.per: (p, s) real for inverse_dft=True, their DFTs (p_hat, s_hat) otherwise"""
        image = self._img(image)
        dt = self.rdtype if inverse_dft else self.cdtype
        pc, sc = np.empty(self.shape, dtype=dt), np.empty(self.shape, dtype=dt)
        check(self.lib.gpa_per(self.handle, _ptr(image), 1 if inverse_dft else 0, _ptr(pc), _ptr(sc)), 'gpa_per')
        return pc, sc

    def unwrap_prediff(self, dx, dy, weight=None, kmax=100, eps=1e-9, axes_compat=True):
        n0, n1 = self.shape
        dx = np.ascontiguousarray(dx, dtype=self.rdtype)
        dy = np.ascontiguousarray(dy, dtype=self.rdtype)
        if dx.shape != (n0, n1 - 1) or dy.shape != (n0 - 1, n1):
            raise ValueError('dx/dy shapes %s %s do not match the plan %s' % (dx.shape, dy.shape, self.shape))
        w = None if weight is None else self._img(weight)
        phi = np.empty((n0, n1), dtype=self.rdtype)
        iters = C.c_int(0)
        check(self.lib.gpa_unwrap_prediff(self.handle, _ptr(dx), _ptr(dy), _ptr(w), int(kmax), float(eps),
                                          1 if axes_compat else 0, _ptr(phi), C.byref(iters)), 'gpa_unwrap_prediff')
        return phi, iters.value

    def unwrap(self, psi, weight=None, kmax=100, eps=1e-9, axes_compat=True):
        psi = self._img(psi)
        w = None if weight is None else self._img(weight)
        phi = np.empty(self.shape, dtype=self.rdtype)
        iters = C.c_int(0)
        check(self.lib.gpa_unwrap(self.handle, _ptr(psi), _ptr(w), int(kmax), float(eps), 1 if axes_compat else 0,
                                  _ptr(phi), C.byref(iters)), 'gpa_unwrap')
        return phi, iters.value

    def extract_displacement_field(self, image, kvecs, klists, sigma, mask_border, kmax=10,
                                   want_lockins=False, want_kidx=False, out=None):
        kvecs = _f64(kvecs).reshape(-1, 2)
        klists = _f64(klists)
        P = len(kvecs)
        klists = klists.reshape(P, -1, 2)
        K = klists.shape[1]
        if out is not None and (out.shape != (2,) + self.shape or out.dtype != self.rdtype or not out.flags.c_contiguous):
            raise ValueError('out must be a C-contiguous (2,) + plan shape array of the plan dtype')
        if not want_lockins and not want_kidx and self.shape[0] * self.shape[1] >= 512 * 512:
            return self._extract_host_pipelined(image, kvecs, klists, sigma, mask_border, kmax, out)
        u = out if out is not None else np.empty((2,) + self.shape, dtype=self.rdtype)
        image = self._img(image)
        lock = np.empty((P,) + self.shape, dtype=self.cdtype) if want_lockins else None
        kidx = np.empty((P,) + self.shape, dtype=np.int32) if want_kidx else None
        iters = (C.c_int * 2)()
        check(self.lib.gpa_extract_displacement_field(self.handle, _ptr(image), _ptr(kvecs), P, _ptr(klists), K,
                                                      float(sigma), int(mask_border), int(kmax), _ptr(u), _ptr(lock),
                                                      _ptr(kidx), iters), 'gpa_extract_displacement_field')
        return u, lock, kidx, (iters[0], iters[1])

    def _extract_host_pipelined(self, image, kvecs, klists, sigma, mask_border, kmax, out):
        """The host-array form of the driver with its host work overlapped.  A 4096^2 f32 call spent 18 of its 29 ms on
        the host: `astype` of the image into a fresh array (8.6 ms) and page-faulting a fresh 128 MB result array
        during the download (8 ms).  Here the image is converted into a staging array kept by the plan (resident
        pages: 4.3 ms), the device work is enqueued asynchronously, the result array is allocated and touched WHILE the
        GPU computes, and the download lands in resident pages (2.4 ms).  (Several host threads for the conversion and
        the first touch were tried and dropped: on the 256-core box the pages end up on other NUMA nodes and the
        copies slow down by more than the threads save.)"""
        image = np.asarray(image)
        if image.shape != self.shape:
            raise ValueError('image shape %s does not match the plan %s' % (image.shape, self.shape))
        npx = self.shape[0] * self.shape[1]
        item = np.dtype(self.rdtype).itemsize
        if getattr(self, '_d_img', None) is None:
            self._d_img, self._d_u = DeviceBuffer(npx * item, self.device), DeviceBuffer(2 * npx * item, self.device)
        if image.dtype == self.rdtype and image.flags.c_contiguous:
            src = image
        else:
            if getattr(self, '_stage', None) is None:
                self._stage = np.empty(self.shape, dtype=self.rdtype)
            src = self._stage
            np.copyto(src, image, casting='unsafe')
        self._d_img.upload(src)
        self.extract_displacement_field_async(self._d_img.ptr, kvecs, klists, sigma, mask_border, kmax, self._d_u.ptr)
        if out is None:
            out = np.empty((2,) + self.shape, dtype=self.rdtype)
            out.fill(0)                    # first touch while the GPU computes
        iters = self.last_iters()          # waits for the device
        self._d_u.download_into(out)
        return out, None, None, iters

    def extract_gradients(self, image, kvecs, klists, sigma, mask_border):
        image = self._img(image)
        kvecs = _f64(kvecs).reshape(-1, 2)
        P = len(kvecs)
        klists = _f64(klists).reshape(P, -1, 2)
        n0, n1 = self.shape
        dudx = np.empty((2, n0, n1 - 1), dtype=self.rdtype)
        dudy = np.empty((2, n0 - 1, n1), dtype=self.rdtype)
        wnorm = np.empty((n0, n1), dtype=self.rdtype)
        check(self.lib.gpa_extract_gradients(self.handle, _ptr(image), _ptr(kvecs), P, _ptr(klists), klists.shape[1],
                                             float(sigma), int(mask_border), _ptr(dudx), _ptr(dudy), _ptr(wnorm)),
              'gpa_extract_gradients')
        return dudx, dudy, wnorm

    # ---- device-pointer entry points (ints from torch.Tensor.data_ptr()) -----
    def mean_dev(self, ptr, count):
        out = C.c_double(0.0)
        check(self.lib.gpa_mean_dev(self.handle, _ptr(int(ptr)), int(count), C.byref(out)), 'gpa_mean_dev')
        return out.value

    def tile_gradients_dev(self, image_ptr, image_pitch, r0, c0, mean, kvecs, klists, sigma, mask_border,
                           interior, dx, dy, wn):
        """interior = (i0, j0, t0, t1); dx, dy = (ptr, pitch, plane), wn = (ptr, pitch); element units."""
        kvecs = _f64(kvecs).reshape(-1, 2)
        P = len(kvecs)
        klists = _f64(klists).reshape(P, -1, 2)
        i0, j0, t0, t1 = (int(v) for v in interior)
        check(self.lib.gpa_tile_gradients_dev(self.handle, _ptr(int(image_ptr)), int(image_pitch), int(r0), int(c0),
                                              float(mean), _ptr(kvecs), P, _ptr(klists), klists.shape[1], float(sigma),
                                              int(mask_border), i0, j0, t0, t1, _ptr(int(dx[0])), int(dx[1]), int(dx[2]),
                                              _ptr(int(dy[0])), int(dy[1]), int(dy[2]), _ptr(int(wn[0])), int(wn[1])),
              'gpa_tile_gradients_dev')

    def tile_sums_dev(self, wins_ptr, win_stride, win_pitch, rects_dev_ptr, ntiles, max_rows, sum_dev_ptr):
        """sum over the interior rectangles (device table of (o0, o1, z0, z1)) of ntiles windows -> one double on the device"""
        check(self.lib.gpa_tile_sums_dev(self.handle, _ptr(int(wins_ptr)), int(win_stride), int(win_pitch),
                                         _ptr(int(rects_dev_ptr)), int(ntiles), int(max_rows), _ptr(int(sum_dev_ptr))),
              'gpa_tile_sums_dev')

    def tile_set_mean_dev(self, sum_dev_ptr, scale):
        check(self.lib.gpa_tile_set_mean_dev(self.handle, _ptr(int(sum_dev_ptr)), float(scale)), 'gpa_tile_set_mean_dev')

    def tile_gradients_meandev_dev(self, image_ptr, image_pitch, r0, c0, kvecs, klists, sigma, mask_border, interior, dx, dy, wn):
        """tile_gradients_dev with the mean of gpa_tile_set_mean_dev; wn = (ptr, pitch, plane): plane != 0 writes the
        weight twice (plane elements apart)"""
        kvecs = _f64(kvecs).reshape(-1, 2)
        P = len(kvecs)
        klists = _f64(klists).reshape(P, -1, 2)
        i0, j0, t0, t1 = (int(v) for v in interior)
        check(self.lib.gpa_tile_gradients_meandev_dev(self.handle, _ptr(int(image_ptr)), int(image_pitch), int(r0), int(c0),
                                                      _ptr(kvecs), P, _ptr(klists), klists.shape[1], float(sigma),
                                                      int(mask_border), i0, j0, t0, t1, _ptr(int(dx[0])), int(dx[1]), int(dx[2]),
                                                      _ptr(int(dy[0])), int(dy[1]), int(dy[2]), _ptr(int(wn[0])), int(wn[1]),
                                                      int(wn[2])),
              'gpa_tile_gradients_meandev_dev')

    def stitch_tiles_dev(self, tiles_ptr, slot_stride, field_stride, tile_pitch, table_dev_ptr, ntiles, t0, t1, dst):
        """dst: list of (device pointer, pitch, rows, cols) per field (<= 6); one launch on the plan's stream"""
        nf = len(dst)
        ptrs = (C.c_void_p * nf)(*[int(d[0]) for d in dst])
        pitch = (C.c_size_t * nf)(*[int(d[1]) for d in dst])
        rows = (C.c_int * nf)(*[int(d[2]) for d in dst])
        cols = (C.c_int * nf)(*[int(d[3]) for d in dst])
        check(self.lib.gpa_stitch_tiles_dev(self.handle, _ptr(int(tiles_ptr)), int(slot_stride), int(field_stride),
                                            int(tile_pitch), _ptr(int(table_dev_ptr)), int(ntiles), int(t0), int(t1), nf,
                                            ptrs, pitch, rows, cols), 'gpa_stitch_tiles_dev')

    def stream(self):
        """the plan's main stream (raw hipStream_t as an int)"""
        return int(self.lib.gpa_plan_stream(self.handle) or 0)

    def wait_stream(self, stream):
        """the plan's stream waits for what has been enqueued on `stream` (raw hipStream_t as an int, 0 = default)"""
        check(self.lib.gpa_plan_wait_stream(self.handle, C.c_void_p(int(stream))), 'gpa_plan_wait_stream')

    def stream_wait(self, stream):
        """`stream` waits for what has been enqueued on the plan's stream"""
        check(self.lib.gpa_stream_wait_plan(self.handle, C.c_void_p(int(stream))), 'gpa_stream_wait_plan')

    def unwrap_prediff_dev(self, dx_ptr, dy_ptr, weight_ptr, phi_ptr, kmax=100, eps=1e-9, axes_compat=True):
        iters = C.c_int(0)
        check(self.lib.gpa_unwrap_prediff_dev(self.handle, _ptr(int(dx_ptr)), _ptr(int(dy_ptr)),
                                              _ptr(None if weight_ptr is None else int(weight_ptr)), int(kmax), float(eps),
                                              int(bool(axes_compat)), _ptr(int(phi_ptr)), C.byref(iters)),
              'gpa_unwrap_prediff_dev')
        return iters.value

    def unwrap_prediff_enqueue_dev(self, dx_ptr, dy_ptr, weight_ptr, phi_ptr, kmax=100, eps=1e-9, axes_compat=True):
        """the solve of unwrap_prediff_dev put on the plan's stream without waiting for it (unwrap_finish does)"""
        check(self.lib.gpa_unwrap_prediff_enqueue_dev(self.handle, _ptr(int(dx_ptr)), _ptr(int(dy_ptr)),
                                                      _ptr(None if weight_ptr is None else int(weight_ptr)), int(kmax),
                                                      float(eps), int(bool(axes_compat)), _ptr(int(phi_ptr))),
              'gpa_unwrap_prediff_enqueue_dev')

    def unwrap_finish(self):
        iters = C.c_int(0)
        check(self.lib.gpa_unwrap_finish(self.handle, C.byref(iters)), 'gpa_unwrap_finish')
        return iters.value

    def extract_displacement_field_dev(self, image_ptr, kvecs, klists, sigma, mask_border, kmax, u_ptr,
                                       lockins_ptr=None, kidx_ptr=None):
        kvecs = _f64(kvecs).reshape(-1, 2)
        P = len(kvecs)
        klists = _f64(klists).reshape(P, -1, 2)
        iters = (C.c_int * 2)()
        check(self.lib.gpa_extract_displacement_field_dev(self.handle, _ptr(image_ptr), _ptr(kvecs), P, _ptr(klists),
                                                          klists.shape[1], float(sigma), int(mask_border), int(kmax),
                                                          _ptr(u_ptr), _ptr(lockins_ptr), _ptr(kidx_ptr), iters),
              'gpa_extract_displacement_field_dev')
        return iters[0], iters[1]

    def extract_displacement_field_async(self, image_ptr, kvecs, klists, sigma, mask_border, kmax, u_ptr,
                                         lockins_ptr=None, kidx_ptr=None):
        """enqueue only; pair with sync() / last_iters()"""
        kvecs = _f64(kvecs).reshape(-1, 2)
        P = len(kvecs)
        klists = _f64(klists).reshape(P, -1, 2)
        check(self.lib.gpa_extract_displacement_field_async(self.handle, _ptr(image_ptr), _ptr(kvecs), P, _ptr(klists),
                                                            klists.shape[1], float(sigma), int(mask_border), int(kmax),
                                                            _ptr(u_ptr), _ptr(lockins_ptr), _ptr(kidx_ptr)),
              'gpa_extract_displacement_field_async')

    def extract_displacement_field_batch_dev(self, images_ptr, nimages, kvecs, klists, sigma, mask_border, kmax, u_ptr,
                                             want_iters=True):
        """a stack of `nimages` images (device pointers: images nimages x n0 x n1, u nimages x 2 x n0 x n1): sweeps
        image after image, the 2 * nimages unwraps in one set of launches.  Returns the (nimages, 2) iteration counts,
        or None without synchronising when want_iters is False (pair with sync())"""
        kvecs = _f64(kvecs).reshape(-1, 2)
        P = len(kvecs)
        klists = _f64(klists).reshape(P, -1, 2)
        it = (C.c_int * (2 * int(nimages)))() if want_iters else None
        check(self.lib.gpa_extract_displacement_field_batch_dev(self.handle, _ptr(images_ptr), int(nimages), _ptr(kvecs), P,
                                                                _ptr(klists), klists.shape[1], float(sigma), int(mask_border),
                                                                int(kmax), _ptr(u_ptr), it),
              'gpa_extract_displacement_field_batch_dev')
        return None if it is None else np.array(it[:], dtype=np.int64).reshape(int(nimages), 2)

    def extract_displacement_field_stack(self, images, kvecs, klists, sigma, mask_border, kmax=10, chunk=None, out=None):
        """host arrays: images (B, n0, n1) -> u (B, 2, n0, n1) (written into `out` if given) and the (B, 2)
        iteration counts, through extract_displacement_field_batch_dev.  The stack goes through the GPU in chunks of
        `chunk` frames (default: ~16 Mpixel per chunk) on two sets of device buffers: while the GPU works on chunk i a
        host thread uploads chunk i + 1 and another downloads chunk i - 1 (PCIe is full duplex, plain hipMemcpy on the
        default stream does not wait for the plan's non-blocking streams, ctypes releases the GIL)."""
        from concurrent.futures import ThreadPoolExecutor
        images = np.asarray(images)
        if images.ndim != 3 or images.shape[1:] != tuple(self.shape):
            raise ValueError('images must be (B, %d, %d)' % tuple(self.shape))
        B = images.shape[0]
        npx = int(self.shape[0]) * int(self.shape[1])
        if chunk is None:
            chunk = max(1, min(B, (16 << 20) // npx))
        chunk = int(max(1, min(chunk, B, 4096)))     # the batched driver takes at most 4096 frames per call
        nchunks = -(-B // chunk)
        item = np.dtype(self.rdtype).itemsize
        # (a plain array: page-locking hundreds of MB costs more than it saves -- 104 ms for 256 MB on the MI355X box,
        #  where hipMemcpy into pageable memory runs at ~50 GB/s anyway; pass out= to reuse a buffer)
        u = out if out is not None else np.empty((B, 2) + tuple(self.shape), self.rdtype)
        if u.shape != (B, 2) + tuple(self.shape) or u.dtype != self.rdtype or not u.flags.c_contiguous:
            raise ValueError('out must be a C-contiguous (B, 2, n0, n1) array of the plan dtype')
        iters = np.zeros((B, 2), dtype=np.int64)
        d_img = [DeviceBuffer(chunk * npx * item, self.device) for _ in range(min(2, nchunks))]
        d_u = [DeviceBuffer(2 * chunk * npx * item, self.device) for _ in range(min(2, nchunks))]
        bounds = [(c * chunk, min(B, (c + 1) * chunk)) for c in range(nchunks)]
        batched = bool(self.lib.gpa_supports_batch(self.handle))   # asked once, not discovered per chunk from an error text

        def upload(c):
            a, b = bounds[c]
            d_img[c % 2].upload(np.ascontiguousarray(images[a:b], dtype=self.rdtype))

        def download(c):
            a, b = bounds[c]
            d_u[c % 2].download_into(u[a:b])

        try:
            with ThreadPoolExecutor(1) as up_pool, ThreadPoolExecutor(1) as dn_pool:
                ups = {0: up_pool.submit(upload, 0)}
                dns = {}
                for c in range(nchunks):
                    ups[c].result()                       # chunk c is on the device
                    if c - 2 in dns:
                        dns.pop(c - 2).result()           # its u buffer has been read out
                    a, b = bounds[c]
                    if batched:
                        self.extract_displacement_field_batch_dev(d_img[c % 2].ptr, b - a, kvecs, klists, sigma, mask_border,
                                                                  kmax, d_u[c % 2].ptr, want_iters=False)
                    else:
                        # shapes without a batched unwrap (sides no fused path covers): the same frames one call each --
                        # the numbers the docstring promises, just without the shared launch chain
                        for f in range(b - a):
                            iters[a + f] = self.extract_displacement_field_dev(d_img[c % 2].ptr + f * npx * item, kvecs, klists, sigma,
                                                                               mask_border, kmax, d_u[c % 2].ptr + 2 * f * npx * item)
                    if c + 1 < nchunks:                    # (its image buffer was last read by chunk c - 1: done, see sync below)
                        ups[c + 1] = up_pool.submit(upload, c + 1)
                    if batched:
                        iters[a:b] = self._batch_iters(b - a)   # synchronises: chunk c is finished
                    dns[c] = dn_pool.submit(download, c)
                for f in dns.values():
                    f.result()
        finally:
            for buf in d_img + d_u:
                buf.free()
        return u, iters

    def _batch_iters(self, nimages):
        it = (C.c_int * (2 * int(nimages)))()
        check(self.lib.gpa_last_batch_iters(self.handle, int(nimages), it), 'gpa_last_batch_iters')
        return np.array(it[:], dtype=np.int64).reshape(int(nimages), 2)

    def last_iters(self):
        it = (C.c_int * 2)()
        check(self.lib.gpa_last_iters(self.handle, it), 'gpa_last_iters')
        return it[0], it[1]

    def timer_start(self):
        check(self.lib.gpa_timer_start(self.handle), 'gpa_timer_start')

    def timer_stop(self):
        ms = C.c_float(0)
        check(self.lib.gpa_timer_stop(self.handle, C.byref(ms)), 'gpa_timer_stop')
        return ms.value

    def set_profiling(self, on):
        check(self.lib.gpa_set_profiling(self.handle, 1 if on else 0), 'gpa_set_profiling')

    def last_kernel_profile(self):
        """{kernel name: (launches, total ms)} of the last profiled fused-driver call"""
        buf = C.create_string_buffer(1 << 16)
        check(self.lib.gpa_last_kernel_profile(self.handle, buf, len(buf)), 'gpa_last_kernel_profile')
        out = {}
        for line in buf.value.decode().splitlines():
            name, calls, ms = line.rsplit(' ', 2)
            out[name] = (int(calls), float(ms))
        return out

    def download_async(self, host_array, dev_ptr, slot):
        """enqueue the D2H copy of host_array.nbytes from dev_ptr into the (page-locked) host_array"""
        check(self.lib.gpa_download_async(self.handle, _ptr(host_array), _ptr(int(dev_ptr)), host_array.nbytes, int(slot)),
              'gpa_download_async')

    def download_wait(self, slot):
        check(self.lib.gpa_download_wait(self.handle, int(slot)), 'gpa_download_wait')

    def last_stage_ms(self):
        ms = (C.c_float * 5)()
        check(self.lib.gpa_last_stage_ms(self.handle, ms), 'gpa_last_stage_ms')
        return [ms[i] for i in range(5)]


class DeviceBuffer:
    """A raw device allocation through the HIP runtime (hipMalloc / hipMemcpy via ctypes) for
    callers of the `_dev` / `_async` entry points that do not want torch for it."""
    _hip = None

    def __init__(self, nbytes, device=0):
        """`device`: the GPU the memory lives on.  hipMalloc / hipMemcpy act on the calling THREAD's current device, so
        every method selects it first (a plan on GPU 1 used from a thread whose current device is 0 -- the default of
        every new thread, e.g. the upload / download workers of the stack call -- would otherwise allocate on GPU 0
        and hand the plan's kernels a pointer they cannot reach)."""
        load()
        if DeviceBuffer._hip is None:
            DeviceBuffer._hip = C.CDLL('libamdhip64.so')   # already mapped as a dependency of libgpa_hip.so
        self.device = int(device)
        self._select()
        p = C.c_void_p()
        rc = self._hip.hipMalloc(C.byref(p), C.c_size_t(int(nbytes)))
        if rc != 0 or not p.value:
            raise GPAError('hipMalloc(%d) on device %d failed (%d)' % (nbytes, self.device, rc))
        self.ptr, self.nbytes = p.value, int(nbytes)

    def _select(self):
        rc = self._hip.hipSetDevice(C.c_int(self.device))
        if rc != 0:
            raise GPAError('hipSetDevice(%d) failed (%d)' % (self.device, rc))

    def upload(self, host):
        host = np.ascontiguousarray(host)
        self._select()
        if self._hip.hipMemcpy(C.c_void_p(self.ptr), host.ctypes.data_as(C.c_void_p), C.c_size_t(host.nbytes), 1) != 0:
            raise GPAError('hipMemcpy H2D failed')

    def upload_at(self, host, byte_offset):
        """H2D of `host` to ptr + byte_offset (a large buffer filled piece by piece)"""
        host = np.ascontiguousarray(host)
        if byte_offset < 0 or byte_offset + host.nbytes > self.nbytes:
            raise ValueError('upload_at: outside the buffer')
        self._select()
        if self._hip.hipMemcpy(C.c_void_p(self.ptr + int(byte_offset)), host.ctypes.data_as(C.c_void_p), C.c_size_t(host.nbytes), 1) != 0:
            raise GPAError('hipMemcpy H2D failed')

    def download(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self._select()
        if self._hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(out.nbytes), 2) != 0:
            raise GPAError('hipMemcpy D2H failed')
        return out

    def download_into(self, out):
        """copy the first out.nbytes bytes into the (C-contiguous) array `out`"""
        if not out.flags.c_contiguous:
            raise ValueError('destination must be C-contiguous')
        self._select()
        if self._hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(out.nbytes), 2) != 0:
            raise GPAError('hipMemcpy D2H failed')

    def free(self):
        if getattr(self, 'ptr', None):
            self._hip.hipSetDevice(C.c_int(self.device))
            self._hip.hipFree(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float64):
    """NumPy array in page-locked host memory (hipHostMalloc).  The host-pointer entry points copy
    from / to such arrays at full PCIe rate (about 4x the rate of pageable memory); use it for images
    and pass it as `out=` where offered.  Freed when the array (and every view of it) is collected."""
    import weakref
    load()
    if DeviceBuffer._hip is None:
        DeviceBuffer._hip = C.CDLL('libamdhip64.so')
    hip = DeviceBuffer._hip
    dtype = np.dtype(dtype)
    nbytes = max(int(np.prod(shape)) * dtype.itemsize, 1)
    ptr = C.c_void_p()
    rc = hip.hipHostMalloc(C.byref(ptr), C.c_size_t(nbytes), C.c_uint(0))
    if rc != 0 or not ptr.value:
        raise GPAError('hipHostMalloc(%d) failed (%d)' % (nbytes, rc))
    buf = (C.c_char * nbytes).from_address(ptr.value)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    weakref.finalize(buf, hip.hipHostFree, C.c_void_p(ptr.value))
    return arr


# small plan cache so the drop-in functions do not rebuild tables on every call
_plans = {}


def get_plan(shape, batch, dtype=np.float64, device=0):
    key = (int(shape[0]), int(shape[1]), np.dtype(dtype).name, int(device))
    p = _plans.get(key)
    if p is None or p.max_batch < batch:
        # evicted plans are only dropped from the cache: a caller may still hold one, and
        # Plan.__del__ releases the device memory with the last reference
        if len(_plans) >= 4 and key not in _plans:
            _plans.pop(next(iter(_plans)))
        p = Plan(shape, max(int(batch), 1), dtype, device)
        _plans[key] = p
    return p


def props_from_jac(jac, add_identity=False, refangle=0., refscale=1., diff=False, dtype=np.float64, device=0):
    """Plan-free f-2 entry: jac (..., 2, 2) -> props (4, ...)."""
    dtype = np.dtype(dtype)
    if dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise ValueError('dtype must be float32 or float64')
    jac = np.ascontiguousarray(jac, dtype=dtype)
    if jac.shape[-2:] != (2, 2):
        raise ValueError('jac must have shape (..., 2, 2)')
    lead = jac.shape[:-2]
    props = np.empty((4,) + lead, dtype=dtype)
    check(load().gpa_props_from_jac(int(device), 0 if dtype == np.float32 else 1, jac.size // 4, _ptr(jac),
                                    int(bool(add_identity)), float(refangle), float(refscale), int(bool(diff)),
                                    _ptr(props)), 'gpa_props_from_jac')
    return props


@atexit.register
def _close_plans():
    for p in list(_plans.values()):
        try:
            p.close()
        except Exception:
            pass
    _plans.clear()
