"""ctypes binding of include/agx.h.  Fails loudly when the HIP library is missing: there is no CPU fallback."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libagx.so")


class AgxError(RuntimeError):
    pass


class AgxNetDesc(ctypes.Structure):
    _fields_ = [("rows", ctypes.c_int), ("cols", ctypes.c_int), ("blocks", ctypes.c_int),
                ("filters", ctypes.c_int), ("in_channels", ctypes.c_int), ("value_hidden", ctypes.c_int)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise AgxError(
            "HIP extension %s is missing — build it with `python -m alphagomoku_amd.build` "
            "(there is no CPU fallback for the product path)" % LIB_PATH)
    return ctypes.CDLL(LIB_PATH)


class _Lazy:
    """Loads libagx.so on first attribute access so that importing the package never touches the GPU."""
    _cdll = None

    def _get(self):
        if _Lazy._cdll is None:
            cdll = _load()
            _declare(cdll)
            _Lazy._cdll = cdll
        return _Lazy._cdll

    def __getattr__(self, name):
        return getattr(self._get(), name)


def _declare(c):
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    c.agx_last_error.restype = ctypes.c_char_p
    c.agx_last_error.argtypes = []
    c.agx_version.restype = ci
    c.agx_set_device.argtypes = [ci]
    c.agx_net_blob_floats.restype = sz
    c.agx_net_blob_floats.argtypes = [ctypes.POINTER(AgxNetDesc)]
    c.agx_net_create.argtypes = [ctypes.POINTER(AgxNetDesc), ctypes.POINTER(vp)]
    c.agx_net_load_weights.argtypes = [vp, vp, sz]
    c.agx_nn_forward.argtypes = [vp, vp, ci, vp, vp, vp]
    c.agx_net_destroy.argtypes = [vp]
    c.agx_malloc.argtypes = [ctypes.POINTER(vp), sz]
    c.agx_free.argtypes = [vp]
    c.agx_memcpy_h2d.argtypes = [vp, vp, sz]
    c.agx_memcpy_d2h.argtypes = [vp, vp, sz]
    c.agx_memset.argtypes = [vp, ci, sz]
    c.agx_device_synchronize.argtypes = []
    c.agx_timer_create.argtypes = [ctypes.POINTER(vp)]
    c.agx_timer_start.argtypes = [vp, vp]
    c.agx_timer_stop.argtypes = [vp, vp]
    c.agx_timer_elapsed_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    c.agx_timer_destroy.argtypes = [vp]


lib = _Lazy()


def check(status):
    if status != 0:
        raise AgxError("agx error %d: %s" % (status, lib.agx_last_error().decode()))
