"""Host-side mirror of the reference's AGNetwork interface for the ResnetPV path (include/alphagomoku/networks/
AGNetwork.hpp:60-98): the subset the self-play path uses — construct from a config, load weights, forward a batch.
All arithmetic happens in libagx.so (HIP); this class only owns handles and device buffers."""
import ctypes
import numpy as np

from ._lib import lib, check, AgxNetDesc


class DeviceBuffer:
    def __init__(self, nbytes):
        self.ptr = ctypes.c_void_p()
        self.nbytes = nbytes
        check(lib.agx_malloc(ctypes.byref(self.ptr), max(nbytes, 16)))

    def upload(self, array):
        a = np.ascontiguousarray(array)
        assert a.nbytes <= self.nbytes
        check(lib.agx_memcpy_h2d(self.ptr, a.ctypes.data_as(ctypes.c_void_p), a.nbytes))

    def download(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(lib.agx_memcpy_d2h(out.ctypes.data_as(ctypes.c_void_p), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib.agx_free(self.ptr)
            self.ptr = ctypes.c_void_p()


class AGNetwork:
    """ResnetPV on the device.  `desc` is a dict as made by synthetic.net_desc()."""

    def __init__(self, desc):
        self.desc = dict(desc)
        self._cdesc = AgxNetDesc(desc["rows"], desc["cols"], desc["blocks"], desc["filters"],
                                 desc["in_channels"], desc["value_hidden"], desc.get("action_values", 0))
        self._net = ctypes.c_void_p()
        check(lib.agx_net_create(ctypes.byref(self._cdesc), ctypes.byref(self._net)))

    def blobFloats(self):
        return int(lib.agx_net_blob_floats(ctypes.byref(self._cdesc)))

    def loadWeights(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        check(lib.agx_net_load_weights(self._net, blob.ctypes.data_as(ctypes.c_void_p), blob.size))

    def forwardDevice(self, d_features, batch, d_policy, d_value, stream=None, d_action_values=None):
        if d_action_values is None:
            check(lib.agx_nn_forward(self._net, d_features, batch, d_policy, d_value, stream))
        else:
            check(lib.agx_nn_forward_pvq(self._net, d_features, batch, d_policy, d_value, d_action_values, stream))

    def forward(self, features):
        """Convenience host round trip (tests): features uint32 [B, HW] -> (policy [B, HW], value [B, 3]) and, for a network
        with the action-values head, additionally q [B, HW, 2] = (win, draw) per cell."""
        features = np.ascontiguousarray(features, dtype=np.uint32)
        batch, hw = features.shape
        with_q = bool(self.desc.get("action_values", 0))
        f = DeviceBuffer(features.nbytes)
        p = DeviceBuffer(batch * hw * 4)
        v = DeviceBuffer(batch * 3 * 4)
        q = DeviceBuffer(batch * hw * 2 * 4) if with_q else None
        try:
            f.upload(features)
            self.forwardDevice(f.ptr, batch, p.ptr, v.ptr, None, q.ptr if with_q else None)
            check(lib.agx_device_synchronize())
            out = (p.download((batch, hw), np.float32), v.download((batch, 3), np.float32))
            if with_q:
                out = out + (q.download((batch, hw, 2), np.float32),)
            return out
        finally:
            f.free(); p.free(); v.free()
            if q is not None:
                q.free()

    def close(self):
        if self._net:
            lib.agx_net_destroy(self._net)
            self._net = ctypes.c_void_p()
