"""ctypes binding of the batched training-sequence search (include/tetra_burst_scan.h)."""
import ctypes as C

import numpy as np

from .binding import TetraDemodError, load_library

SCAN_EXPORTS = ["tetra_find_train_seq_batch_device", "tetra_find_train_seq_batch", "tetra_ts_indicator_create",
                "tetra_ts_indicator_reset", "tetra_ts_indicator_process_device", "tetra_ts_indicator_process"]
TRAIN_NORM_1, TRAIN_NORM_2, TRAIN_NORM_3, TRAIN_SYNC, TRAIN_EXT = 0, 1, 2, 3, 4
ALL_MASK = 0x1f

_ready = False


def _lib():
    global _ready
    L = load_library()
    if not _ready:
        vp, i32 = C.c_void_p, C.c_int
        L.tetra_find_train_seq_batch_device.argtypes = [vp, i32, i32, vp, C.c_uint32, vp, vp, vp]
        L.tetra_find_train_seq_batch.argtypes = [vp, i32, i32, vp, C.c_uint32, vp, vp, i32]
        L.tetra_ts_indicator_create.argtypes = [i32, i32, C.POINTER(vp)]
        L.tetra_ts_indicator_destroy.argtypes = [vp]
        L.tetra_ts_indicator_destroy.restype = None
        L.tetra_ts_indicator_reset.argtypes = [vp, i32]
        L.tetra_ts_indicator_process_device.argtypes = [vp, vp, i32, vp, vp, vp, vp]
        L.tetra_ts_indicator_process.argtypes = [vp, vp, i32, vp, vp, vp]
        for n in SCAN_EXPORTS:
            getattr(L, n).restype = i32
        _ready = True
    return L


def find_train_seq_batch(bits, end_of_in, mask=ALL_MASK, device=-1):
    """bits uint8 [C][stride] (stride % 4 == 0), end_of_in int32 [C] -> (type int32 [C], offset int32 [C])."""
    bits = np.ascontiguousarray(bits, np.uint8)
    end = np.ascontiguousarray(end_of_in, np.int32)
    Cn, stride = bits.shape
    t = np.zeros(Cn, np.int32)
    o = np.zeros(Cn, np.int32)
    rc = _lib().tetra_find_train_seq_batch(bits.ctypes.data_as(C.c_void_p), Cn, stride, end.ctypes.data_as(C.c_void_p),
                                           int(mask), t.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), device)
    if rc:
        raise TetraDemodError(rc, "tetra_find_train_seq_batch")
    return t, o


def find_train_seq_batch_device(d_bits, n_channels, bits_stride, d_end, mask, d_type, d_offset, stream=None):
    s = None
    if stream is not None:
        s = C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))
    rc = _lib().tetra_find_train_seq_batch_device(C.c_void_p(d_bits.data_ptr()), int(n_channels), int(bits_stride),
                                                  C.c_void_p(d_end.data_ptr()), int(mask), C.c_void_p(d_type.data_ptr()),
                                                  C.c_void_p(d_offset.data_ptr()), s)
    if rc:
        raise TetraDemodError(rc, "tetra_find_train_seq_batch_device")


class TsIndicator:
    """The plugin's training-sequence indicator (src/main.cpp:385-414) for C channels on one GPU, state carried."""

    def __init__(self, n_channels, device=-1):
        self._h = C.c_void_p()
        rc = _lib().tetra_ts_indicator_create(int(n_channels), int(device), C.byref(self._h))
        if rc:
            raise TetraDemodError(rc, "tetra_ts_indicator_create")
        self.n_channels = int(n_channels)

    def close(self):
        if self._h:
            _lib().tetra_ts_indicator_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, channel=-1):
        rc = _lib().tetra_ts_indicator_reset(self._h, int(channel))
        if rc:
            raise TetraDemodError(rc, "tetra_ts_indicator_reset")

    def process(self, bits, n_bits):
        """bits uint8 [C][stride] (stride % 4 == 0), n_bits int32 [C] -> (found bool [C], expire int32 [C])."""
        b = np.ascontiguousarray(bits, np.uint8)
        nb = np.ascontiguousarray(n_bits, np.int32)
        assert b.shape[0] == self.n_channels and nb.shape[0] == self.n_channels
        found = np.zeros(self.n_channels, np.uint8)
        expire = np.zeros(self.n_channels, np.int32)
        vp = C.c_void_p
        rc = _lib().tetra_ts_indicator_process(self._h, b.ctypes.data_as(vp), b.shape[1], nb.ctypes.data_as(vp),
                                               found.ctypes.data_as(vp), expire.ctypes.data_as(vp))
        if rc:
            raise TetraDemodError(rc, "tetra_ts_indicator_process")
        return found.astype(bool), expire

    def process_device(self, d_bits, bits_stride, d_n_bits, d_found, d_expire=None, stream=None):
        s = None
        if stream is not None:
            s = C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))
        vp = C.c_void_p
        rc = _lib().tetra_ts_indicator_process_device(self._h, vp(d_bits.data_ptr()), int(bits_stride), vp(d_n_bits.data_ptr()),
                                                      vp(d_found.data_ptr()), None if d_expire is None else vp(d_expire.data_ptr()), s)
        if rc:
            raise TetraDemodError(rc, "tetra_ts_indicator_process_device")
