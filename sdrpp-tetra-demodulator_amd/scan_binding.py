"""ctypes binding of the batched training-sequence search (include/tetra_burst_scan.h)."""
import ctypes as C

import numpy as np

from .binding import TetraDemodError, load_library

SCAN_EXPORTS = ["tetra_find_train_seq_batch_device", "tetra_find_train_seq_batch"]
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
