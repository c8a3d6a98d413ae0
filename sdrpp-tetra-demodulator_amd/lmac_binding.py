"""ctypes binding of the batched lower-MAC channel decoding (include/tetra_lmac.h)."""
import ctypes as C

import numpy as np

from .binding import TetraDemodError, load_library

LMAC_EXPORTS = ["tetra_lmac_blk_param", "tetra_lmac_scramb_init", "tetra_lmac_decode_batch_device", "tetra_lmac_decode_batch",
                "tetra_lmac_track_scramb_device", "tetra_lmac_decode_counted_device", "tetra_lmac_track_sync_device",
                "tetra_lmac_debug_force_byte_route", "tetra_lmac_decode_frames_device", "tetra_lmac_track_sync_lists_device",
                "tetra_lmac_decode_frames_workspace_bytes"]
# enum tp_sap_data_type (src/decoder/src/phy/tetra_burst.h:9-16)
TPSAP_T_SB1, TPSAP_T_SB2, TPSAP_T_NDB, TPSAP_T_BBK, TPSAP_T_SCH_HU, TPSAP_T_SCH_F = range(6)


class Label(C.Structure):
    """tetra_lmac_label_t (= tetra_rx_block_t)."""
    _fields_ = [("channel", C.c_int32), ("frame_slot", C.c_int32), ("bitnum", C.c_uint32), ("tdma_time_rx", C.c_uint32),
                ("tdma_time", C.c_uint32), ("crc_ok", C.c_int32)]


class Frames(C.Structure):
    """tetra_lmac_frames_t."""
    _fields_ = [("d_frames", C.c_void_p), ("d_frame_type", C.c_void_p), ("n_frames", C.c_int32), ("frames_per_channel", C.c_int32),
                ("d_frame_bitnum", C.c_void_p), ("d_time_rx", C.c_void_p), ("d_time", C.c_void_p), ("d_workspace", C.c_void_p),
                ("workspace_bytes", C.c_size_t)]


class Job(C.Structure):
    """tetra_lmac_job_t."""
    _fields_ = [("type", C.c_int32), ("blk_num", C.c_int32), ("d_row_frame", C.c_void_p), ("d_n_rows", C.c_void_p), ("max_rows", C.c_int32),
                ("out_stride", C.c_int32), ("d_frame_scramb", C.c_void_p), ("d_type2", C.c_void_p), ("d_crc_ok", C.c_void_p),
                ("d_labels", C.c_void_p)]


class BlkParam(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("type345_bits", "type2_bits", "type1_bits", "interleave_a", "have_crc16")]


_ready = False


def _lib():
    global _ready
    L = load_library()
    if not _ready:
        vp, i32 = C.c_void_p, C.c_int
        L.tetra_lmac_blk_param.argtypes = [i32, C.POINTER(BlkParam)]
        L.tetra_lmac_blk_param.restype = i32
        L.tetra_lmac_scramb_init.argtypes = [C.c_uint16, C.c_uint16, C.c_uint8]
        L.tetra_lmac_scramb_init.restype = C.c_uint32
        L.tetra_lmac_decode_batch_device.argtypes = [i32, vp, i32, i32, vp, vp, i32, vp, vp]
        L.tetra_lmac_decode_batch_device.restype = i32
        L.tetra_lmac_decode_counted_device.argtypes = [i32, vp, i32, vp, i32, vp, vp, vp, i32, vp, vp]
        L.tetra_lmac_decode_counted_device.restype = i32
        L.tetra_lmac_decode_batch.argtypes = [i32, vp, i32, i32, vp, vp, i32, vp, i32]
        L.tetra_lmac_decode_batch.restype = i32
        L.tetra_lmac_track_scramb_device.argtypes = [vp, i32, vp, vp, i32, i32, vp, vp, vp]
        L.tetra_lmac_track_scramb_device.restype = i32
        _ready = True
    return L


def force_byte_route(on):
    """tetra_lmac_debug_force_byte_route: every row through the decoder's byte route (process-wide).  Returns the old setting."""
    L = _lib()
    L.tetra_lmac_debug_force_byte_route.argtypes = [C.c_int]
    L.tetra_lmac_debug_force_byte_route.restype = C.c_int
    return bool(L.tetra_lmac_debug_force_byte_route(int(bool(on))))


def blk_param(blk_type):
    p = BlkParam()
    rc = _lib().tetra_lmac_blk_param(int(blk_type), C.byref(p))
    if rc:
        raise TetraDemodError(rc, "tetra_lmac_blk_param")
    return p


def scramb_init(mcc, mnc, colour):
    return int(_lib().tetra_lmac_scramb_init(int(mcc) & 0xffff, int(mnc) & 0xffff, int(colour) & 0xff))


def out_stride_for(blk_type):
    return (blk_param(blk_type).type2_bits + 3) & ~3


def decode_batch(blk_type, type5, scramb=None, device=-1):
    """type5 uint8 [n][in_stride] (in_stride % 4 == 0), scramb uint32 [n] (None for SB1)
    -> (type2 uint8 [n][type2_bits rounded up to 4], crc_ok int32 [n])."""
    rows = np.ascontiguousarray(type5, np.uint8)
    n, in_stride = rows.shape
    ost = out_stride_for(blk_type)
    out = np.zeros((n, ost), np.uint8)
    ok = np.zeros(n, np.int32)
    si = None if scramb is None else np.ascontiguousarray(scramb, np.uint32)
    rc = _lib().tetra_lmac_decode_batch(int(blk_type), rows.ctypes.data_as(C.c_void_p), n, in_stride,
                                        None if si is None else si.ctypes.data_as(C.c_void_p),
                                        out.ctypes.data_as(C.c_void_p), ost, ok.ctypes.data_as(C.c_void_p), device)
    if rc:
        raise TetraDemodError(rc, "tetra_lmac_decode_batch")
    return out, ok


def decode_batch_device(blk_type, d_type5, n_blocks, in_stride, d_scramb, d_type2, out_stride, d_crc_ok, stream=None):
    """torch tensors already on the GPU; enqueues on `stream` (torch stream or raw handle), no synchronisation."""
    s = None
    if stream is not None:
        s = C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))
    rc = _lib().tetra_lmac_decode_batch_device(int(blk_type), C.c_void_p(d_type5.data_ptr()), int(n_blocks), int(in_stride),
                                               None if d_scramb is None else C.c_void_p(d_scramb.data_ptr()),
                                               C.c_void_p(d_type2.data_ptr()), int(out_stride), C.c_void_p(d_crc_ok.data_ptr()), s)
    if rc:
        raise TetraDemodError(rc, "tetra_lmac_decode_batch_device")


def decode_counted_device(blk_type, d_type5, capacity, d_n_blocks, in_stride, d_scramb, d_init_index, d_type2, out_stride, d_crc_ok,
                          stream=None):
    """Rows from the compacting demultiplexer: count and scrambling-code index are read on the device."""
    s = None
    if stream is not None:
        s = C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))
    vp = C.c_void_p
    rc = _lib().tetra_lmac_decode_counted_device(int(blk_type), vp(d_type5.data_ptr()), int(capacity),
                                                 None if d_n_blocks is None else vp(d_n_blocks.data_ptr()), int(in_stride),
                                                 None if d_scramb is None else vp(d_scramb.data_ptr()),
                                                 None if d_init_index is None else vp(d_init_index.data_ptr()),
                                                 vp(d_type2.data_ptr()), int(out_stride), vp(d_crc_ok.data_ptr()), s)
    if rc:
        raise TetraDemodError(rc, "tetra_lmac_decode_counted_device")


def track_scramb_device(d_sb1_type2, type2_stride, d_crc_ok, d_valid, n_channels, frames_per_channel, d_chan_scramb, d_row_scramb,
                        stream=None):
    s = None
    if stream is not None:
        s = C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))
    vp = C.c_void_p
    rc = _lib().tetra_lmac_track_scramb_device(vp(d_sb1_type2.data_ptr()), int(type2_stride), vp(d_crc_ok.data_ptr()),
                                               vp(d_valid.data_ptr()), int(n_channels), int(frames_per_channel),
                                               vp(d_chan_scramb.data_ptr()), vp(d_row_scramb.data_ptr()), s)
    if rc:
        raise TetraDemodError(rc, "tetra_lmac_track_scramb_device")


def track_sync_device(d_sb1_type2, type2_stride, d_crc_ok, d_valid, d_n_frames, n_channels, frames_per_channel, d_cell, d_row_scramb,
                      d_row_time_rx=None, d_row_time=None, stream=None):
    """tetra_lmac_track_sync_device: d_cell = torch int32 / uint32 tensor [n_channels][10] (tetra_lmac_cell_state_t: scramb_init,
    colour_code, mcc, mnc, tcd tn / fn / mn, phy tn / fn / mn)."""
    s = None
    if stream is not None:
        s = C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))
    vp = C.c_void_p

    def p(t):
        return vp(t.data_ptr()) if t is not None else None
    L = _lib()
    L.tetra_lmac_track_sync_device.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.tetra_lmac_track_sync_device.restype = C.c_int
    rc = L.tetra_lmac_track_sync_device(p(d_sb1_type2), int(type2_stride), p(d_crc_ok), p(d_valid), p(d_n_frames), int(n_channels),
                                        int(frames_per_channel), p(d_cell), p(d_row_scramb), p(d_row_time_rx), p(d_row_time), s)
    if rc:
        raise TetraDemodError(rc, "tetra_lmac_track_sync_device")


def _ptr(t):
    return None if t is None else t.data_ptr()


def _job_array(jobs):
    arr = (Job * max(1, len(jobs)))()
    for i, j in enumerate(jobs):
        arr[i] = Job(int(j["type"]), int(j.get("blk_num", 0)), _ptr(j.get("row_frame")), _ptr(j.get("n_rows")), int(j["max_rows"]),
                     int(j.get("out_stride", 0)), _ptr(j.get("frame_scramb")), _ptr(j.get("type2")), _ptr(j.get("crc_ok")), _ptr(j.get("labels")))
    return arr


def decode_frames_workspace_bytes(jobs):
    """tetra_lmac_decode_frames_workspace_bytes (jobs as for decode_frames_device; only type and max_rows matter)."""
    L = _lib()
    L.tetra_lmac_decode_frames_workspace_bytes.argtypes = [C.POINTER(Job), C.c_int]
    L.tetra_lmac_decode_frames_workspace_bytes.restype = C.c_size_t
    return int(L.tetra_lmac_decode_frames_workspace_bytes(_job_array(jobs), len(jobs)))


def decode_frames_device(d_frames, d_frame_type, jobs, frames_per_channel=0, d_frame_bitnum=None, d_time_rx=None, d_time=None, stream=None,
                         d_workspace=None):
    """tetra_lmac_decode_frames_device.  jobs: dicts with type, blk_num, row_frame, n_rows (tensor or None), max_rows, out_stride,
    frame_scramb (tensor or None), type2, crc_ok, labels (int32 tensor [rows][6] or None); d_workspace: uint8 tensor or None (pool)."""
    L = _lib()
    L.tetra_lmac_decode_frames_device.argtypes = [C.POINTER(Frames), C.POINTER(Job), C.c_int, C.c_void_p]
    L.tetra_lmac_decode_frames_device.restype = C.c_int
    src = Frames(_ptr(d_frames), _ptr(d_frame_type), int(d_frame_type.numel()), int(frames_per_channel), _ptr(d_frame_bitnum), _ptr(d_time_rx),
                 _ptr(d_time), _ptr(d_workspace), 0 if d_workspace is None else int(d_workspace.numel() * d_workspace.element_size()))
    arr = _job_array(jobs)
    s = None
    if stream is not None:
        s = C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))
    rc = L.tetra_lmac_decode_frames_device(C.byref(src), arr, len(jobs), s)
    if rc:
        raise TetraDemodError(rc, "tetra_lmac_decode_frames_device")


def track_sync_lists_device(d_sb1_type2, type2_stride, d_crc_ok, d_frame_type, d_n_frames, d_chan_first_sync, n_channels, frames_per_channel,
                            d_cell, d_row_scramb, d_row_time_rx=None, d_row_time=None, d_frame_bitnum=None, d_sb1_labels=None, stream=None):
    """tetra_lmac_track_sync_lists_device (compact SB1 rows = the SYNC list's)."""
    vp = C.c_void_p
    L = _lib()
    L.tetra_lmac_track_sync_lists_device.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.tetra_lmac_track_sync_lists_device.restype = C.c_int
    s = None
    if stream is not None:
        s = vp(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))
    rc = L.tetra_lmac_track_sync_lists_device(_ptr(d_sb1_type2), int(type2_stride), _ptr(d_crc_ok), _ptr(d_frame_type), _ptr(d_n_frames),
                                              _ptr(d_chan_first_sync), int(n_channels), int(frames_per_channel), _ptr(d_cell), _ptr(d_row_scramb),
                                              _ptr(d_row_time_rx), _ptr(d_row_time), _ptr(d_frame_bitnum), _ptr(d_sb1_labels), s)
    if rc:
        raise TetraDemodError(rc, "tetra_lmac_track_sync_lists_device")
