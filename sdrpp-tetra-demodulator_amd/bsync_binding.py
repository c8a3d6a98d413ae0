"""ctypes binding of the batched burst synchroniser + burst demultiplexer (include/tetra_burst_sync.h)."""
import ctypes as C

import numpy as np

from .binding import TetraDemodError, load_library

BSYNC_EXPORTS = ["tetra_bsync_create", "tetra_bsync_destroy", "tetra_bsync_reset", "tetra_bsync_max_frames",
                 "tetra_bsync_process_device", "tetra_bsync_process", "tetra_bsync_get_state", "tetra_burst_demux_device",
                 "tetra_burst_demux_compact_device", "tetra_bsync_process_packed_device", "tetra_burst_demux_packed_device",
                 "tetra_burst_demux_compact_packed_device", "tetra_burst_index_device"]
LIST_SYNC, LIST_NORM_1, LIST_NORM_2, LIST_ANY, N_LISTS = 0, 1, 2, 3, 4
FRAME_WORDS = 16
RX_S_UNLOCKED, RX_S_KNOW_FSTART, RX_S_LOCKED = 0, 1, 2
FRAME_STRIDE, FRAME_NONE, BITS_PER_TS = 512, -2, 510


class BsyncState(C.Structure):
    _fields_ = [("state", C.c_int32), ("bits_in_buf", C.c_uint32), ("bitbuf_start_bitnum", C.c_uint32),
                ("next_frame_start_bitnum", C.c_uint32)]


_ready = False


def _lib():
    global _ready
    L = load_library()
    if not _ready:
        vp, i32 = C.c_void_p, C.c_int
        L.tetra_bsync_create.argtypes = [i32, i32, i32, C.POINTER(vp)]
        L.tetra_bsync_destroy.argtypes = [vp]
        L.tetra_bsync_reset.argtypes = [vp]
        L.tetra_bsync_max_frames.argtypes = [vp]
        L.tetra_bsync_process_device.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp]
        L.tetra_bsync_process.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp]
        L.tetra_bsync_get_state.argtypes = [vp, i32, i32, vp]
        L.tetra_burst_demux_device.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp]
        L.tetra_burst_demux_compact_device.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp, vp]
        L.tetra_bsync_process_packed_device.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp]
        L.tetra_burst_demux_packed_device.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp]
        L.tetra_burst_demux_compact_packed_device.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp, vp]
        L.tetra_burst_index_device.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
        for n in BSYNC_EXPORTS:
            getattr(L, n).restype = i32
        _ready = True
    return L


def _stream(stream):
    if stream is None:
        return None
    return C.c_void_p(stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream))


class BurstSync:
    """C independent tetra_rx_state receivers on one GPU."""

    def __init__(self, n_channels, max_bits, device=-1):
        self._h = C.c_void_p()
        rc = _lib().tetra_bsync_create(int(n_channels), int(max_bits), int(device), C.byref(self._h))
        if rc:
            raise TetraDemodError(rc, "tetra_bsync_create")
        self.n_channels, self.max_bits = int(n_channels), int(max_bits)
        self.max_frames = _lib().tetra_bsync_max_frames(self._h)

    def close(self):
        if self._h:
            _lib().tetra_bsync_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        rc = _lib().tetra_bsync_reset(self._h)
        if rc:
            raise TetraDemodError(rc, "tetra_bsync_reset")

    def process(self, bits, n_bits):
        """bits uint8 [C][stride] (stride % 4 == 0), n_bits int32 [C] -> (frames uint8 [C][F][512], frame_type int32 [C][F],
        frame_bitnum uint32 [C][F], n_frames int32 [C])."""
        b = np.ascontiguousarray(bits, np.uint8)
        nb = np.ascontiguousarray(n_bits, np.int32)
        Cn, F = self.n_channels, self.max_frames
        assert b.shape[0] == Cn and nb.shape[0] == Cn
        frames = np.zeros((Cn, F, FRAME_STRIDE), np.uint8)
        ft = np.zeros((Cn, F), np.int32)
        fb = np.zeros((Cn, F), np.uint32)
        nf = np.zeros(Cn, np.int32)
        vp = C.c_void_p
        rc = _lib().tetra_bsync_process(self._h, b.ctypes.data_as(vp), b.shape[1], nb.ctypes.data_as(vp), frames.ctypes.data_as(vp),
                                        ft.ctypes.data_as(vp), fb.ctypes.data_as(vp), nf.ctypes.data_as(vp))
        if rc:
            raise TetraDemodError(rc, "tetra_bsync_process")
        return frames, ft, fb, nf

    def process_device(self, d_bits, bits_stride, d_n_bits, d_frames, d_frame_type, d_frame_bitnum, d_n_frames, stream=None):
        vp = C.c_void_p
        rc = _lib().tetra_bsync_process_device(self._h, vp(d_bits.data_ptr()), int(bits_stride), vp(d_n_bits.data_ptr()),
                                               vp(d_frames.data_ptr()), vp(d_frame_type.data_ptr()), vp(d_frame_bitnum.data_ptr()),
                                               vp(d_n_frames.data_ptr()), _stream(stream))
        if rc:
            raise TetraDemodError(rc, "tetra_bsync_process_device")

    def process_packed_device(self, d_bits, bits_stride, d_n_bits, d_frames_packed, d_frame_type, d_frame_bitnum, d_n_frames, stream=None):
        """Frames as [C][max_frames][16] int32 / uint32 words (first bit = most significant) instead of [C][max_frames][512] bytes."""
        vp = C.c_void_p
        rc = _lib().tetra_bsync_process_packed_device(self._h, vp(d_bits.data_ptr()), int(bits_stride), vp(d_n_bits.data_ptr()),
                                                      vp(d_frames_packed.data_ptr()), vp(d_frame_type.data_ptr()),
                                                      vp(d_frame_bitnum.data_ptr()), vp(d_n_frames.data_ptr()), _stream(stream))
        if rc:
            raise TetraDemodError(rc, "tetra_bsync_process_packed_device")

    def states(self, first=0, count=None):
        count = self.n_channels - first if count is None else count
        arr = (BsyncState * count)()
        rc = _lib().tetra_bsync_get_state(self._h, int(first), int(count), arr)
        if rc:
            raise TetraDemodError(rc, "tetra_bsync_get_state")
        return [(s.state, s.bits_in_buf, s.bitbuf_start_bitnum, s.next_frame_start_bitnum) for s in arr]


def demux_compact_device(d_frames, d_frame_type, n, tpsap, blk_num, d_rows, row_stride, d_row_frame, d_n_rows, stream=None, packed=False):
    vp = C.c_void_p
    fn = _lib().tetra_burst_demux_compact_packed_device if packed else _lib().tetra_burst_demux_compact_device
    rc = fn(vp(d_frames.data_ptr()), vp(d_frame_type.data_ptr()), int(n), int(tpsap), int(blk_num),
                                                 vp(d_rows.data_ptr()), int(row_stride), vp(d_row_frame.data_ptr()),
                                                 vp(d_n_rows.data_ptr()), _stream(stream))
    if rc:
        raise TetraDemodError(rc, "tetra_burst_demux_compact_device")


def demux_device(d_frames, d_frame_type, n, tpsap, blk_num, d_rows, row_stride, d_valid, stream=None, packed=False):
    vp = C.c_void_p
    fn = _lib().tetra_burst_demux_packed_device if packed else _lib().tetra_burst_demux_device
    rc = fn(vp(d_frames.data_ptr()), vp(d_frame_type.data_ptr()), int(n), int(tpsap), int(blk_num),
                                         vp(d_rows.data_ptr()), int(row_stride), vp(d_valid.data_ptr()), _stream(stream))
    if rc:
        raise TetraDemodError(rc, "tetra_burst_demux_device")


def index_device(d_frame_type, frames_per_channel, d_lists, d_counts, d_chan_first=None, d_work=None, stream=None):
    """tetra_burst_index_device: d_lists [4][n], d_counts [4], d_chan_first [4][n / frames_per_channel] or None (int32 tensors)."""
    import torch
    n = int(d_frame_type.numel())
    if d_work is None:
        d_work = torch.empty(N_LISTS * ((n + 255) // 256) + 1, dtype=torch.int32, device=d_frame_type.device)
    vp = C.c_void_p
    rc = _lib().tetra_burst_index_device(vp(d_frame_type.data_ptr()), n, int(frames_per_channel), vp(d_lists.data_ptr()), vp(d_counts.data_ptr()),  # (empty tensors: NULL)
                                         None if d_chan_first is None else vp(d_chan_first.data_ptr()), vp(d_work.data_ptr()), _stream(stream))
    if rc:
        raise TetraDemodError(rc, "tetra_burst_index_device")
