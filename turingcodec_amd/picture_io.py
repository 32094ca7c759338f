"""Host side of the device picture store: raw planar YUV files in -- the reference's on-disk input format
(turing/encode.cpp:600-640: frames of Y, U, V planes, 4:2:0, 8-bit bytes or 16-bit little-endian words, no header) -- and the
`havoc_mi355x_picture_*` entry points (include/havoc_mi355x.h) as a small class.  No pixel arithmetic happens here: the frame
bytes go to the device as they are; widening, placement in the padded layout, border replication and the fractional-sample
planes are kernels."""
import ctypes as C
import os

import numpy as np


class YuvReader:
    """frames of a raw planar 4:2:0 file: reader[i] / iteration -> the frame's bytes (a numpy uint8 view of the file, no copy)"""

    def __init__(self, path, width, height, bit_depth=8):
        if width % 2 or height % 2:
            raise ValueError("4:2:0 pictures have even dimensions")
        self.path, self.width, self.height, self.bit_depth = path, width, height, bit_depth
        self.sample_bytes = 1 if bit_depth == 8 else 2
        self.frame_bytes = width * height * 3 // 2 * self.sample_bytes
        size = os.path.getsize(path)
        if size < self.frame_bytes:
            raise ValueError(f"{path}: {size} bytes is less than one {width}x{height} frame ({self.frame_bytes})")
        self.frames = size // self.frame_bytes      # a trailing partial frame is ignored, as the reference's reader stops at it
        self._map = np.memmap(path, np.uint8, "r")

    def __len__(self):
        return self.frames

    def __getitem__(self, i):
        if not 0 <= i < self.frames:
            raise IndexError(i)
        return self._map[i * self.frame_bytes:(i + 1) * self.frame_bytes]

    def __iter__(self):
        return (self[i] for i in range(self.frames))

    def planes(self, i):
        """(Y, U, V) numpy views of frame i, for checks on the host"""
        dt = np.uint8 if self.sample_bytes == 1 else np.dtype("<u2")
        a = np.frombuffer(self[i], dt)
        n, c = self.width * self.height, self.width * self.height // 4
        return (a[:n].reshape(self.height, self.width), a[n:n + c].reshape(self.height // 2, self.width // 2),
                a[n + c:].reshape(self.height // 2, self.width // 2))


class DevicePicture:
    """one picture of the store: three planes in the reference's padded layout (turing/Picture.cpp:91-125), resident in HBM"""

    def __init__(self, hv, width, height, bit_depth=8, pad=96, alignment=64):
        self.hv, self.width, self.height, self.bit_depth = hv, width, height, bit_depth
        self.S = 1 if bit_depth == 8 else 2
        self.pic = C.c_void_p()
        hv._ck(hv.L.havoc_mi355x_picture_create(hv.h, self.S, bit_depth, width, height, pad, alignment, C.byref(self.pic)))

    def close(self):
        if self.pic:
            self.hv.L.havoc_mi355x_picture_destroy(self.hv.h, self.pic)
            self.pic = C.c_void_p()

    def upload(self, frame_bytes, src_bit_depth=None, pad=True):
        """frame_bytes: one frame as YuvReader returns it; an 8-bit file on a 16-bit picture is stored << (bit_depth - 8), as the
        reference does (turing/encode.cpp:397)"""
        src_bd = self.bit_depth if src_bit_depth is None else src_bit_depth
        src_S = 1 if src_bd == 8 else 2
        buf = np.ascontiguousarray(frame_bytes)
        if buf.nbytes != self.width * self.height * 3 // 2 * src_S:
            raise ValueError("frame size does not match the picture")
        self.hv._ck(self.hv.L.havoc_mi355x_picture_upload_yuv(self.hv.h, self.pic, buf.ctypes.data, src_S, self.bit_depth - src_bd if src_S == 1 else 0, int(pad)))

    def pad(self):
        self.hv._ck(self.hv.L.havoc_mi355x_picture_pad(self.hv.h, self.pic))

    def plane(self, c):
        """(device base pointer, sample offset of (0, 0), stride, width, height, pad) of plane c"""
        base, org, st, w, h, pd = C.c_void_p(), C.c_int64(), C.c_ssize_t(), C.c_int(), C.c_int(), C.c_int()
        self.hv._ck(self.hv.L.havoc_mi355x_picture_plane(self.pic, c, C.byref(base), C.byref(org), C.byref(st), C.byref(w), C.byref(h), C.byref(pd)))
        return base.value, org.value, st.value, w.value, h.value, pd.value

    def download(self, c, with_padding=False):
        _, _, _, w, h, pd = self.plane(c)
        p = pd if with_padding else 0
        a = np.zeros((h + 2 * p, w + 2 * p), np.uint8 if self.S == 1 else np.uint16)
        self.hv._ck(self.hv.L.havoc_mi355x_picture_download_plane(self.hv.h, self.pic, c, a.ctypes.data + (p * a.shape[1] + p) * a.itemsize, a.shape[1],
                                                                  int(with_padding)))
        return a

    def phase_planes(self):
        """interpolates the 15 fractional-sample luma planes (the picture becomes a reference); -> (device pointer, plane elements, origin)"""
        base, elems, org = C.c_void_p(), C.c_ssize_t(), C.c_int64()
        self.hv._ck(self.hv.L.havoc_mi355x_picture_phase_planes(self.hv.h, self.pic, C.byref(base), C.byref(elems), C.byref(org)))
        return base.value, elems.value, org.value
