"""Device-side picture store + input upload through the C ABI: geometry as turing/Picture.cpp:91-125 lays a
Picture<Sample> out, input frames in the on-disk format turing/encode.cpp:600-640 reads (planar Y, U, V; 8-bit bytes or
16-bit little-endian words; 8-bit input on the 16-bit path << 2, encode.cpp:397), border replication (Padding::padImage),
plane up / download, and the phase planes kept with a reference picture."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hv():
    from turingcodec_amd import Havoc
    return Havoc(0)


def geometry(S, width, height, pad, alignment):
    """turing/Picture.cpp:91-125 restated: [(width, height, pad, stride, front)] per plane"""
    out = []
    w, h, p = width, height, pad
    for c in range(3):
        if c == 1:
            w, h, p = w // 2, h // 2, p // 2
        n = alignment // S
        stride = p + w + p
        if stride % n:
            stride += n - stride % n
        front = (n - p % n) if p % n else 0
        out.append((w, h, p, stride, front))
    return out


def plane_info(hv, pic, c):
    base, org, st, w, h, pd = C.c_void_p(), C.c_int64(), C.c_ssize_t(), C.c_int(), C.c_int(), C.c_int()
    assert hv.L.havoc_mi355x_picture_plane(pic, c, C.byref(base), C.byref(org), C.byref(st), C.byref(w), C.byref(h), C.byref(pd)) == 0
    return base.value, org.value, st.value, w.value, h.value, pd.value


def download(hv, pic, c, dtype, with_padding):
    _, _, _, w, h, pd = plane_info(hv, pic, c)
    p = pd if with_padding else 0
    a = np.zeros((h + 2 * p, w + 2 * p), dtype)
    assert hv.L.havoc_mi355x_picture_download_plane(hv.h, pic, c, a.ctypes.data + (p * a.shape[1] + p) * a.itemsize, a.shape[1], int(with_padding)) == 0
    return a


@pytest.mark.parametrize("S,bd,src_S,shift", [(1, 8, 1, 0), (2, 10, 2, 0), (2, 10, 1, 2)])
@pytest.mark.parametrize("width,height,pad,alignment", [(640, 360, 96, 32), (208, 120, 96, 64), (66, 34, 16, 32)])
def test_upload_yuv_layout_and_padding(hv, S, bd, src_S, shift, width, height, pad, alignment):
    rng = np.random.default_rng(width + S)
    pic = C.c_void_p()
    assert hv.L.havoc_mi355x_picture_create(hv.h, S, bd, width, height, pad, alignment, C.byref(pic)) == 0
    try:
        geo = geometry(S, width, height, pad, alignment)
        at = 0
        for c in range(3):
            base, org, st, w, h, pd = plane_info(hv, pic, c)
            gw, gh, gp, gstride, front = geo[c]
            assert (w, h, pd, st) == (gw, gh, gp, gstride)
            assert org == at + front + gp * gstride + gp and (org * S) % min(alignment, 32) == 0   # first sample of the picture aligned
            at += (front + gstride * (gp + gh + gp) + 255) & ~255
        sdt = np.uint8 if src_S == 1 else np.dtype("<u2")
        mx = 255 if src_S == 1 else 1023
        frame = [rng.integers(0, mx + 1, (h_, w_)).astype(sdt) for (w_, h_, *_r) in geo]
        yuv = np.concatenate([p.ravel() for p in frame])
        assert yuv.nbytes == width * height * 3 // 2 * src_S
        assert hv.L.havoc_mi355x_picture_upload_yuv(hv.h, pic, yuv.ctypes.data, src_S, shift, 1) == 0
        ddt = np.uint8 if S == 1 else np.uint16
        for c in range(3):
            exp = (frame[c].astype(np.uint32) << shift).astype(ddt)
            got = download(hv, pic, c, ddt, True)
            assert np.array_equal(got, np.pad(exp, geo[c][2], mode="edge")), c          # Padding::padImage
            assert np.array_equal(download(hv, pic, c, ddt, False), exp)
    finally:
        hv.L.havoc_mi355x_picture_destroy(hv.h, pic)


def test_plane_upload_pad_and_phase_planes(hv, oracle):
    """a host reconstruction goes up plane by plane, is padded on the device, and its 16 phase planes (made once, kept with the
    picture) hold what HavocPredUni gives per sample"""
    import suite
    rng = np.random.default_rng(9)
    width, height, pad = 192, 128, 96
    pic = C.c_void_p()
    assert hv.L.havoc_mi355x_picture_create(hv.h, 1, 8, width, height, pad, 64, C.byref(pic)) == 0
    try:
        host = [rng.integers(0, 256, (height >> (c > 0), width >> (c > 0))).astype(np.uint8) for c in range(3)]
        for c in range(3):
            assert hv.L.havoc_mi355x_picture_upload_plane(hv.h, pic, c, host[c].ctypes.data, host[c].shape[1], 0) == 0
        assert hv.L.havoc_mi355x_picture_pad(hv.h, pic) == 0
        for c in range(3):
            assert np.array_equal(download(hv, pic, c, np.uint8, True), np.pad(host[c], pad >> (c > 0), mode="edge"))
        dpl, pe, first = C.c_void_p(), C.c_ssize_t(), C.c_int64()
        for _ in range(2):   # second request: served from the kept planes
            assert hv.L.havoc_mi355x_picture_phase_planes(hv.h, pic, C.byref(dpl), C.byref(pe), C.byref(first)) == 0
        base, org, st, w, h, pd = plane_info(hv, pic, 0)
        n = pe.value
        planes = np.zeros(16 * n, np.uint8)
        assert hv.L.havoc_mi355x_d2h(hv.h, planes.ctypes.data, dpl, planes.nbytes) == 0
        planes = planes.reshape(16, n)
        padded = np.pad(host[0], pad, mode="edge")
        rows = h + 2 * pd
        front = (org - first.value) - (pd * st + pd)
        assert np.array_equal(planes[0][front:front + rows * st].reshape(rows, st)[:, :w + 2 * pd], padded)   # slot 0 = the picture
        # a block of every phase against the oracle's HavocPredUni on the padded host plane
        flat = np.zeros(rows * st + 64, np.uint8)
        flat[:rows * st].reshape(rows, st)[:, :w + 2 * pd] = padded
        for yf in range(4):
            for xf in range(4):
                if not (xf or yf):
                    continue
                bx, by, bw, bh = int(rng.integers(-60, w + 20)), int(rng.integers(-60, h + 20)), 32, 16
                dst = np.zeros(bw * bh, np.uint8)
                oracle.pred_uni(dst, 0, bw, flat, (by + pd) * st + bx + pd, st, bw, bh, xf, yf, 8, 8)
                o = front + (by + pd) * st + bx + pd
                got = np.stack([planes[4 * yf + xf][o + r * st:o + r * st + bw] for r in range(bh)])
                assert np.array_equal(got.ravel(), dst), (xf, yf)
    finally:
        hv.L.havoc_mi355x_picture_destroy(hv.h, pic)


def test_yuv_file_to_device_pictures(hv, tmp_path):
    """the on-disk input format (turing/encode.cpp:600-640): a raw planar 4:2:0 file, frame by frame, into device pictures; 8-bit
    file on an 8-bit and on a 10-bit picture (<< 2), 16-bit little-endian file on a 10-bit picture; a truncated last frame is ignored"""
    from turingcodec_amd.picture_io import DevicePicture, YuvReader
    from turingcodec_amd.workload import synth_frames
    W, H = 208, 120
    for file_bd, pic_bd in ((8, 8), (8, 10), (10, 10)):
        frames = synth_frames(W, H, 3, 5, file_bd)
        path = tmp_path / f"clip_{file_bd}.yuv"
        with open(path, "wb") as f:
            for fr in frames:
                for plane in fr:
                    f.write(np.ascontiguousarray(plane, np.uint8 if file_bd == 8 else "<u2").tobytes())
            f.write(b"\x00" * 1000)      # a partial frame at the end
        rd = YuvReader(str(path), W, H, file_bd)
        assert len(rd) == 3
        pic = DevicePicture(hv, W, H, pic_bd)
        for i, frame in enumerate(rd):
            pic.upload(frame, src_bit_depth=file_bd)
            for c in range(3):
                want = frames[i][c].astype(np.uint16) << (pic_bd - file_bd)
                got = pic.download(c, with_padding=True)
                p = (got.shape[0] - want.shape[0]) // 2
                assert np.array_equal(got[p:-p, p:-p], want), (file_bd, pic_bd, i, c)
                assert np.array_equal(got, np.pad(want, p, mode="edge")), "border replication"
                assert np.array_equal(rd.planes(i)[c], frames[i][c])
        base, elems, org = pic.phase_planes()
        assert base and elems > 0
        pic.close()
    with pytest.raises(ValueError):
        YuvReader(str(path), 4096, 4096, 10)
