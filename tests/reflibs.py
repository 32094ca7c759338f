"""ctypes bindings for the two CHECKERS (test infrastructure, never the product):

  * ``Oracle``    -> oracle/liboracle.so       (our plain-C restatement, oracle/havoc_oracle.c)
  * ``Reference`` -> oracle/_ref/libhavoc_ref.so (the reference's own havoc sources compiled by oracle/Makefile,
                    behind oracle/ref_shim.cpp).  ``handle`` 0 = C_REF|C_OPT tables, 1 = x86 JIT tables.

Both take numpy arrays; a "view" is (array, offset) so that negative tap offsets stay inside the buffer.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libhavoc_ref.so")

_ip = C.c_ssize_t
_vp = C.c_void_p


def build_oracle():
    """(Re)build oracle/liboracle.so with gcc -- seconds."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])


def _addr(a, off=0):
    """device-agnostic pointer to element `off` of numpy array a"""
    return a.ctypes.data + int(off) * a.itemsize


def _S(a):
    assert a.dtype in (np.uint8, np.uint16), a.dtype
    return a.itemsize


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < max(
                os.path.getmtime(os.path.join(ROOT, "oracle", f)) for f in ("havoc_oracle.c", "rdoq_oracle.c", "sao_oracle.c", "havoc_oracle.h")):
            build_oracle()
        L = self.L = C.CDLL(ORACLE_SO)
        L.oracle_sad.restype = C.c_int
        L.oracle_sad.argtypes = [_vp, _ip, _vp, _ip, C.c_int, C.c_int, C.c_int]
        L.oracle_sad4.restype = None
        L.oracle_sad4.argtypes = [_vp, _ip, C.POINTER(_vp), _ip, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int]
        L.oracle_ssd.restype = C.c_uint32
        L.oracle_ssd.argtypes = [_vp, _ip, _vp, _ip, C.c_int, C.c_int, C.c_int]
        L.oracle_satd.restype = C.c_int
        L.oracle_satd.argtypes = [_vp, _ip, _vp, _ip, C.c_int, C.c_int]
        L.oracle_pu_satd.restype = C.c_int
        L.oracle_pu_satd.argtypes = [_vp, _ip, _vp, _ip, C.c_int, C.c_int, C.c_int]
        L.oracle_ssd_linear.restype = C.c_int
        L.oracle_ssd_linear.argtypes = [_vp, _vp, C.c_int]
        L.oracle_pred_uni.restype = None
        L.oracle_pred_uni.argtypes = [_vp, _ip, _vp, _ip] + [C.c_int] * 7
        L.oracle_pred_bi.restype = None
        L.oracle_pred_bi.argtypes = [_vp, _ip, _vp, _vp, _ip] + [C.c_int] * 9
        L.oracle_subtract_bi.restype = None
        L.oracle_subtract_bi.argtypes = [_vp, _ip, _vp, _ip, _vp, _ip] + [C.c_int] * 4
        L.oracle_intra.restype = None
        L.oracle_intra.argtypes = [_vp, _ip, _vp] + [C.c_int] * 5
        L.oracle_transform.restype = None
        L.oracle_transform.argtypes = [_vp, _vp, _ip, C.c_int, C.c_int, C.c_int]
        L.oracle_inverse_transform.restype = None
        L.oracle_inverse_transform.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int]
        L.oracle_inverse_transform_add.restype = None
        L.oracle_inverse_transform_add.argtypes = [_vp, _ip, _vp, _ip, _vp] + [C.c_int] * 4
        L.oracle_quantize_inverse.restype = None
        L.oracle_quantize_inverse.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int]
        L.oracle_quantize.restype = C.c_int
        L.oracle_quantize.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.oracle_quantize_reconstruct.restype = None
        L.oracle_quantize_reconstruct.argtypes = [_vp, _ip, _vp, _ip, _vp, C.c_int]
        L.oracle_residual.restype = None
        L.oracle_residual.argtypes = [_vp, _ip, _vp, _ip, _vp, _ip, C.c_int, C.c_int, C.c_int]
        L.oracle_pad_block.restype = None
        L.oracle_pad_block.argtypes = [_vp, C.c_int, C.c_int, _ip] + [C.c_int] * 6

        L.oracle_deblock.restype = None
        L.oracle_deblock.argtypes = [_vp, _ip, _vp, _vp, _ip, C.c_int, C.c_int, C.c_int, _vp, _vp] + [C.c_int] * 5

        L.oracle_rdoq.restype = C.c_int
        L.oracle_rdoq.argtypes = [_vp, _vp] + [C.c_int] * 11 + [_vp]
        L.oracle_rdoq_lambda.restype = None
        L.oracle_rdoq_lambda.argtypes = [C.c_double, C.c_int, _vp, _vp]
        L.oracle_scan_order.restype = C.c_int
        L.oracle_scan_order.argtypes = [C.c_int] * 4

    def sao_stats(self, src, so, ss, rec, ro, rs, w, h, bd):
        """-> int64[105] (oracle/sao_oracle.c); src / rec: flat arrays, offsets and strides in samples"""
        out = np.zeros(105, np.int64)
        self.L.oracle_sao_stats.restype = None
        self.L.oracle_sao_stats.argtypes = [_vp, _ip, _vp, _ip] + [C.c_int] * 4 + [_vp]
        self.L.oracle_sao_stats(_addr(src, so), ss, _addr(rec, ro), rs, w, h, bd - 8, _S(src), _addr(out))
        return out

    def sao_band_chroma(self, src_u, src_v, so, ss, rec_u, rec_v, ro, rs, w, h, bd):
        """-> int64[65]: E[32], count[32], band position (oracle/sao_oracle.c)"""
        out = np.zeros(65, np.int64)
        self.L.oracle_sao_band_chroma.restype = None
        self.L.oracle_sao_band_chroma.argtypes = [_vp, _vp, _ip, _vp, _vp, _ip] + [C.c_int] * 4 + [_vp]
        self.L.oracle_sao_band_chroma(_addr(src_u, so), _addr(src_v, so), ss, _addr(rec_u, ro), _addr(rec_v, ro), rs, w, h, bd - 8, _S(src_u), _addr(out))
        return out

    def sao_filter(self, dst, do, ds, src, so, ss, w, h, kind, eo_class, offsets, bd):
        offsets = np.ascontiguousarray(offsets, np.int16)
        self.L.oracle_sao_filter.restype = None
        self.L.oracle_sao_filter.argtypes = [_vp, _ip, _vp, _ip] + [C.c_int] * 4 + [_vp, C.c_int, C.c_int]
        self.L.oracle_sao_filter(_addr(dst, do), ds, _addr(src, so), ss, w, h, kind, eo_class, _addr(offsets), bd, _S(src))

    def rdoq_lambda(self, lam, inv_scale):
        """(lambda in Q16, sign-data-hiding factor) as the reference's Rdoq constructor derives them from the double"""
        out = np.zeros(2, np.int32)
        self.L.oracle_rdoq_lambda(lam, inv_scale, _addr(out, 0), _addr(out, 1))
        return int(out[0]), int(out[1])

    def rdoq(self, src, log2, c_idx, scan_idx, is_intra, sdh, q_scale, q_shift, inv_scale, bd, lam, states):
        """-> (levels int16[n], OR of kept absolute levels)"""
        dst = np.zeros(1 << 2 * log2, np.int16)
        lq, sf = self.rdoq_lambda(lam, inv_scale)
        r = self.L.oracle_rdoq(_addr(dst), _addr(src), log2, c_idx, scan_idx, int(is_intra), int(sdh), q_scale, q_shift, inv_scale, bd, lq, sf, _addr(states))
        return dst, r

    def scan_order(self, log2, scan_idx, pos, comp):
        return self.L.oracle_scan_order(log2, scan_idx, pos, comp)

    def deblock(self, luma, sy, cb, cr, sc, width, height, bd, data, bs, tc2=0, beta2=0, cb_qp=0, cr_qp=0):
        """in place on the three planes (numpy arrays whose element 0 is sample (0, 0))"""
        self.L.oracle_deblock(_addr(luma), sy, _addr(cb), _addr(cr), sc, width, height, bd, _addr(data), _addr(bs), tc2, beta2, cb_qp, cr_qp, _S(luma))

    def intra_filter_neighbours(self, samples, n, bit_depth, strong):
        """IntraReferenceSamples::filter on the 4n + 1 samples in array order (left column from the bottom, corner, row above) -> the filtered copy"""
        a = np.ascontiguousarray(samples, np.int32)
        out = np.zeros_like(a)
        mid = (2 * n + 1) * 4
        self.L.oracle_intra_filter_neighbours.restype = None
        self.L.oracle_intra_filter_neighbours.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int]
        self.L.oracle_intra_filter_neighbours(a.ctypes.data + mid, out.ctypes.data + mid, n, bit_depth, strong)
        return out

    def intra_substitute(self, values, have, n, bit_depth):
        a = np.ascontiguousarray(values, np.int32).copy()
        h = np.ascontiguousarray(have, np.uint8)
        self.L.oracle_intra_substitute.restype = None
        self.L.oracle_intra_substitute.argtypes = [_vp, _vp, C.c_int, C.c_int]
        self.L.oracle_intra_substitute(a.ctypes.data, h.ctypes.data, n, bit_depth)
        return a

    def derive_bs(self, cells, width, height):
        """cells: CELL_DT-like structured array [height / 4, width / 4] -> (block_data int8, block_bs uint8) of the region grid"""
        n = ((width + 63) // 64 * 8 + 1) * ((height + 63) // 64 * 8 + 1)
        data, bs = np.zeros(n, np.int8), np.zeros(n, np.uint8)
        cells = np.ascontiguousarray(cells)
        self.L.oracle_derive_bs.restype = None
        self.L.oracle_derive_bs.argtypes = [_vp, _ip, C.c_int, C.c_int, _vp, _vp]
        self.L.oracle_derive_bs(cells.ctypes.data, cells.shape[1], width, height, data.ctypes.data, bs.ctypes.data)
        return data, bs

    # every method: arrays are flat (or 2-D C-contiguous) numpy arrays, offsets/strides in samples
    def sad(self, src, so, ss, ref, ro, rs, w, h):
        return self.L.oracle_sad(_addr(src, so), ss, _addr(ref, ro), rs, w, h, _S(src))

    def sad4(self, src, so, ss, ref, ros, rs, w, h):
        refs = (_vp * 4)(*[_addr(ref, r) for r in ros])
        out = (C.c_int * 4)()
        self.L.oracle_sad4(_addr(src, so), ss, refs, rs, out, w, h, _S(src))
        return list(out)

    def ssd(self, a, ao, sa, b, bo, sb, w, h):
        return self.L.oracle_ssd(_addr(a, ao), sa, _addr(b, bo), sb, w, h, _S(a))

    def satd(self, a, ao, sa, b, bo, sb, n):
        return self.L.oracle_satd(_addr(a, ao), sa, _addr(b, bo), sb, n, _S(a))

    def pu_satd(self, a, ao, sa, b, bo, sb, w, h):
        return self.L.oracle_pu_satd(_addr(a, ao), sa, _addr(b, bo), sb, w, h, _S(a))

    def ssd_linear(self, a, b, n):
        return self.L.oracle_ssd_linear(_addr(a), _addr(b), n)

    def pred_uni(self, dst, do, sd, ref, ro, sr, w, h, xf, yf, bd, taps):
        self.L.oracle_pred_uni(_addr(dst, do), sd, _addr(ref, ro), sr, w, h, xf, yf, bd, taps, _S(ref))

    def pred_bi(self, dst, do, sd, ref, r0, r1, sr, w, h, xf0, yf0, xf1, yf1, bd, taps):
        self.L.oracle_pred_bi(_addr(dst, do), sd, _addr(ref, r0), _addr(ref, r1), sr, w, h, xf0, yf0, xf1, yf1, bd,
                              taps, _S(ref))

    def subtract_bi(self, dst, do, sd, pred, po, sp, src, so, ss, w, h, bd):
        self.L.oracle_subtract_bi(_addr(dst, do), sd, _addr(pred, po), sp, _addr(src, so), ss, w, h, bd, _S(src))

    def intra(self, dst, do, sd, nb, no, log2, mode, edge, bd):
        self.L.oracle_intra(_addr(dst, do), sd, _addr(nb, no), log2, mode, edge, bd, _S(nb))

    def transform(self, coeffs, co, src, so, stride, log2, tr, bd):
        self.L.oracle_transform(_addr(coeffs, co), _addr(src, so), stride, log2, tr, bd)

    def inverse_transform(self, dst, do, coeffs, co, log2, tr, bd):
        self.L.oracle_inverse_transform(_addr(dst, do), _addr(coeffs, co), log2, tr, bd)

    def inverse_transform_add(self, dst, do, sd, pred, po, sp, coeffs, co, log2, tr, bd):
        self.L.oracle_inverse_transform_add(_addr(dst, do), sd, _addr(pred, po), sp, _addr(coeffs, co), log2, tr, bd,
                                            _S(pred))

    def quantize_inverse(self, dst, do, src, so, scale, shift, n):
        self.L.oracle_quantize_inverse(_addr(dst, do), _addr(src, so), scale, shift, n)

    def quantize(self, dst, do, src, so, scale, shift, offset, n):
        return self.L.oracle_quantize(_addr(dst, do), _addr(src, so), scale, shift, offset, n)

    def quantize_reconstruct(self, rec, ro, sr, pred, po, sp, res, so, n):
        self.L.oracle_quantize_reconstruct(_addr(rec, ro), sr, _addr(pred, po), sp, _addr(res, so), n)

    def pad_block(self, plane, off, w, h, stride, pad, top, bottom, left, right):
        """in place on `plane` (flat array); off = sample offset of the block's (0, 0)"""
        self.L.oracle_pad_block(_addr(plane, off), w, h, stride, pad, int(top), int(bottom), int(left), int(right), _S(plane))

    def residual(self, res, ro, sres, src, so, ss, pred, po, sp, w, h):
        self.L.oracle_residual(_addr(res, ro), sres, _addr(src, so), ss, _addr(pred, po), sp, w, h, _S(src))


def have_reference():
    return os.path.exists(REF_SO)


class Reference:
    """The reference's own havoc functions (C tables: handle 0; x86 JIT tables: handle 1)."""

    def __init__(self, handle=0, path=None):
        """path: another library exporting the same ref_* veneer -- tests/test_classic_api.py compiles
        oracle/ref_shim.cpp (a client of the reference's table API) against OUR include/havoc headers"""
        self.h = handle
        L = self.L = C.CDLL(path or REF_SO)
        for sfx in ("u8", "u16"):
            getattr(L, "ref_sad_" + sfx).restype = C.c_int
            getattr(L, "ref_sad_" + sfx).argtypes = [C.c_int, _vp, _ip, _vp, _ip, C.c_int, C.c_int]
            getattr(L, "ref_sad4_" + sfx).restype = C.c_int
            getattr(L, "ref_sad4_" + sfx).argtypes = [C.c_int, _vp, _ip, _vp, _vp, _vp, _vp, _ip,
                                                      C.POINTER(C.c_int), C.c_int, C.c_int]
            getattr(L, "ref_ssd_" + sfx).restype = C.c_longlong
            getattr(L, "ref_ssd_" + sfx).argtypes = [C.c_int, _vp, _ip, _vp, _ip, C.c_int]
            getattr(L, "ref_satd_" + sfx).restype = C.c_int
            getattr(L, "ref_satd_" + sfx).argtypes = [C.c_int, _vp, _ip, _vp, _ip, C.c_int]
            getattr(L, "ref_pred_uni_" + sfx).restype = C.c_int
            getattr(L, "ref_pred_uni_" + sfx).argtypes = [C.c_int, _vp, _ip, _vp, _ip] + [C.c_int] * 6
            getattr(L, "ref_pred_bi_" + sfx).restype = C.c_int
            getattr(L, "ref_pred_bi_" + sfx).argtypes = [C.c_int, _vp, _ip, _vp, _vp, _ip] + [C.c_int] * 8
            getattr(L, "ref_subtract_bi_" + sfx).restype = C.c_int
            getattr(L, "ref_subtract_bi_" + sfx).argtypes = [C.c_int, _vp, _ip, _vp, _ip, _vp, _ip] + [C.c_int] * 3
            getattr(L, "ref_intra_" + sfx).restype = C.c_int
            getattr(L, "ref_intra_" + sfx).argtypes = [C.c_int, _vp, _ip, _vp] + [C.c_int] * 4
            getattr(L, "ref_inverse_transform_add_" + sfx).restype = C.c_int
            getattr(L, "ref_inverse_transform_add_" + sfx).argtypes = [C.c_int, _vp, _ip, _vp, _ip, _vp] + [C.c_int] * 3
        L.ref_inverse_transform.restype = C.c_int
        L.ref_inverse_transform.argtypes = [C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int]
        L.ref_transform.restype = C.c_int
        L.ref_transform.argtypes = [C.c_int, _vp, _vp, _ip, C.c_int, C.c_int, C.c_int]
        L.ref_quantize_inverse.restype = C.c_int
        L.ref_quantize_inverse.argtypes = [C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int]
        L.ref_quantize.restype = C.c_int
        L.ref_quantize.argtypes = [C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_quantize_reconstruct.restype = C.c_int
        L.ref_quantize_reconstruct.argtypes = [C.c_int, _vp, _ip, _vp, _ip, _vp, C.c_int]
        L.ref_ssd_linear.restype = C.c_int
        L.ref_ssd_linear.argtypes = [C.c_int, _vp, _vp, C.c_int]
        L.ref_mask.restype = C.c_int
        L.ref_mask.argtypes = [C.c_int]
        for sfx in ("u8", "u16"):   # only in libhavoc_ref.so (ref_shim_turing.cpp), not in the classic-API client
            if hasattr(L, "ref_pad_block_" + sfx):
                getattr(L, "ref_pad_block_" + sfx).restype = None
                getattr(L, "ref_pad_block_" + sfx).argtypes = [_vp, C.c_int, C.c_int, _ip] + [C.c_int] * 5

    @staticmethod
    def _sfx(a):
        return "u8" if a.itemsize == 1 else "u16"

    def deblock(self, luma, sy, cb, cr, sc, width, height, bd, data, bs, tc2=0, beta2=0, cb_qp=0, cr_qp=0):
        """the reference's own LoopFilter::Picture::deblock templates in its CTU order (oracle/ref_shim_deblock.cpp); in place"""
        f = self._f("ref_deblock", luma)
        f.restype = None
        f.argtypes = [_vp, _ip, _vp, _vp, _ip, C.c_int, C.c_int, C.c_int, _vp, _vp] + [C.c_int] * 4
        f(_addr(luma), sy, _addr(cb), _addr(cr), sc, width, height, bd, _addr(data), _addr(bs), tc2, beta2, cb_qp, cr_qp)

    def derive_bs(self, width, height, cus, pus, tus):
        """the reference's own LoopFilter::Picture::processCu / Tu / Rc and sameMotion over lists of units (oracle/ref_shim_deblock.cpp):
        cus int32 [n, 6], pus int32 [n, 10], tus int32 [n, 5] -> (block_data int8, block_bs uint8)"""
        n = ((width + 63) // 64 * 8 + 1) * ((height + 63) // 64 * 8 + 1)
        data, bs = np.zeros(n, np.int8), np.zeros(n, np.uint8)
        cus, pus, tus = (np.ascontiguousarray(a, np.int32) for a in (cus, pus, tus))
        f = self.L.ref_derive_bs
        f.restype = None
        f.argtypes = [C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp]
        f(width, height, cus.ctypes.data, len(cus), pus.ctypes.data, len(pus), tus.ctypes.data, len(tus), data.ctypes.data, bs.ctypes.data)
        return data, bs

    def _f(self, name, a):
        return getattr(self.L, name + "_" + self._sfx(a))

    def rdoq(self, src, log2, c_idx, scan_idx, is_intra, sdh, q_scale, q_shift, inv_scale, bd, lam, states):
        """the reference's own Rdoq::runQuantisation (oracle/ref_shim_rdoq.cpp) -> (levels int16[n], return value)"""
        f = self.L.ref_rdoq
        f.restype = C.c_int
        f.argtypes = [_vp, _vp] + [C.c_int] * 9 + [C.c_double, _vp]
        dst = np.zeros(1 << 2 * log2, np.int16)
        r = f(_addr(dst), _addr(src), log2, c_idx, scan_idx, int(is_intra), int(sdh), q_scale, q_shift, inv_scale, bd, float(lam), _addr(states))
        return dst, r

    def sao_stats(self, src, so, ss, rec, ro, rs, w, h, bd):
        """the reference's EncSao statistics templates (oracle/ref_shim_sao.cpp) -> int64[105]"""
        f = self._f("ref_sao_stats", src)
        f.restype = None
        f.argtypes = [_vp, _ip, _vp, _ip] + [C.c_int] * 3 + [_vp]
        out = np.zeros(105, np.int64)
        f(_addr(src, so), ss, _addr(rec, ro), rs, w, h, bd - 8, _addr(out))
        return out

    def sao_band_chroma(self, src_u, src_v, so, ss, rec_u, rec_v, ro, rs, w, h, bd):
        """the reference's EncSao::band_offset_chroma_stats (EncSao.h:62-109) -> int64[65]: E[32], count[32], returned band position"""
        f = self._f("ref_sao_band_chroma", src_u)
        f.restype = C.c_int
        f.argtypes = [_vp, _vp, _ip, _vp, _vp, _ip, C.c_int, C.c_int, C.c_int, _vp, _vp]
        out = np.zeros(65, np.int64)
        out[64] = f(_addr(src_u, so), _addr(src_v, so), ss, _addr(rec_u, ro), _addr(rec_v, ro), rs, w, h, bd - 8, _addr(out), _addr(out, 32))
        return out

    def sao_filter(self, dst, do, ds, src, so, ss, w, h, kind, eo_class, offsets, bd):
        """turing/sao.cpp: kind 1 = sao_filter_band (32-entry table), 2 = sao_filter_edge (SaoOffsetVal[5])"""
        offsets = np.ascontiguousarray(offsets, np.int16)
        if kind == 1:
            f = self._f("ref_sao_band", src)
            f.restype = None
            f.argtypes = [_vp, _ip, _vp, _ip, C.c_int, C.c_int, _vp, C.c_int]
            f(_addr(dst, do), ds, _addr(src, so), ss, w, h, _addr(offsets), bd)
        else:
            f = self._f("ref_sao_edge", src)
            f.restype = None
            f.argtypes = [_vp, _ip, _vp, _ip, C.c_int, C.c_int, _vp, C.c_int, C.c_int]
            f(_addr(dst, do), ds, _addr(src, so), ss, w, h, _addr(offsets), eo_class, bd)

    def rdoq_initial_states(self, qp, init_type):
        """the 128-byte state snapshot a slice of that QP / initType starts from (Contexts::initialize)"""
        f = self.L.ref_rdoq_initial_states
        f.restype = None
        f.argtypes = [C.c_int, C.c_int, _vp]
        out = np.zeros(128, np.uint8)
        f(qp, init_type, _addr(out))
        return out

    def scan_order(self, log2, scan_idx, pos, comp):
        f = self.L.ref_scan_order
        f.restype = C.c_int
        f.argtypes = [C.c_int] * 4
        return f(log2, scan_idx, pos, comp)

    def mask(self):
        return self.L.ref_mask(self.h)

    def pad_block(self, plane, off, w, h, stride, pad, top, bottom, left, right):
        self._f("ref_pad_block", plane)(_addr(plane, off), w, h, stride, pad, int(top), int(bottom), int(left), int(right))

    def sad(self, src, so, ss, ref, ro, rs, w, h):
        return self._f("ref_sad", src)(self.h, _addr(src, so), ss, _addr(ref, ro), rs, w, h)

    def sad4(self, src, so, ss, ref, ros, rs, w, h):
        out = (C.c_int * 4)()
        r = self._f("ref_sad4", src)(self.h, _addr(src, so), ss, *[_addr(ref, x) for x in ros], rs, out, w, h)
        assert r == 0
        return list(out)

    def ssd(self, a, ao, sa, b, bo, sb, w, h):
        assert w == h
        return self._f("ref_ssd", a)(self.h, _addr(a, ao), sa, _addr(b, bo), sb, int(w).bit_length() - 1)

    def satd(self, a, ao, sa, b, bo, sb, n):
        return self._f("ref_satd", a)(self.h, _addr(a, ao), sa, _addr(b, bo), sb, int(n).bit_length() - 1)

    def ssd_linear(self, a, b, n):
        return self.L.ref_ssd_linear(self.h, _addr(a), _addr(b), n)

    def pred_uni(self, dst, do, sd, ref, ro, sr, w, h, xf, yf, bd, taps):
        assert self._f("ref_pred_uni", ref)(self.h, _addr(dst, do), sd, _addr(ref, ro), sr, w, h, xf, yf, bd, taps) == 0

    def pred_bi(self, dst, do, sd, ref, r0, r1, sr, w, h, xf0, yf0, xf1, yf1, bd, taps):
        assert self._f("ref_pred_bi", ref)(self.h, _addr(dst, do), sd, _addr(ref, r0), _addr(ref, r1), sr, w, h,
                                           xf0, yf0, xf1, yf1, bd, taps) == 0

    def subtract_bi(self, dst, do, sd, pred, po, sp, src, so, ss, w, h, bd):
        assert self._f("ref_subtract_bi", src)(self.h, _addr(dst, do), sd, _addr(pred, po), sp, _addr(src, so), ss,
                                               w, h, bd) == 0

    def intra(self, dst, do, sd, nb, no, log2, mode, edge, bd):
        # edge <=> cIdx == 0 (the table itself applies log2 < 5), havoc/pred_intra.h:41-48
        assert self._f("ref_intra", nb)(self.h, _addr(dst, do), sd, _addr(nb, no), 0 if edge else 1, bd, log2, mode) == 0

    def transform(self, coeffs, co, src, so, stride, log2, tr, bd):
        assert self.L.ref_transform(self.h, _addr(coeffs, co), _addr(src, so), stride, bd, tr, log2) == 0

    def inverse_transform(self, dst, do, coeffs, co, log2, tr, bd):
        assert self.L.ref_inverse_transform(self.h, _addr(dst, do), _addr(coeffs, co), bd, tr, log2) == 0

    def inverse_transform_add(self, dst, do, sd, pred, po, sp, coeffs, co, log2, tr, bd):
        assert self._f("ref_inverse_transform_add", pred)(self.h, _addr(dst, do), sd, _addr(pred, po), sp,
                                                          _addr(coeffs, co), bd, tr, log2) == 0

    def quantize_inverse(self, dst, do, src, so, scale, shift, n):
        assert self.L.ref_quantize_inverse(self.h, _addr(dst, do), _addr(src, so), scale, shift, n) == 0

    def quantize(self, dst, do, src, so, scale, shift, offset, n):
        return self.L.ref_quantize(self.h, _addr(dst, do), _addr(src, so), scale, shift, offset, n)

    def quantize_reconstruct(self, rec, ro, sr, pred, po, sp, res, so, n):
        assert self.L.ref_quantize_reconstruct(self.h, _addr(rec, ro), sr, _addr(pred, po), sp, _addr(res, so),
                                               int(n).bit_length() - 1) == 0
