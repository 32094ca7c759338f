"""Host-side binding of libhavoc_mi355x.so (C ABI: include/havoc_mi355x.h).

PyTorch is used only as plumbing: device buffers (torch.uint8/int16/int32 CUDA tensors) and the current HIP
stream.  Every compute call goes through the C ABI into the hand-written gfx950 kernels; if the shared library
is missing or there is no GPU this module raises -- there is no CPU/eager fallback.

Two layers:
  * ``Havoc.<primitive>_d(...)``  device-level: tensors already in HBM, asynchronous on the context stream
  * ``Havoc.<primitive>(...)``    numpy-level batch interface used by the parity suite (tests/suite.py): uploads
                                  inputs, launches once per primitive (per size class where the reference's table
                                  is indexed by size), downloads the result.
Names and argument meaning mirror the reference's function types (havoc/sad.h:58, pred_inter.h:35, ...).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HAVOC_MI355X_LIB") or os.path.join(_HERE, "libhavoc_mi355x.so")      # the override: diagnostic builds of profiles/micro/ (timing variants)

_vp = C.c_void_p
_ip = C.c_ssize_t
_i = C.c_int



def search_workspace(width, height):
    """bytes of the device workspace a picture search of this size needs (havoc_mi355x_search_workspace)"""
    L, _ = _load()
    return int(L.havoc_mi355x_search_workspace(width, height))


class HavocError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise HavocError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(make -C turingcodec_amd/csrc).  There is no fallback path.")
    L = C.CDLL(LIB_PATH)
    L.havoc_mi355x_last_error.restype = C.c_char_p
    L.havoc_mi355x_version.restype = C.c_char_p
    sig = {
        "create": [C.POINTER(_vp), _i, _vp],
        "destroy": [_vp],
        "set_stream": [_vp, _vp],
        "sync": [_vp],
        "sync_spin": [_vp],
        "device_info": [_vp, C.POINTER(C.c_int64)],
        "malloc": [_vp, C.POINTER(_vp), C.c_size_t],
        "free": [_vp, _vp],
        "h2d": [_vp, _vp, _vp, C.c_size_t],
        "d2h": [_vp, _vp, _vp, C.c_size_t],
        "host_alloc": [_vp, C.c_size_t, C.POINTER(_vp), C.POINTER(_vp)],
        "host_free": [_vp, _vp],
        "h2d_async": [_vp, _vp, _vp, C.c_size_t],
        "d2h_async": [_vp, _vp, _vp, C.c_size_t],
        "copy_2d": [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.c_size_t, C.c_size_t, _i],
        "picture_create": [_vp, _i, _i, _i, _i, _i, _i, C.POINTER(_vp)],
        "picture_destroy": [_vp, _vp],
        "picture_plane": [_vp, _i, C.POINTER(_vp), C.POINTER(C.c_int64), C.POINTER(_ip), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)],
        "picture_upload_yuv": [_vp, _vp, _vp, _i, _i, _i],
        "picture_upload_plane": [_vp, _vp, _i, _vp, _ip, _i],
        "picture_download_plane": [_vp, _vp, _i, _vp, _ip, _i],
        "picture_pad": [_vp, _vp],
        "picture_phase_planes": [_vp, _vp, C.POINTER(_vp), C.POINTER(_ip), C.POINTER(C.c_int64)],
        "timer_start": [_vp],
        "timer_stop_ms": [_vp, C.POINTER(C.c_float)],
        "graph_begin": [_vp],
        "graph_end": [_vp, C.POINTER(_vp)],
        "graph_launch": [_vp, _vp],
        "graph_destroy": [_vp],
        "fork": [_vp, _i],
        "lane": [_vp, _i],
        "join": [_vp],
        "sad": [_vp, _i, _vp, _ip, _vp, _ip, _vp, _i, _vp],
        "sad4": [_vp, _i, _vp, _ip, _vp, _ip, _vp, _i, _vp],
        "sad4_runs": [_vp, _i, _vp, _ip, _vp, _ip, _vp, _i, _vp, _i, _vp],
        "sad4_make_runs": [_vp, _i, _i, _ip, _i, _vp],
        "pad_block": [_vp, _i, _vp, C.c_int64, _i, _i, _ip, _i, _i, _i, _i, _i],
        "sad_surface": [_vp, _i, _i, _i, _i, _vp, _ip, _vp, _ip, _vp, _i, _vp],
        "deblock": [_vp, _i, _i, _vp, _ip, _vp, _vp, _ip, _i, _i, _vp, _vp, _i, _i, _i, _i],
        "derive_bs": [_vp, _vp, _ip, _i, _i, _vp, _vp],
        "ssd": [_vp, _i, _vp, _ip, _vp, _ip, _vp, _i, _vp],
        "satd": [_vp, _i, _i, _i, _vp, _ip, _vp, _ip, _vp, _i, _vp],
        "satd_multi": [_vp, _i, _i, _i, _vp, _ip, _vp, _ip, _vp, _i, _vp],
        "ssd_linear": [_vp, _vp, _vp, _i, _vp],
        "pred_uni": [_vp, _i, _i, _i, _i, _i, _vp, _ip, _vp, _ip, _vp, _i],
        "pred_bi": [_vp, _i, _i, _i, _i, _i, _vp, _ip, _vp, _ip, _vp, _i],
        "subtract_bi": [_vp, _i, _i, _vp, _ip, _vp, _ip, _vp, _ip, _vp, _i],
        "pred_uni_classes": [_vp, _i, _i, _i, _vp, _ip, _vp, _ip, _vp, C.POINTER(C.c_int32)],
        "pred_bi_classes": [_vp, _i, _i, _i, _vp, _ip, _vp, _ip, _vp, C.POINTER(C.c_int32)],
        "intra": [_vp, _i, _i, _i, _vp, _ip, _vp, _vp, _i],
        "intra_satd35": [_vp, _i, _i, _i, _vp, _ip, _vp, _vp, _i, _vp],
        "subpel_satd": [_vp, _i, _i, _i, _i, _i, _vp, _ip, _vp, _ip, _vp, _i, _vp],
        "interp_planes": [_vp, _i, _i, _vp, _ip, _vp, _ip, _i, _i, _i, _i],
        "residual": [_vp, _i, _vp, _ip, _vp, _vp, _ip, _vp, _ip, _vp, _i],
        "transform": [_vp, _i, _i, _i, _vp, _vp, _ip, _vp, _i],
        "inverse_transform": [_vp, _i, _i, _i, _vp, _vp, _vp, _i],
        "inverse_transform_add": [_vp, _i, _i, _i, _i, _vp, _ip, _vp, _ip, _vp, _vp, _i],
        "tu_forward": [_vp, _i, _i, _i, _i, _vp, _vp, _ip, _vp, _ip, _vp, _i],
        "tu_reconstruct": [_vp, _i, _i, _i, _i, _i, _i, _vp, _ip, _vp, _ip, _vp, _ip, _vp, _vp, _i, _vp],
        "intra_measure": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _ip, _vp, _ip, _vp, _i, _i],
        "level_stats": [_vp, _vp, _vp, _i, _vp],
        "search_motion_uni": [_vp, _i, _vp, _vp, C.c_int64, _ip, _vp, C.c_int64, _ip, _vp, _ip, C.c_int64, _vp, _i, _vp],
        "rqt_decide": [_vp, _vp, _i, _vp, _vp, _vp, C.c_int64, C.c_ssize_t, _i, _i, _vp],
        "block_cells": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
        "intra_gather": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp],
        "intra_commit": [_vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i],
        "intra_fill_spare": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp],
        "merge_jobs": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
        "merge_decide": [_vp, _vp, _vp, _vp, _i, C.c_int64, _vp, _vp],
        "pred_jobs": [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp],
        "search_motion_bi": [_vp, _i, _vp, _vp, C.c_int64, _ip, _vp, C.c_int64, _ip, _vp, _ip, C.c_int64, _vp, C.c_int64, _vp, _vp, _i, _vp],
        "search_picture_uni": [_vp, _i, _vp, _vp, _vp, C.c_int64, _ip, _vp, _vp, _ip, _vp, _ip, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i],
        "search_gate": [_vp, _vp],
        "search_wait_rows": [_vp, _vp, _i, _i, _i, _vp],
        "block_cells_add": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
        "intra_order": [_vp, _vp, _vp, _i, C.c_int32, _vp, _vp, _vp, _vp],
        "intra_expand": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
        "intra_decide": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, C.c_int32, _vp, _vp],
        "quantize": [_vp, _vp, _vp, _vp, _i, _vp],
        "quantize_inverse": [_vp, _vp, _vp, _vp, _i],
        "quantize_reconstruct": [_vp, _i, _vp, _ip, _vp, _ip, _vp, _vp, _i],
        "rdoq": [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, C.c_size_t],
        "tu_forward_scan": [_vp, _i, _i, _i, _vp, _vp, _ip, _vp, _ip, _vp, _i, _vp, _vp, _vp, C.c_size_t],
        "rdoq_prescanned": [_vp, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, C.c_size_t],
        "sao_stats": [_vp, _i, _i, _vp, _ip, _vp, _ip, _vp, _i, _vp],
        "sao_filter": [_vp, _i, _i, _vp, _ip, _vp, _ip, _vp, _i],
        "sao_band_chroma": [_vp, _i, _i, _vp, _ip, _vp, _ip, _vp, _i, _vp],
    }
    L.havoc_mi355x_rdoq_lambda.argtypes = [C.c_double, _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.havoc_mi355x_rdoq_lambda.restype = None
    L.havoc_mi355x_rdoq_workspace.argtypes = [_i]
    L.havoc_mi355x_rdoq_workspace.restype = C.c_size_t
    L.havoc_mi355x_search_workspace.argtypes = [_i, _i]
    L.havoc_mi355x_search_workspace.restype = C.c_size_t
    for name, args in sig.items():
        f = getattr(L, "havoc_mi355x_" + name)
        f.argtypes = args
        f.restype = None if name in ("destroy", "graph_destroy", "picture_destroy") else _i
    return L, sorted(sig)


def exported_symbols():
    """names the C ABI must export (checked against include/havoc_mi355x.h by the CPU tests)"""
    _, names = _load()
    return ["havoc_mi355x_" + n for n in names] + ["havoc_mi355x_last_error", "havoc_mi355x_version", "havoc_mi355x_rdoq_lambda", "havoc_mi355x_rdoq_workspace",
                                                "havoc_mi355x_search_workspace"]


# one havoc_mi355x_cell (include/havoc_mi355x.h), 16 bytes: a 4x4 luma cell of a picture's block structure
CELL_DT = np.dtype([("mv", "<i2", (2, 2)), ("dpb_index", "i1", (2,)), ("flags", "u1"), ("qp_y", "i1"), ("tu_log2", "u1"), ("reserved", "u1", (3,))])
assert CELL_DT.itemsize == 16
CELL_INTRA, CELL_CODED, CELL_NO_FILTER, CELL_PU_LEFT, CELL_PU_TOP = 1, 2, 4, 8, 16

# one havoc_mi355x_sao_job (include/havoc_mi355x.h), 96 bytes
SAO_JOB_DT = np.dtype([("dst_off", "<i4"), ("src_off", "<i4"), ("w", "<i4"), ("h", "<i4"), ("type", "<i4"), ("eo_class", "<i4"), ("offsets", "<i2", 32),
                       ("reserved", "<i4", 2)])
assert SAO_JOB_DT.itemsize == 96

# one havoc_mi355x_rdoq_job (include/havoc_mi355x.h), 48 bytes
RDOQ_JOB_DT = np.dtype([("dst_off", "<i4"), ("src_off", "<i4"), ("quant_scale", "<i4"), ("quant_shift", "<i4"), ("inv_scale", "<i4"),
                        ("lambda_q16", "<i4"), ("sdh_factor", "<i4"), ("ctx_index", "<i4"), ("c_idx", "u1"), ("scan_idx", "u1"),
                        ("is_intra", "u1"), ("sdh", "u1"), ("reserved", "<i4", 3)])
assert RDOQ_JOB_DT.itemsize == 48


def rdoq_lambda(lam, inv_scale):
    """(lambda_q16, sdh_factor) of a job, as the reference's Rdoq constructor derives them (turing/Rdoq.h:163-167)"""
    L, _ = _load()
    a, b = C.c_int32(), C.c_int32()
    L.havoc_mi355x_rdoq_lambda(float(lam), int(inv_scale), C.byref(a), C.byref(b))
    return a.value, b.value


def _ptr(t):
    return t.data_ptr() if t is not None else None


class Havoc:
    """One context per process / GPU (the reference's `havoc_code`, havoc/havoc.h:138-147)."""

    def __init__(self, device=0, stream=None):
        import torch
        self.torch = torch
        self.L, _ = _load()
        if not torch.cuda.is_available():
            raise HavocError("no GPU visible: libhavoc_mi355x has no CPU path")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        # stream: None -> torch's current stream; "new" -> a private non-blocking stream (needed for graph capture), made
        # through torch so that the numpy-level wrappers below can run their allocations, fills and copies on the SAME
        # stream as the kernels (`self.tstream`): nothing then depends on incidental synchronisation; an int -> that
        # hipStream_t (wrapped as an external torch stream for the same reason)
        if stream == "new":
            self.tstream = torch.cuda.Stream(device=self.device)
        elif stream is None:
            self.tstream = torch.cuda.current_stream(self.device)
        else:
            self.tstream = torch.cuda.ExternalStream(int(stream), device=self.device)
        s = _vp(self.tstream.cuda_stream)
        h = _vp()
        self._ck(self.L.havoc_mi355x_create(C.byref(h), device, s))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.havoc_mi355x_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise HavocError(f"libhavoc_mi355x error {rc}: {self.L.havoc_mi355x_last_error().decode()}")

    # ---------------------------------------------------------------- plumbing
    def sync(self):
        self._ck(self.L.havoc_mi355x_sync(self.h))

    def device_info(self):
        a = (C.c_int64 * 8)()
        self._ck(self.L.havoc_mi355x_device_info(self.h, a))
        keys = ["cus", "clock_khz", "mem_clock_khz", "bus_bits", "l2_bytes", "wave", "lds_per_wg", "mem_mib"]
        return dict(zip(keys, [int(x) for x in a]))

    def fork(self, nlanes):
        self._ck(self.L.havoc_mi355x_fork(self.h, nlanes))

    def lane(self, k):
        self._ck(self.L.havoc_mi355x_lane(self.h, k))

    def join(self):
        self._ck(self.L.havoc_mi355x_join(self.h))

    def graph_capture(self, fn):
        """record the launches `fn()` issues into a HIP graph; returns a handle for graph_launch"""
        self._ck(self.L.havoc_mi355x_graph_begin(self.h))
        try:
            fn()
        finally:
            g = _vp()
            rc = self.L.havoc_mi355x_graph_end(self.h, C.byref(g))
        self._ck(rc)
        return g

    def graph_launch(self, g):
        self._ck(self.L.havoc_mi355x_graph_launch(self.h, g))

    def graph_destroy(self, g):
        self.L.havoc_mi355x_graph_destroy(g)

    def timer_start(self):
        self._ck(self.L.havoc_mi355x_timer_start(self.h))

    def timer_stop_ms(self):
        ms = C.c_float()
        self._ck(self.L.havoc_mi355x_timer_stop_ms(self.h, C.byref(ms)))
        return ms.value

    def up(self, a):
        """numpy -> device tensor (uint16 travels as int16 bits: torch has no uint16 arithmetic, none is needed)"""
        a = np.ascontiguousarray(a)
        if a.dtype == np.uint16:
            a = a.view(np.int16)
        if a.dtype == np.uint32:
            a = a.view(np.int32)
        with self.torch.cuda.stream(self.tstream):   # ordered with the kernels: same stream
            return self.torch.from_numpy(a).to(self.device)

    def zeros(self, n, dtype):
        t = {np.uint8: self.torch.uint8, np.uint16: self.torch.int16, np.int16: self.torch.int16,
             np.int32: self.torch.int32, np.uint32: self.torch.int32}[np.dtype(dtype).type]
        with self.torch.cuda.stream(self.tstream):
            return self.torch.zeros(int(n), dtype=t, device=self.device)

    def down(self, t, dtype):
        with self.torch.cuda.stream(self.tstream):   # the copy queues behind the kernels that produced `t`
            a = t.cpu().numpy()
        return a.view(dtype) if a.dtype != np.dtype(dtype) else a

    @staticmethod
    def _S(t):
        return t.element_size()

    # ---------------------------------------------------------------- device-level API (tensors in HBM)
    def sad_d(self, src, ss, ref, rs, jobs, out):
        self._ck(self.L.havoc_mi355x_sad(self.h, self._S(src), _ptr(src), ss, _ptr(ref), rs, _ptr(jobs), jobs.shape[0], _ptr(out)))

    def sad_surface_d(self, src, ss, ref, rs, rng, max_w, max_h, jobs, out):
        self._ck(self.L.havoc_mi355x_sad_surface(self.h, self._S(src), rng, max_w, max_h, _ptr(src), ss, _ptr(ref), rs, _ptr(jobs),
                                                 jobs.shape[0], _ptr(out)))

    def pad_block_d(self, plane, origin, w, h, stride, pad, top=True, bottom=True, left=True, right=True):
        self._ck(self.L.havoc_mi355x_pad_block(self.h, self._S(plane), _ptr(plane), origin, w, h, stride, pad, int(top), int(bottom), int(left),
                                               int(right)))

    def pad_block(self, plane, origin, w, h, stride, pad, top=True, bottom=True, left=True, right=True):
        d = self.up(plane)
        self.pad_block_d(d, origin, w, h, stride, pad, top, bottom, left, right)
        return self.down(d, plane.dtype)

    def deblock_d(self, bd, luma, luma_off, sy, chroma, cb_off, cr_off, sc, width, height, data, bs, tc2=0, beta2=0, cb_qp=0, cr_qp=0):
        """in place; luma / chroma = device tensors, *_off = element offset of sample (0, 0) of each plane"""
        S = self._S(luma)
        self._ck(self.L.havoc_mi355x_deblock(self.h, S, bd, luma.data_ptr() + luma_off * S, sy, chroma.data_ptr() + cb_off * S,
                                             chroma.data_ptr() + cr_off * S, sc, width, height, _ptr(data), _ptr(bs), tc2, beta2, cb_qp, cr_qp))

    def deblock(self, bd, luma, sy, cb, cr, sc, width, height, data, bs, tc2=0, beta2=0, cb_qp=0, cr_qp=0):
        """numpy level: planes WITHOUT padding (stride = row length); returns the three filtered planes"""
        y = self.up(luma)
        c = self.up(np.concatenate([cb.ravel(), cr.ravel()]))
        d = self.torch.from_numpy(np.ascontiguousarray(data, np.int8)).to(self.device)
        b = self.torch.from_numpy(np.ascontiguousarray(bs, np.uint8)).to(self.device)
        self.deblock_d(bd, y, 0, sy, c, 0, cb.size, sc, width, height, d, b, tc2, beta2, cb_qp, cr_qp)
        o = self.down(c, cb.dtype)
        return self.down(y, luma.dtype), o[:cb.size].reshape(cb.shape), o[cb.size:].reshape(cr.shape)

    def derive_bs_d(self, cells, cells_stride, width, height, data, bs):
        """cells: uint8 tensor holding CELL_DT records (cells_stride cells per row); data int8 / bs uint8 tensors of the region grid"""
        self._ck(self.L.havoc_mi355x_derive_bs(self.h, _ptr(cells), cells_stride, width, height, _ptr(data), _ptr(bs)))

    def derive_bs(self, cells, width, height):
        """numpy level: cells CELL_DT [height / 4, width / 4] -> (block_data int8, block_bs uint8) flat arrays of the region grid"""
        import torch
        cells = np.ascontiguousarray(cells)
        n = ((width + 63) // 64 * 8 + 1) * ((height + 63) // 64 * 8 + 1)
        with torch.cuda.stream(self.tstream):
            d_cells = torch.from_numpy(cells.view(np.uint8).reshape(-1)).to(self.device)
            data = torch.zeros(n, dtype=torch.int8, device=self.device)
            bs = torch.zeros(n, dtype=torch.uint8, device=self.device)
        self.derive_bs_d(d_cells, cells.shape[1], width, height, data, bs)
        with torch.cuda.stream(self.tstream):
            return data.cpu().numpy(), bs.cpu().numpy()

    def sad4_d(self, src, ss, ref, rs, jobs, out):
        self._ck(self.L.havoc_mi355x_sad4(self.h, self._S(src), _ptr(src), ss, _ptr(ref), rs, _ptr(jobs), jobs.shape[0], _ptr(out)))

    def sad4_runs_d(self, src, ss, ref, rs, jobs, runs, out):
        """the same calls by runs (include/havoc_mi355x.h: havoc_mi355x_sad4_runs); runs: int32 tensor [nruns, 8] = havoc_mi355x_sad4_run records"""
        assert runs.shape[1] == 8, "runs: [n, 8] havoc_mi355x_sad4_run records (sad4_make_runs makes them; as_runs widens [n, 2] = (first job, count) pairs)"
        self._ck(self.L.havoc_mi355x_sad4_runs(self.h, self._S(src), _ptr(src), ss, _ptr(ref), rs, _ptr(jobs), jobs.shape[0], _ptr(runs), runs.shape[0], _ptr(out)))

    @staticmethod
    def as_runs(pairs):
        """(first job, count) pairs -> havoc_mi355x_sad4_run records without a box (the kernel then finds each run's box itself)"""
        pairs = np.asarray(pairs, np.int32).reshape(-1, 2)
        runs = np.zeros((len(pairs), 8), np.int32)
        runs[:, :2] = pairs
        return runs

    @staticmethod
    def sad4_make_runs(jobs, max_run=0, stride=None, S=1):
        """host: cut a sad4 job table (int32 [n, 8]) into runs of consecutive calls with equal source block and size -> int32 [nruns, 8] (first job, count, box offset,
        box width, box rows, 0, 0, 0).  stride = the reference plane's row stride in samples: with it every run gets the box of its candidates (and is cut where the box
        would outgrow the kernel's window); without it the runs have no box.  max_run 0 = by block size."""
        L, _ = _load()
        jobs = np.ascontiguousarray(jobs, np.int32)
        runs = np.zeros((max(1, len(jobs)), 8), np.int32)
        n = L.havoc_mi355x_sad4_make_runs(jobs.ctypes.data, len(jobs), max_run, stride if stride is not None else 0, S, runs.ctypes.data)
        if n < 0:
            raise HavocError("havoc_mi355x_sad4_make_runs failed")
        return np.ascontiguousarray(runs[:n])

    def ssd_d(self, a, sa, b, sb, jobs, out):
        self._ck(self.L.havoc_mi355x_ssd(self.h, self._S(a), _ptr(a), sa, _ptr(b), sb, _ptr(jobs), jobs.shape[0], _ptr(out)))

    def satd_d(self, a, sa, b, sb, jobs, out, max_w=64, max_h=64):
        self._ck(self.L.havoc_mi355x_satd(self.h, self._S(a), max_w, max_h, _ptr(a), sa, _ptr(b), sb, _ptr(jobs), jobs.shape[0], _ptr(out)))

    def satd_multi_d(self, a, sa, b, sb, jobs, out, max_w=64, max_h=64):
        self._ck(self.L.havoc_mi355x_satd_multi(self.h, self._S(a), max_w, max_h, _ptr(a), sa, _ptr(b), sb, _ptr(jobs), jobs.shape[0], _ptr(out)))

    def pred_uni_d(self, taps, bd, dst, sd, ref, sr, jobs, max_w=64, max_h=64):
        self._ck(self.L.havoc_mi355x_pred_uni(self.h, self._S(ref), taps, bd, max_w, max_h, _ptr(dst), sd, _ptr(ref), sr, _ptr(jobs), jobs.shape[0]))

    def pred_bi_d(self, taps, bd, dst, sd, ref, sr, jobs, max_w=64, max_h=64):
        self._ck(self.L.havoc_mi355x_pred_bi(self.h, self._S(ref), taps, bd, max_w, max_h, _ptr(dst), sd, _ptr(ref), sr, _ptr(jobs), jobs.shape[0]))

    # ---- job tables made on the device from the decided motion field (kernels_decide.hip: k_merge_jobs, k_pred_jobs, k_merge_decide) ----
    @staticmethod
    def field_layout(pic_width, pic_height, luma_stride, luma_pad, luma_plane_elems, chroma_stride, chroma_pad, chroma_plane_elems, search_range=64):
        return (C.c_int32 * 12)(pic_width, pic_height, search_range, (pic_width + 3) // 4, luma_stride, luma_pad, luma_plane_elems, chroma_stride, chroma_pad, chroma_plane_elems, 0, 0)

    # ---- the transform-tree decision and the block structure on the device (kernels_decide.hip: k_rqt_decide, k_block_cells) ----
    def rqt_decide_d(self, units, zero_at, one_at, sizes, rec_origin, rec_stride, dump_off, rl_q16, out):
        """sizes: numpy uint64 [4, 5] of device addresses (d_cbf, d_ssd, d_stats, d_jobs, d_final) per transform size 4, 8, 16, 32"""
        self._ck(self.L.havoc_mi355x_rqt_decide(self.h, _ptr(units), units.shape[0], _ptr(zero_at), _ptr(one_at), sizes.ctypes.data, int(rec_origin), int(rec_stride), int(dump_off),
                                                int(rl_q16), _ptr(out)))

    def block_cells_add_d(self, width, height, qp, dpb_index0, field, units, decisions, cells):
        """the cells of `units` into cells that keep what they hold elsewhere (havoc_mi355x_block_cells_add)"""
        self._ck(self.L.havoc_mi355x_block_cells_add(self.h, width, height, qp, dpb_index0, _ptr(field), _ptr(units), _ptr(decisions), units.shape[0], _ptr(cells)))

    def block_cells_d(self, width, height, qp, dpb_index0, field, units, decisions, cells):
        self._ck(self.L.havoc_mi355x_block_cells(self.h, width, height, qp, dpb_index0, _ptr(field), _ptr(units), _ptr(decisions), units.shape[0], _ptr(cells)))

    # ---- an intra picture's running state (kernels_decide.hip: k_intra_gather, k_intra_commit) ----
    @staticmethod
    def intra_chain_layout(pic_width, pic_height, stride, pad, cells_per_row, bit_depth, ctb_log2=6, strong_intra_smoothing=1):
        """strong_intra_smoothing: the reference encoder's default (turing/Encoder.cpp:688)"""
        return (C.c_int32 * 8)(pic_width, pic_height, stride, pad, cells_per_row, bit_depth, ctb_log2, int(bool(strong_intra_smoothing)))

    def intra_gather_a(self, S, layout, rec, owner, modes, parts, n, jobs, neighbours, mpm):
        """device ADDRESSES (ints): a level's slice of a size's tables"""
        self._ck(self.L.havoc_mi355x_intra_gather(self.h, S, layout, rec, owner, modes, parts, n, jobs, neighbours, mpm))

    def intra_commit_a(self, S, layout, rec, modes, parts, n, blocks, mode, mode_stride=1):
        self._ck(self.L.havoc_mi355x_intra_commit(self.h, S, layout, rec, modes, parts, n, blocks, mode, mode_stride))

    def merge_jobs_d(self, layout, field, x0, y0, log2, luma_jobs, cb_jobs, cr_jobs, vectors):
        self._ck(self.L.havoc_mi355x_merge_jobs(self.h, layout, _ptr(field), _ptr(x0), _ptr(y0), x0.shape[0], log2, _ptr(luma_jobs), _ptr(cb_jobs), _ptr(cr_jobs), _ptr(vectors)))

    def merge_decide_d(self, satd_y, satd_cb, satd_cr, n, lam_q16, cost, best):
        self._ck(self.L.havoc_mi355x_merge_decide(self.h, _ptr(satd_y), _ptr(satd_cb), _ptr(satd_cr), n, int(lam_q16), _ptr(cost), _ptr(best)))

    def pred_jobs_d(self, layout, field, lst, x0, y0, log2, plane, dst_off, jobs):
        self._ck(self.L.havoc_mi355x_pred_jobs(self.h, layout, _ptr(field), lst, _ptr(x0), _ptr(y0), x0.shape[0], log2, plane, _ptr(dst_off), _ptr(jobs)))

    @staticmethod
    def sort_by_class(jobs, w, h):
        """(jobs reordered so that the four size classes <= 8, 16, 32, 64 follow each other, counts[4], the permutation applied)"""
        big = np.maximum(np.asarray(w), np.asarray(h))
        cls = np.searchsorted([8, 16, 32], big, side="left")
        order = np.argsort(cls, kind="stable")
        counts = np.bincount(cls, minlength=4).astype(np.int32)
        return np.ascontiguousarray(np.asarray(jobs)[order]), counts, order

    def pred_classes_d(self, bi, taps, bd, dst, sd, ref, sr, jobs, counts):
        """all size classes of a (class-sorted) job table in one launch; counts: int32[4]"""
        c = (C.c_int32 * 4)(*[int(v) for v in counts])
        f = self.L.havoc_mi355x_pred_bi_classes if bi else self.L.havoc_mi355x_pred_uni_classes
        self._ck(f(self.h, self._S(ref), taps, bd, _ptr(dst), sd, _ptr(ref), sr, _ptr(jobs), c))

    @staticmethod
    def size_classes(w, h):
        """[(index array, max_w, max_h)]: the four block-size classes the interpolation kernels are instantiated for"""
        big = np.maximum(np.asarray(w), np.asarray(h))
        out = []
        for lo, hi in ((0, 8), (8, 16), (16, 32), (32, 64)):
            idx = np.flatnonzero((big > lo) & (big <= hi))
            if len(idx):
                out.append((idx, hi, hi))
        return out

    def subtract_bi_d(self, bd, dst, sd, pred, sp, src, ss, jobs):
        self._ck(self.L.havoc_mi355x_subtract_bi(self.h, self._S(src), bd, _ptr(dst), sd, _ptr(pred), sp, _ptr(src), ss, _ptr(jobs), jobs.shape[0]))

    def intra_d(self, bd, log2, dst, sd, nb, jobs):
        self._ck(self.L.havoc_mi355x_intra(self.h, self._S(nb), bd, log2, _ptr(dst), sd, _ptr(nb), _ptr(jobs), jobs.shape[0]))

    def search_gate(self, rows_ready=None):
        """havoc_mi355x_search_gate: the picture searches this context launches from now on wait, CTU row by CTU row, for their reference pictures -- rows_ready = an int32
        device tensor of 2 (rows of list 0 / list 1 that are final in the picture and its 16 phase planes, raised on another stream); None removes the gate"""
        self._gate = rows_ready      # (kept alive)
        self._ck(self.L.havoc_mi355x_search_gate(self.h, _ptr(rows_ready) if rows_ready is not None else None))

    def search_wait_rows(self, work, width, height, ctu_row, gave_up):
        """havoc_mi355x_search_wait_rows: a launch on THIS context's stream that ends when CTU rows 0 .. ctu_row (both lists) of the picture search running over the workspace
        `work` on another stream are done"""
        self._ck(self.L.havoc_mi355x_search_wait_rows(self.h, _ptr(work), width, height, ctu_row, _ptr(gave_up)))

    def search_picture_uni_d(self, S, params, mvp_rate, src, src_origin, src_stride, ref, ref_origin, ref_stride, phase, plane_elems, phase_origin, pus, ctu_first, ctus_x, ctus_y,
                             n_pus, out, out_bi, field, work):
        """havoc_mi355x_search_picture_uni with everything on the device and NOTHING downloaded: asynchronous (d_* = device tensors, params = the ctypes parameter record)"""
        ro, po, mr = (C.c_int64 * 2)(*[int(v) for v in ref_origin]), (C.c_int64 * 2)(*[int(v) for v in phase_origin]), (C.c_int64 * 2)(*[int(v) for v in mvp_rate])
        self._ck(self.L.havoc_mi355x_search_picture_uni(self.h, S, C.byref(params), mr, _ptr(src), int(src_origin), src_stride, _ptr(ref), ro, ref_stride, _ptr(phase), plane_elems, po,
                                                        _ptr(pus), _ptr(ctu_first), ctus_x, ctus_y, n_pus, _ptr(out), _ptr(out_bi) if out_bi is not None else None, _ptr(field),
                                                        _ptr(work), 0))

    def interp_planes_d(self, bd, planes, plane_elems, ref, stride, x0, y0, width, height):
        self._ck(self.L.havoc_mi355x_interp_planes(self.h, self._S(ref), bd, _ptr(planes), plane_elems, _ptr(ref), stride, x0, y0, width, height))

    def interp_planes(self, bd, ref, stride, x0, y0, width, height):
        """numpy level: returns the 16 planes as an array [16, len(ref)] (plane 0 and everything outside the rectangle 0)"""
        planes = self.zeros(16 * len(ref), ref.dtype)
        self.interp_planes_d(bd, planes, len(ref), self.up(ref), stride, x0, y0, width, height)
        return self.down(planes, ref.dtype).reshape(16, -1)

    def subpel_satd_d(self, taps, bd, max_w, max_h, src, ss, ref, sr, jobs, cost):
        self._ck(self.L.havoc_mi355x_subpel_satd(self.h, self._S(ref), taps, bd, max_w, max_h, _ptr(src), ss, _ptr(ref), sr, _ptr(jobs),
                                                 jobs.shape[0], _ptr(cost)))

    def subpel_satd(self, taps, bd, src, ss, ref, sr, jobs):
        """jobs: int32 [n, 8] = (src_off, ref_off, w, h, xFrac, yFrac, 0, 0); one launch per size class"""
        jobs = np.asarray(jobs, np.int32)
        out = np.zeros(len(jobs), np.int32)
        s, r = self.up(src), self.up(ref)
        big = np.maximum(jobs[:, 2], jobs[:, 3])
        for lo, hi in ((0, 8), (8, 16), (16, 32), (32, 64)):
            idx = np.flatnonzero((big > lo) & (big <= hi))
            if len(idx):
                cost = self.zeros(len(idx), np.int32)
                self.subpel_satd_d(taps, bd, hi, hi, s, ss, r, sr, self._jobs(jobs[idx], 8), cost)
                out[idx] = self.down(cost, np.int32)
        return out

    def intra_satd35_d(self, bd, log2, src, ss, nb, jobs, cost):
        self._ck(self.L.havoc_mi355x_intra_satd35(self.h, self._S(src), bd, log2, _ptr(src), ss, _ptr(nb), _ptr(jobs), jobs.shape[0], _ptr(cost)))

    def residual_d(self, res, sres, res_off, src, ss, pred, sp, jobs):
        self._ck(self.L.havoc_mi355x_residual(self.h, self._S(src), _ptr(res), sres, _ptr(res_off), _ptr(src), ss, _ptr(pred), sp, _ptr(jobs), jobs.shape[0]))

    def transform_d(self, bd, tr, log2, coeffs, res, sres, jobs):
        self._ck(self.L.havoc_mi355x_transform(self.h, bd, tr, log2, _ptr(coeffs), _ptr(res), sres, _ptr(jobs), jobs.shape[0]))

    def inverse_transform_d(self, bd, tr, log2, res, coeffs, jobs):
        self._ck(self.L.havoc_mi355x_inverse_transform(self.h, bd, tr, log2, _ptr(res), _ptr(coeffs), _ptr(jobs), jobs.shape[0]))

    def inverse_transform_add_d(self, bd, tr, log2, dst, sd, pred, sp, coeffs, jobs):
        self._ck(self.L.havoc_mi355x_inverse_transform_add(self.h, self._S(pred), bd, tr, log2, _ptr(dst), sd, _ptr(pred), sp, _ptr(coeffs), _ptr(jobs), jobs.shape[0]))

    def tu_forward_d(self, bd, tr, log2, coeffs, src, ss, pred, sp, jobs):
        self._ck(self.L.havoc_mi355x_tu_forward(self.h, self._S(src), bd, tr, log2, _ptr(coeffs), _ptr(src), ss, _ptr(pred), sp, _ptr(jobs), jobs.shape[0]))

    def tu_reconstruct_d(self, bd, tr, log2, scale, shift, rec, sr, pred, sp, src, ss, levels, jobs, ssd):
        self._ck(self.L.havoc_mi355x_tu_reconstruct(self.h, self._S(src), bd, tr, log2, scale, shift, _ptr(rec), sr, _ptr(pred), sp, _ptr(src), ss,
                                                    _ptr(levels), _ptr(jobs), jobs.shape[0], _ptr(ssd)))

    def intra_measure_d(self, bd, log2, coeffs, coeffs_dct, satd, rec0, ssd0, src, ss, pred, sp, jobs, with_satd=True):
        """the 35-mode stage of one intra partition in one launch (csrc/kernels_tu_fused.hip k_intra_measure): per job the tile SATDs, the forward transform
        (DST-VII for 4x4, plus the 4x4 DCT in coeffs_dct), and the reconstruction from zero levels with its SSD"""
        self._ck(self.L.havoc_mi355x_intra_measure(self.h, self._S(src), bd, log2, _ptr(coeffs), _ptr(coeffs_dct) if coeffs_dct is not None else None,
                                                   _ptr(satd) if satd is not None else None, _ptr(rec0), _ptr(ssd0), _ptr(src), ss, _ptr(pred), sp, _ptr(jobs),
                                                   jobs.shape[0], 1 if with_satd else 0))

    def level_stats_d(self, levels, jobs, njobs, out):
        """jobs: int32 [njobs, 2] (offset, count) into `levels`; out: int32 [2 * njobs] (non-zero levels, sum of magnitudes)"""
        self._ck(self.L.havoc_mi355x_level_stats(self.h, _ptr(levels), _ptr(jobs), njobs, _ptr(out)))

    def tu_forward(self, bd, ncoef, src, ss, pred, sp, jobs):
        """jobs: int32 [n, 8] = (coef_off, src_off, pred_off, rec_off, log2, trType, 0, 0)"""
        co = self.zeros(ncoef, np.int16)
        s, p = self.up(src), self.up(pred)
        for log2, tr, sel in self._tu_groups(jobs):
            self.tu_forward_d(bd, tr, log2, co, s, ss, p, sp, self._jobs(sel, 4))
        return self.down(co, np.int16)

    def tu_reconstruct(self, bd, qp, rec_len, sr, pred, sp, src, ss, levels, jobs):
        """de-quantiser parameters from qp as turing/QpState.h:85-86 / Reconstruct.cpp:315; returns (rec, ssd)"""
        jobs = np.asarray(jobs, np.int32)
        rec = self.zeros(rec_len, src.dtype)
        ssd = np.zeros(len(jobs), np.uint32)
        s, p, lv = self.up(src), self.up(pred), self.up(levels)
        for log2, tr in ((2, 1), (2, 0), (3, 0), (4, 0), (5, 0)):
            idx = np.flatnonzero((jobs[:, 4] == log2) & (jobs[:, 5] == tr))
            if len(idx):
                scale, shift = [40, 45, 51, 57, 64, 72][qp % 6] << (qp // 6), log2 - 1 + bd - 8
                o = self.zeros(len(idx), np.uint32)
                self.tu_reconstruct_d(bd, tr, log2, scale, shift, rec, sr, p, sp, s, ss, lv, self._jobs(jobs[idx], 4), o)
                ssd[idx] = self.down(o, np.uint32)
        return self.down(rec, src.dtype), ssd

    def quantize_d(self, dst, src, jobs, cbf):
        self._ck(self.L.havoc_mi355x_quantize(self.h, _ptr(dst), _ptr(src), _ptr(jobs), jobs.shape[0], _ptr(cbf)))

    def quantize_inverse_d(self, dst, src, jobs):
        self._ck(self.L.havoc_mi355x_quantize_inverse(self.h, _ptr(dst), _ptr(src), _ptr(jobs), jobs.shape[0]))

    def quantize_reconstruct_d(self, log2, rec, sr, pred, sp, res, jobs):
        self._ck(self.L.havoc_mi355x_quantize_reconstruct(self.h, log2, _ptr(rec), sr, _ptr(pred), sp, _ptr(res), _ptr(jobs), jobs.shape[0]))

    def sao_stats(self, bd, src, ss, rec, rs, jobs):
        """numpy level: jobs int32 [n, 4] = (src_off, rec_off, w, h) -> int64 [n, 105]"""
        jobs = np.ascontiguousarray(jobs, np.int32)
        with self.torch.cuda.stream(self.tstream):
            out = self.torch.zeros(105 * len(jobs), dtype=self.torch.int64, device=self.device)
        s, r = self.up(src), self.up(rec)
        self._ck(self.L.havoc_mi355x_sao_stats(self.h, self._S(s), bd, _ptr(s), ss, _ptr(r), rs, _ptr(self.up(jobs)), len(jobs), _ptr(out)))
        return self.down(out, np.int64).reshape(-1, 105)

    def sao_band_chroma(self, bd, src, ss, rec, rs, jobs):
        """jobs int32 [n, 8] (src_u, src_v, rec_u, rec_v, w, h, 0, 0) -> int64 [n, 65]: E[32], count[32], band position (EncSao.h:62-109)"""
        torch = self.torch
        jobs = np.ascontiguousarray(jobs, np.int32)
        d_src, d_rec, d_jobs = self.up(src), self.up(rec), self.up(jobs)
        with torch.cuda.stream(self.tstream):
            out = torch.zeros(65 * len(jobs), dtype=torch.int64, device=self.device)
        self._ck(self.L.havoc_mi355x_sao_band_chroma(self.h, self._S(d_src), bd, _ptr(d_src), ss, _ptr(d_rec), rs, _ptr(d_jobs), len(jobs), _ptr(out)))
        with torch.cuda.stream(self.tstream):
            return out.cpu().numpy().reshape(-1, 65)

    def sao_filter(self, bd, dst_like, sd, src, ss, jobs):
        """numpy level: jobs = SAO_JOB_DT array; returns the destination plane (zeros where no job wrote)"""
        jobs = np.ascontiguousarray(jobs, SAO_JOB_DT)
        dst = self.zeros(len(dst_like), dst_like.dtype)
        s = self.up(src)
        with self.torch.cuda.stream(self.tstream):
            j = self.torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(self.device)
        self._ck(self.L.havoc_mi355x_sao_filter(self.h, self._S(s), bd, _ptr(dst), sd, _ptr(s), ss, _ptr(j), len(jobs)))
        return self.down(dst, dst_like.dtype)

    def rdoq_workspace(self, njobs):
        """device scratch for one rdoq launch of `njobs` blocks (an int64 tensor: 16-byte aligned)"""
        n = int(self.L.havoc_mi355x_rdoq_workspace(int(njobs)))
        with self.torch.cuda.stream(self.tstream):
            return self.torch.zeros((n + 7) // 8 + 2, dtype=self.torch.int64, device=self.device)

    def rdoq_d(self, bd, log2, dst, src, states, jobs, cbf, work):
        """jobs: uint8 tensor holding RDOQ_JOB_DT records; states: uint8 tensor of 128-byte snapshots; work: rdoq_workspace(njobs)"""
        self._ck(self.L.havoc_mi355x_rdoq(self.h, bd, log2, _ptr(dst), _ptr(src), _ptr(states), _ptr(jobs), jobs.numel() // RDOQ_JOB_DT.itemsize, _ptr(cbf),
                                          _ptr(work), work.numel() * 8))

    def tu_forward_scan_d(self, bd, log2, coeffs, src, ss, pred, sp, jobs, rdoq_jobs, levels, work):
        """tu_forward with the RDOQ scan folded in (16x16 / 32x32): coefficients + RdoqInfo per block in `work`, level blocks zeroed"""
        self._ck(self.L.havoc_mi355x_tu_forward_scan(self.h, self._S(src), bd, log2, _ptr(coeffs), _ptr(src), ss, _ptr(pred), sp, _ptr(jobs), jobs.shape[0],
                                                     _ptr(rdoq_jobs), _ptr(levels), _ptr(work), work.numel() * 8))

    def rdoq_prescanned_d(self, bd, log2, dst, src, states, jobs, cbf, work):
        self._ck(self.L.havoc_mi355x_rdoq_prescanned(self.h, bd, log2, _ptr(dst), _ptr(src), _ptr(states), _ptr(jobs), jobs.numel() // RDOQ_JOB_DT.itemsize,
                                                     _ptr(cbf), _ptr(work), work.numel() * 8))

    def rdoq(self, bd, log2, src, states, jobs):
        """numpy level: src int16 (all blocks), states uint8 [k, 128], jobs RDOQ_JOB_DT array -> (levels int16 like src, cbf int32[njobs])"""
        jobs = np.ascontiguousarray(jobs, RDOQ_JOB_DT)
        dst = self.zeros(len(src), np.int16)
        cbf = self.zeros(len(jobs), np.int32)
        with self.torch.cuda.stream(self.tstream):
            st = self.torch.from_numpy(np.ascontiguousarray(states, np.uint8).reshape(-1)).to(self.device)
            j = self.torch.from_numpy(jobs.view(np.uint8).reshape(-1)).to(self.device)
        self.rdoq_d(bd, log2, dst, self.up(src), st, j, cbf, self.rdoq_workspace(len(jobs)))
        return self.down(dst, np.int16), self.down(cbf, np.int32)

    def ssd_linear_d(self, a, b, n, out):
        self._ck(self.L.havoc_mi355x_ssd_linear(self.h, _ptr(a), _ptr(b), n, _ptr(out)))

    # ---------------------------------------------------------------- numpy-level batch interface (tests/suite.py)
    def _jobs(self, j, ncols):
        j = np.ascontiguousarray(np.asarray(j, np.int32)[:, :ncols])
        return self.up(j)

    def sad(self, a, sa, b, sb, jobs):
        out = self.zeros(len(jobs), np.int32)
        self.sad_d(self.up(a), sa, self.up(b), sb, self._jobs(jobs, 4), out)
        return self.down(out, np.int32)

    def sad4(self, a, sa, b, sb, jobs):
        out = self.zeros(4 * len(jobs), np.int32)
        self.sad4_d(self.up(a), sa, self.up(b), sb, self._jobs(jobs, 8), out)
        return self.down(out, np.int32).reshape(-1, 4)

    def sad4_runs(self, a, sa, b, sb, jobs, runs=None, max_run=0):
        """havoc_sad_multiref calls by runs (one search's consecutive calls share a staged window); runs None = cut by sad4_make_runs"""
        jobs = np.ascontiguousarray(jobs, np.int32)
        if runs is None:
            runs = self.sad4_make_runs(jobs, max_run, sb, np.asarray(b).itemsize)
        runs = np.ascontiguousarray(runs, np.int32)
        if runs.ndim == 2 and runs.shape[1] == 2:
            runs = self.as_runs(runs)
        out = self.zeros(4 * len(jobs), np.int32)
        if len(jobs) and len(runs):
            self.sad4_runs_d(self.up(a), sa, self.up(b), sb, self._jobs(jobs, 8), self.up(runs), out)
        return self.down(out, np.int32).reshape(-1, 4)

    def sad_surface(self, a, sa, b, sb, rng, jobs):
        """jobs rows: src_off, ref_off, w, h; returns [njobs, 2R+1, 2R+1] (dy, dx)"""
        jobs = np.asarray(jobs, np.int32)
        side = 2 * rng + 1
        j = np.zeros((len(jobs), 8), np.int32)
        j[:, :4] = jobs[:, :4]
        j[:, 4] = np.arange(len(jobs)) * side * side
        out = self.zeros(len(jobs) * side * side, np.int32)
        if len(jobs):
            mw = (int(j[:, 2].max()) + 3) & ~3
            self.sad_surface_d(self.up(a), sa, self.up(b), sb, rng, mw, int(j[:, 3].max()), self.up(j), out)
        return self.down(out, np.int32).reshape(-1, side, side)

    def ssd(self, a, sa, b, sb, jobs):
        out = self.zeros(len(jobs), np.uint32)
        self.ssd_d(self.up(a), sa, self.up(b), sb, self._jobs(jobs, 4), out)
        return self.down(out, np.uint32)

    def satd_multi(self, a, sa, b, sb, jobs):
        """jobs rows: a_off, w, h, count, b_off[16]; returns [njobs, 16] (entries k >= count are 0); one launch per
        lane-group class like satd"""
        jobs = np.asarray(jobs, np.int32)
        out = np.zeros((len(jobs), 16), np.int32)
        ad, bdv = self.up(a), self.up(b)
        rows = ((jobs[:, 1] + 7) // 8) * jobs[:, 2]
        for lo, hi, mw, mh in ((0, 8, 8, 8), (8, 16, 16, 8), (16, 32, 16, 16), (32, 128, 32, 32), (128, 1 << 30, 64, 64)):
            idx = np.flatnonzero((rows > lo) & (rows <= hi))
            if len(idx):
                o = self.zeros(16 * len(idx), np.int32)
                self.satd_multi_d(ad, sa, bdv, sb, self._jobs(jobs[idx], 20), o, mw, mh)
                out[idx] = self.down(o, np.int32).reshape(-1, 16)
        return out

    def satd(self, a, sa, b, sb, jobs):
        """one launch per lane-group class so that every group size (8 / 16 / 32 / 64 lanes per job) is exercised"""
        jobs = np.asarray(jobs, np.int32)
        out = np.zeros(len(jobs), np.int32)
        ad, bdv = self.up(a), self.up(b)
        rows = ((jobs[:, 2] + 7) // 8) * jobs[:, 3]
        for lo, hi, mw, mh in ((0, 8, 8, 8), (8, 16, 16, 8), (16, 32, 16, 16), (32, 1 << 30, 64, 64)):
            idx = np.flatnonzero((rows > lo) & (rows <= hi))
            if len(idx):
                o = self.zeros(len(idx), np.int32)
                self.satd_d(ad, sa, bdv, sb, self._jobs(jobs[idx], 4), o, mw, mh)
                out[idx] = self.down(o, np.int32)
        return out

    def ssd_linear(self, a, b, n):
        out = self.zeros(1, np.int32)
        self.ssd_linear_d(self.up(a), self.up(b), n, out)
        return int(self.down(out, np.int32)[0])

    def pred_uni(self, taps, bd, dst_len, sd, ref, sr, jobs):
        jobs = np.asarray(jobs, np.int32)
        dst = self.zeros(dst_len, ref.dtype)
        r = self.up(ref)
        for idx, mw, mh in self.size_classes(jobs[:, 2], jobs[:, 3]):
            self.pred_uni_d(taps, bd, dst, sd, r, sr, self._jobs(jobs[idx], 8), mw, mh)
        return self.down(dst, ref.dtype)

    def pred_bi(self, taps, bd, dst_len, sd, ref, sr, jobs):
        jobs = np.asarray(jobs, np.int32)
        dst = self.zeros(dst_len, ref.dtype)
        r = self.up(ref)
        for idx, mw, mh in self.size_classes(jobs[:, 3], jobs[:, 4]):
            self.pred_bi_d(taps, bd, dst, sd, r, sr, self._jobs(jobs[idx], 12), mw, mh)
        return self.down(dst, ref.dtype)

    def subtract_bi(self, bd, dst_len, sd, pred, sp, src, ss, jobs):
        dst = self.zeros(dst_len, src.dtype)
        self.subtract_bi_d(bd, dst, sd, self.up(pred), sp, self.up(src), ss, self._jobs(jobs, 8))
        return self.down(dst, src.dtype)

    def intra(self, bd, dst_len, sd, nb, jobs):
        jobs = np.asarray(jobs, np.int32)
        dst = self.zeros(dst_len, nb.dtype)
        nbd = self.up(nb)
        for log2 in (2, 3, 4, 5):   # one launch per block size (the reference's table index)
            sel = jobs[jobs[:, 2] == log2]
            if len(sel):
                self.intra_d(bd, log2, dst, sd, nbd, self._jobs(sel, 8))
        return self.down(dst, nb.dtype)

    def intra_satd35(self, bd, src, ss, nb, jobs):
        """jobs: int32 [n, 8] = (src_off, nb_off, nbf_off, filt_lo, filt_hi, edge, log2, 0); returns int32 [n, 35]"""
        jobs = np.asarray(jobs, np.int32)
        out = np.zeros((len(jobs), 35), np.int32)
        s, nbd = self.up(src), self.up(nb)
        for log2 in (2, 3, 4, 5):
            idx = np.flatnonzero(jobs[:, 6] == log2)
            if len(idx):
                cost = self.zeros(35 * len(idx), np.int32)
                self.intra_satd35_d(bd, log2, s, ss, nbd, self._jobs(jobs[idx], 8), cost)
                out[idx] = self.down(cost, np.int32).reshape(-1, 35)
        return out

    def residual(self, res_len, sres, res_off, src, ss, pred, sp, jobs):
        res = self.zeros(res_len, np.int16)
        self.residual_d(res, sres, self.up(np.asarray(res_off, np.int32)), self.up(src), ss, self.up(pred), sp, self._jobs(jobs, 4))
        return self.down(res, np.int16)

    @staticmethod
    def _tu_groups(jobs):
        jobs = np.asarray(jobs, np.int32)
        for log2, tr in ((2, 1), (2, 0), (3, 0), (4, 0), (5, 0)):
            sel = jobs[(jobs[:, 4] == log2) & (jobs[:, 5] == tr)]
            if len(sel):
                yield log2, tr, sel

    def transform(self, bd, ncoef, res, stride, jobs):
        co = self.zeros(ncoef, np.int16)
        r = self.up(res)
        for log2, tr, sel in self._tu_groups(jobs):
            self.transform_d(bd, tr, log2, co, r, stride, self._jobs(sel, 4))
        return self.down(co, np.int16)

    def inverse_transform(self, bd, nres, coeffs, jobs):
        res = self.zeros(nres, np.int16)
        c = self.up(coeffs)
        for log2, tr, sel in self._tu_groups(jobs):
            self.inverse_transform_d(bd, tr, log2, res, c, self._jobs(sel, 4))
        return self.down(res, np.int16)

    def inverse_transform_add(self, bd, dst_len, sd, pred, sp, coeffs, jobs):
        dst = self.zeros(dst_len, pred.dtype)
        c = self.up(coeffs)
        p = self.up(pred)
        for log2, tr, sel in self._tu_groups(jobs):
            self.inverse_transform_add_d(bd, tr, log2, dst, sd, p, sp, c, self._jobs(sel, 4))
        return self.down(dst, pred.dtype)

    def quantize(self, nout, src, jobs):
        dst = self.zeros(nout, np.int16)
        cbf = self.zeros(len(jobs), np.int32)
        self.quantize_d(dst, self.up(src), self._jobs(jobs, 8), cbf)
        return self.down(dst, np.int16), self.down(cbf, np.int32)

    def quantize_inverse(self, nout, src, jobs):
        dst = self.zeros(nout, np.int16)
        self.quantize_inverse_d(dst, self.up(src), self._jobs(jobs, 8))
        return self.down(dst, np.int16)

    def quantize_reconstruct(self, dst_len, sr, pred, sp, res, jobs):
        jobs = np.asarray(jobs, np.int32)
        rec = self.zeros(dst_len, np.uint8)
        p = self.up(pred)
        r = self.up(res)
        for log2 in (2, 3, 4, 5):
            sel = jobs[jobs[:, 4] == log2]
            if len(sel):
                self.quantize_reconstruct_d(log2, rec, sr, p, sp, r, self._jobs(sel, 4))
        return self.down(rec, np.uint8)
