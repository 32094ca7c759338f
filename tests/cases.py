"""Seeded case generators shared by the oracle-vs-reference tests, the golden-vector script and the GPU parity
tests.  Shapes and value ranges follow the reference's own self-tests:
  sad/ssd inputs rand()&0x3ff (havoc/sad.cpp:1117-1118, ssd.cpp:295-296), satd A near max / B near 0
  (hadamard.cpp:875-876), dequant scale {51, 52224} (quantize.cpp:266-270), quant scale 51 shift 20 offset 14
  (quantize.cpp:523-525), inverse-transform coeffs in [-128,127] (transform.cpp:3049), forward-transform residual
  in [-256,255] (transform.cpp:5360-5361), intra: all 35 modes x 4 sizes x bit depths 8/9/10
  (pred_intra.cpp:22096-22115), inter: all PU shapes x {8,4}-tap x {copy,H,V,HV} x {8,9,10}-bit
  (pred_inter.cpp:1113-1187).
"""
import numpy as np

# the 23 HEVC PU sizes of havoc/sad.h:28-51
PU_SIZES = [(64, 64), (64, 48), (64, 32), (64, 16), (48, 64), (32, 64), (32, 32), (32, 24), (32, 16), (32, 8),
            (24, 32), (16, 64), (16, 32), (16, 16), (16, 12), (16, 8), (16, 4), (12, 16), (8, 32), (8, 16), (8, 8),
            (8, 4), (4, 8)]
# chroma (4:2:0) PU sizes: halves of the above
CHROMA_PU_SIZES = sorted({(w // 2, h // 2) for (w, h) in PU_SIZES}, reverse=True)

PAD = 16          # margin around the area jobs may address (covers the 8-tap reach of -3..+4)
PLANE_W = 160     # plane geometry used by the small parity cases
PLANE_H = 160


def sample_dtype(S):
    return np.uint8 if S == 1 else np.uint16


def aligned(a, align=64):
    """copy of `a` (any shape, C order) whose first byte is `align`-byte aligned -- the reference's SIMD kernels
    use aligned loads on the source operand (psadbw xmm, m128 at havoc/sad.cpp:135-138, vmovdqa at ssd.cpp:122-123)"""
    a = np.ascontiguousarray(a)
    raw = np.empty(a.nbytes + align, np.uint8)
    o = (-raw.ctypes.data) % align
    out = raw[o:o + a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def rand_plane(rng, S, bit_depth=None, h=PLANE_H, w=PLANE_W, kind="uniform"):
    """A (h, w) plane.  kind: uniform | high (near max) | low (near 0) | extremes (only 0 / max)"""
    bd = bit_depth or (8 if S == 1 else 10)
    mx = (1 << bd) - 1
    if kind == "uniform":
        a = rng.integers(0, mx + 1, size=(h, w))
    elif kind == "high":
        a = mx - rng.integers(0, 4, size=(h, w))
    elif kind == "low":
        a = rng.integers(0, 4, size=(h, w))
    elif kind == "extremes":
        a = rng.integers(0, 2, size=(h, w)) * mx
    else:
        raise ValueError(kind)
    return aligned(a.astype(sample_dtype(S)))


def rand_pos(rng, w, h, plane_w=PLANE_W, plane_h=PLANE_H, pad=PAD):
    """random top-left (x, y) such that the block plus `pad` margin stays inside the plane"""
    x = int(rng.integers(pad, plane_w - pad - w + 1))
    y = int(rng.integers(pad, plane_h - pad - h + 1))
    return x, y


def off(x, y, stride=PLANE_W):
    return y * stride + x


def block_pair_cases(rng, sizes, n_per_size=2):
    """[(w, h, a_off, b_off)] with random unaligned positions"""
    out = []
    for (w, h) in sizes:
        for _ in range(n_per_size):
            ax, ay = rand_pos(rng, w, h)
            bx, by = rand_pos(rng, w, h)
            out.append((w, h, off(ax, ay), off(bx, by)))
    return out


def sad4_cases(rng, sizes, n_per_size=2):
    out = []
    for (w, h) in sizes:
        for _ in range(n_per_size):
            sx, sy = rand_pos(rng, w, h)
            refs = [off(*rand_pos(rng, w, h)) for _ in range(4)]
            out.append((w, h, off(sx, sy), refs))
    return out


def pred_uni_cases(rng, bit_depths):
    """[(taps, w, h, xFrac, yFrac, bitDepth, ref_off)]: every shape x {copy,H,V,HV} x bit depth"""
    out = []
    for taps, sizes, nfrac in ((8, PU_SIZES, 4), (4, CHROMA_PU_SIZES, 8)):
        for (w, h) in sizes:
            for bd in bit_depths:
                for kind in range(4):
                    xf = int(rng.integers(1, nfrac)) if kind & 1 else 0
                    yf = int(rng.integers(1, nfrac)) if kind & 2 else 0
                    x, y = rand_pos(rng, w, h)
                    out.append((taps, w, h, xf, yf, bd, off(x, y)))
    return out


def pred_bi_cases(rng, bit_depths):
    """[(taps, w, h, xf0, yf0, xf1, yf1, bitDepth, ref0_off, ref1_off)]"""
    out = []
    for taps, sizes, nfrac in ((8, PU_SIZES, 4), (4, CHROMA_PU_SIZES, 8)):
        for (w, h) in sizes:
            for bd in bit_depths:
                for kind in range(3):
                    if kind == 0:
                        fr = [0, 0, 0, 0]
                    elif kind == 1:
                        fr = [int(v) for v in rng.integers(0, nfrac, size=4)]
                    else:
                        fr = [int(v) for v in rng.integers(1, nfrac, size=4)]
                    x0, y0 = rand_pos(rng, w, h)
                    x1, y1 = rand_pos(rng, w, h)
                    out.append((taps, w, h, *fr, bd, off(x0, y0), off(x1, y1)))
    return out


def intra_cases(bit_depths):
    """[(log2, mode, edge, bitDepth)]: all 35 modes x 4 sizes x edge on/off"""
    return [(log2, mode, edge, bd) for bd in bit_depths for log2 in (2, 3, 4, 5) for mode in range(35)
            for edge in (0, 1)]


def rand_neighbours(rng, S, bit_depth, kind="uniform"):
    """flat array of 129 + slack samples; the neighbours pointer is element 64 + 1 = index 65 is p(0,-1)...
    We return (array, centre) with centre the index of neighbours[0] (= p(0,-1)); neighbours[-1] is the corner."""
    mx = (1 << bit_depth) - 1
    n = 160
    if kind == "uniform":
        a = rng.integers(0, mx + 1, size=n)
    elif kind == "extremes":
        a = rng.integers(0, 2, size=n) * mx
    else:
        a = np.full(n, mx)
    return np.ascontiguousarray(a.astype(sample_dtype(S))), 80


TRANSFORMS = [(2, 1), (2, 0), (3, 0), (4, 0), (5, 0)]  # (log2, trType)


def residual_block(rng, n, lo=-256, hi=255, kind="uniform"):
    if kind == "uniform":
        a = rng.integers(lo, hi + 1, size=(n, n))
    elif kind == "extremes":
        a = rng.integers(0, 2, size=(n, n)) * (hi - lo) + lo
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(a.astype(np.int16))


# turing/QpState.h:85-94
QUANT_SCALE = [26214, 23302, 20560, 18396, 16384, 14564]
DEQUANT_SCALE = [40, 45, 51, 57, 64, 72]


def quant_params(qp, log2, bit_depth, intra_slice):
    """(scale, shift, offset) as turing/Reconstruct.cpp:286,311 / :785,817 pass them to havoc_quantize"""
    scale = QUANT_SCALE[qp % 6]
    shift = 29 - bit_depth + qp // 6 - log2
    # intra path passes offsetQuantiseShifted = 171<<7 (QpState.h:91); the inter path passes
    # (85 << (shift-9)) >> (shift-16) = 85<<7: the argument is always (I-slice ? 171 : 85) << 7
    offset = (171 if intra_slice else 85) << 7
    return scale, shift, offset


def dequant_params(qp, log2, bit_depth):
    """(scale, shift) of turing/QpState.h:85-86, Reconstruct.cpp:315"""
    return DEQUANT_SCALE[qp % 6] << (qp // 6), log2 - 1 + bit_depth - 8
