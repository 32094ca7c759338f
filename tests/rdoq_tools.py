"""Synthetic transform blocks for the RDOQ parity tests (tests/test_rdoq.py, bench.py): coefficient statistics, quantiser
parameters and lambdas in the ranges the reference's encoder produces, plus adversarial ones.  Test infrastructure."""
import numpy as np

QUANT_SCALE = [26214, 23302, 20560, 18396, 16384, 14564]     # HEVC quantiser / de-quantiser scales per QP % 6
INV_SCALE = [40, 45, 51, 57, 64, 72]

JOB_FIELDS = ("log2", "c_idx", "scan_idx", "is_intra", "sdh", "quant_scale", "quant_shift", "inv_scale", "bit_depth", "lam", "ctx_index")


def quant_params(qp, log2, bd):
    """(quantiserScale, quantiserShift, invQuantScale) as turing/Reconstruct.cpp:783-790 computes them"""
    return QUANT_SCALE[qp % 6], 29 - bd + qp // 6 - log2, INV_SCALE[qp % 6] << (qp // 6)


def make_blocks(seed, log2, bd, count, n_states=7, initial_states=None):
    """-> (src int16 [count * n * n], states uint8 [n_states, 128], list of per-block dicts (JOB_FIELDS + src_off))"""
    rng = np.random.default_rng(seed)
    n = 1 << log2
    states = rng.integers(0, 126, (n_states, 128)).astype(np.uint8)
    if initial_states is not None:
        for k, s in enumerate(initial_states[:n_states]):
            states[k] = s
    fy, fx = np.mgrid[0:n, 0:n]
    src = np.zeros(count * n * n, np.int16)
    blocks = []
    for i in range(count):
        c_idx = int(rng.integers(0, 3)) if log2 < 5 else 0
        scan = int(rng.integers(0, 3)) if (log2 <= 3 or i % 7 == 0) else 0
        qp = int(rng.integers(10, 46))
        qs, shift, inv = quant_params(qp, log2, bd)
        lam = 0.57 * 2 ** ((qp - 12) / 3.0) * float(rng.uniform(0.5, 2))
        step = 2.0 ** shift / qs
        amp = step * float(rng.choice([0.3, 1, 3, 10, 40])) * np.exp(-(fx + fy) / float(rng.choice([1.5, 4, 12, 50])))
        blk = np.clip(np.rint(rng.laplace(0, 1, (n, n)) * amp), -32768, 32767).astype(np.int16)
        if i % 50 == 7:
            blk = rng.integers(-32768, 32768, (n, n)).astype(np.int16)      # full range, every cost path saturated
        if i % 50 == 8:
            blk[...] = 0
        if i % 50 == 9:
            blk[...] = 0
            blk[n - 1, n - 1] = int(step * 3)                                # a lone coefficient in the last scan position
        src[i * n * n:(i + 1) * n * n] = blk.ravel()
        blocks.append(dict(log2=log2, c_idx=c_idx, scan_idx=scan, is_intra=int(rng.integers(0, 2)), sdh=int(rng.integers(0, 2)), quant_scale=qs,
                           quant_shift=shift, inv_scale=inv, bit_depth=bd, lam=lam, ctx_index=int(rng.integers(0, n_states)), src_off=i * n * n))
    return src, states, blocks


def run_cpu(lib, src, states, blocks):
    """lib = reflibs.Oracle() or reflibs.Reference(); -> (levels int16 like src, cbf int32)"""
    dst = np.zeros_like(src)
    cbf = np.zeros(len(blocks), np.int32)
    for i, b in enumerate(blocks):
        n2 = 1 << 2 * b["log2"]
        o = b["src_off"]
        d, r = lib.rdoq(np.ascontiguousarray(src[o:o + n2]), b["log2"], b["c_idx"], b["scan_idx"], b["is_intra"], b["sdh"], b["quant_scale"],
                        b["quant_shift"], b["inv_scale"], b["bit_depth"], b["lam"], np.ascontiguousarray(states[b["ctx_index"]]))
        dst[o:o + n2] = d
        cbf[i] = r
    return dst, cbf


def device_jobs(blocks, lambda_fn):
    """RDOQ_JOB_DT records for Havoc.rdoq; lambda_fn(lam, inv_scale) -> (lambda_q16, sdh_factor)"""
    from turingcodec_amd.havoc import RDOQ_JOB_DT
    j = np.zeros(len(blocks), RDOQ_JOB_DT)
    for i, b in enumerate(blocks):
        lq, sf = lambda_fn(b["lam"], b["inv_scale"])
        j[i] = (b["src_off"], b["src_off"], b["quant_scale"], b["quant_shift"], b["inv_scale"], lq, sf, b["ctx_index"], b["c_idx"], b["scan_idx"],
                b["is_intra"], b["sdh"], (0, 0, 0))
    return j
