// TEST INFRASTRUCTURE ONLY (see ref_shim.cpp).  The reference's OWN rate-distortion optimised quantiser: turing/Rdoq.cpp,
// turing/ScanOrder.cpp and the CABAC initialisation tables of turing/Cabac.cpp are compiled from where they lie
// (oracle/Makefile, target `ref`); this file only builds the `Contexts` object the class reads its probability states
// from, out of a flat array of state bytes, and calls Rdoq::runQuantisation the way turing/Reconstruct.cpp:289-312 (intra)
// and :794-812 (inter) do.  Nothing of the reference is copied.
//
// Flat context layout (the same one include/havoc_mi355x.h documents for havoc_mi355x_rdoq):
//   [0] rqt_root_cbf  [1..2] cbf_luma  [3..6] cbf_cb/cbf_cr  [8..25] last_sig_coeff_x_prefix  [26..43] last_sig_coeff_y_prefix
//   [44..47] coded_sub_block_flag  [48..91] sig_coeff_flag  [92..115] coeff_abs_level_greater1_flag  [116..121] ..greater2_flag
#include "turing/Rdoq.h"
#include <cstdint>
#include <cstring>

namespace {

template <class Tag> void put(Contexts &c, const uint8_t *s, int n) { for (int i = 0; i < n; ++i) c.get<Tag>(i).state = s[i]; }
template <class Tag> void take(Contexts &c, uint8_t *s, int n) { for (int i = 0; i < n; ++i) s[i] = c.get<Tag>(i).state; }

void fill(Contexts &c, const uint8_t *s)
{
    put<rqt_root_cbf>(c, s + 0, 1);
    put<cbf_luma>(c, s + 1, 2);
    put<cbf_cX>(c, s + 3, 4);
    put<last_sig_coeff_x_prefix>(c, s + 8, 18);
    put<last_sig_coeff_y_prefix>(c, s + 26, 18);
    put<coded_sub_block_flag>(c, s + 44, 4);
    put<sig_coeff_flag>(c, s + 48, 44);
    put<coeff_abs_level_greater1_flag>(c, s + 92, 24);
    put<coeff_abs_level_greater2_flag>(c, s + 116, 6);
}

}

extern "C" int ref_rdoq(int16_t *dst, const int16_t *src, int log2Size, int cIdx, int scanIdx, int isIntra, int sdh, int quantScale, int quantShift,
                        int invScale, int bitDepth, double lambda, const uint8_t *states)
{
    static thread_local Contexts contexts;
    fill(contexts, states);
    Rdoq engine(lambda, &contexts, quantScale, invScale, log2Size, bitDepth);
    residual_coding rc(0, 0, log2Size, cIdx);
    return engine.runQuantisation(dst, src, quantScale, quantShift, 1 << 2 * log2Size, rc, scanIdx, !!isIntra, !!sdh);
}

// the states a slice starts from (turing/Cabac.h:415 Contexts::initialize -> ContextModel(qp, initValue), ContextModel.h:43-57)
extern "C" void ref_rdoq_initial_states(int qp, int initType, uint8_t *states)
{
    Contexts c;
    c.initialize(qp, initType);
    std::memset(states, 0, 128);
    take<rqt_root_cbf>(c, states + 0, 1);
    take<cbf_luma>(c, states + 1, 2);
    take<cbf_cX>(c, states + 3, 4);
    take<last_sig_coeff_x_prefix>(c, states + 8, 18);
    take<last_sig_coeff_y_prefix>(c, states + 26, 18);
    take<coded_sub_block_flag>(c, states + 44, 4);
    take<sig_coeff_flag>(c, states + 48, 44);
    take<coeff_abs_level_greater1_flag>(c, states + 92, 24);
    take<coeff_abs_level_greater2_flag>(c, states + 116, 6);
}

extern "C" int ref_scan_order(int log2BlockSize, int scanIdx, int sPos, int sComp) { return ScanOrder(log2BlockSize, scanIdx, sPos, sComp); }

// bench.py's cpu_baseline leg: jobs [b, e) of a table of 48-byte records laid out like havoc_mi355x_rdoq_job (the integer lambda
// fields are ignored: the reference takes the double), one lambda for the picture, `states` = 128-byte snapshots
struct RunJob
{
    int32_t dst_off, src_off, quant_scale, quant_shift, inv_scale, lambda_q16, sdh_factor, ctx_index;
    uint8_t c_idx, scan_idx, is_intra, sdh;
    int32_t reserved[3];
};
extern "C" void ref_run_rdoq(int bitDepth, int log2Size, int16_t *dst, const int16_t *src, const uint8_t *states, const void *jobs, double lambda,
                             int b, int e, int32_t *cbf)
{
    static_assert(sizeof(RunJob) == 48, "job layout");
    const RunJob *j = static_cast<const RunJob *>(jobs);
    for (int i = b; i < e; ++i)
        cbf[i] = ref_rdoq(dst + j[i].dst_off, src + j[i].src_off, log2Size, j[i].c_idx, j[i].scan_idx, j[i].is_intra, j[i].sdh, j[i].quant_scale,
                          j[i].quant_shift, j[i].inv_scale, bitDepth, lambda, states + 128 * (long)j[i].ctx_index);
}
