/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement ("oracle") of the havoc primitive layer of bbc/turingcodec.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the library built from
 * this file; the product (libhavoc_mi355x.so, turingcodec_amd/) never includes, links or calls it.
 *
 * Parity status: PINNED.  Every function below is checked (tests/test_oracle_vs_reference.py) against the
 * reference's own C functions compiled from /root/reference/havoc into oracle/_ref/libhavoc_ref.so, and
 * against the committed golden vectors in tests/golden/ (generated from that build by
 * tests/golden/make_golden.py).
 *
 * All sample pointers are `const void*` + `S` = bytes per sample (1 = uint8_t, 2 = uint16_t); strides are
 * in samples, exactly as in the reference's function types.
 */
#ifndef HAVOC_ORACLE_H
#define HAVOC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* havoc/sad.cpp:432-449 (havoc_sad_c_ref); 16-bit result >>2 */
int oracle_sad(const void *src, intptr_t stride_src, const void *ref, intptr_t stride_ref, int w, int h, int S);
/* havoc/sad.cpp:513-542 (havoc_sad_multiref_4_c_ref) */
void oracle_sad4(const void *src, intptr_t stride_src, const void *const ref[4], intptr_t stride_ref, int sad[4], int w, int h, int S);
/* havoc/ssd.cpp:28-43 (havoc_ssd_c_ref); uint32 accumulate, 16-bit result >>4 */
uint32_t oracle_ssd(const void *a, intptr_t stride_a, const void *b, intptr_t stride_b, int w, int h, int S);
/* havoc/hadamard.cpp:58-98 (compute_satd_c_ref<n>), n = 2, 4, 8; 16-bit result >>2 */
int oracle_satd(const void *a, intptr_t stride_a, const void *b, intptr_t stride_b, int n, int S);
/* havoc/diff.cpp:29-39 (havoc_ssd_linear_c_ref) */
int oracle_ssd_linear(const uint8_t *a, const uint8_t *b, int n);

/* havoc/pred_inter.cpp:76-202 (copy / 8tap_h / 8tap_v / 8tap_hv and the 4-tap forms); taps = 8 or 4 */
void oracle_pred_uni(void *dst, intptr_t stride_dst, const void *ref, intptr_t stride_ref, int w, int h, int xFrac, int yFrac, int bitDepth, int taps, int S);
/* havoc/pred_inter.cpp:1207-1252 (havocPredBi_c_ref + havoc_pred_bi_mean_c_ref) */
void oracle_pred_bi(void *dst, intptr_t stride_dst, const void *ref0, const void *ref1, intptr_t stride_ref, int w, int h, int xFrac0, int yFrac0, int xFrac1, int yFrac1, int bitDepth, int taps, int S);
/* havoc/pred_inter.cpp:2063-2080 (subtractBi_c_ref) */
void oracle_subtract_bi(void *dst, intptr_t stride_dst, const void *pred, intptr_t stride_pred, const void *src, intptr_t stride_src, int w, int h, int bitDepth, int S);

/* havoc/pred_intra.cpp:20282-20401 (predictPlanar / predictDC / predictAngular); neighbours layout
 * pred_intra.cpp:43-51: p(x,y) = neighbours[x - y - 1].  edge = (cIdx == 0 && log2 < 5), pred_intra.h:41-48. */
void oracle_intra(void *dst, intptr_t stride_dst, const void *neighbours, int log2, int mode, int edge, int bitDepth, int S);

/* havoc/transform.cpp:3071-3397 (forward, wraps to int16); trType 1 = DST 4x4 */
void oracle_transform(int16_t *coeffs, const int16_t *src, intptr_t stride_src, int log2, int trType, int bitDepth);
/* havoc/transform.cpp:50-355 (inverse, saturating) */
void oracle_inverse_transform(int16_t *dst, const int16_t *coeffs, int log2, int trType, int bitDepth);
/* havoc/transform.cpp:358-401 + transform.h:95-114 (inverse + add + clip); pred may alias dst */
void oracle_inverse_transform_add(void *dst, intptr_t stride_dst, const void *pred, intptr_t stride_pred, const int16_t *coeffs, int log2, int trType, int bitDepth, int S);

/* havoc/quantize.cpp:37-46 */
void oracle_quantize_inverse(int16_t *dst, const int16_t *src, int scale, int shift, int n);
/* havoc/quantize.cpp:278-304; returns OR of all outputs */
int oracle_quantize(int16_t *dst, const int16_t *src, int scale, int shift, int offset, int n);
/* havoc/quantize.cpp:538-549 (8-bit only) */
void oracle_quantize_reconstruct(uint8_t *rec, intptr_t stride_rec, const uint8_t *pred, intptr_t stride_pred, const int16_t *res, int n);

/* turing/Reconstruct.cpp:258-260, 1274-1286: res = src - pred (the "residual diff" of the north star) */
void oracle_pad_block(void *p, int w, int h, intptr_t stride, int pad, int top, int bottom, int left, int right, int S);
void oracle_residual(int16_t *res, intptr_t stride_res, const void *src, intptr_t stride_src, const void *pred, intptr_t stride_pred, int w, int h, int S);

/* turing/LoopFilter.h:229-400, 739-777 + TaskDeblock.cpp:105-127: in-place deblocking of a 4:2:0 picture.  luma / cb / cr point at
 * sample (0, 0); block_data[i] = (QpY << 1) | filter-disabled, block_bs[i] = 2-bit strengths (vertical pos 0, 1, horizontal pos
 * 0, 1) on the ((width+63)/64*8 + 1)-wide grid of 8x8 luma regions */
void oracle_deblock(void *luma, intptr_t stride_y, void *cb, void *cr, intptr_t stride_c, int width, int height, int bitDepth,
                    const int8_t *block_data, const uint8_t *block_bs, int tc_offset_div2, int beta_offset_div2, int cb_qp_offset,
                    int cr_qp_offset, int S);

/* turing/LoopFilter.h:402-422 (sameMotion) and :541-737 (processCu / processPu / processTu / processRc), restated per 4x4 luma cell:
 * cells = 16-byte records of include/havoc_mi355x.h (havoc_mi355x_cell), cells_stride per row; outputs the two arrays of LoopFilter::Block
 * on the grid of ((width + 63) / 64 * 8 + 1) x ((height + 63) / 64 * 8 + 1) regions */
void oracle_derive_bs(const void *cells, intptr_t cells_stride, int width, int height, int8_t *block_data, uint8_t *block_bs);

/* turing/IntraReferenceSamples.h:373-421 (IntraReferenceSamples::filter; HEVC 8.4.4.2.3): the filtered copy of a block's 4 * nTbS + 1 reference samples.
 * p / pF point at the MIDDLE of the linear arrays the intra functions read (havoc/pred_intra.cpp:43-51): p(x, y) = p[x - y - 1], i.e. [-1 - 2n, -2] = left
 * column from the bottom, [-1] = corner, [0, 2n - 1] = row above.  strong = strong_intra_smoothing_enabled_flag. */
void oracle_intra_filter_neighbours(const int32_t *p, int32_t *pF, int nTbS, int bitDepthY, int strong);
/* HEVC 8.4.4.2.2 as IntraReferenceSamples::substitute (turing/IntraReferenceSamples.h:286-346) applies it, on the array index k = 0 .. 4n (k = t + 2n + 1 of
 * the layout above: from the bottom of the left column to the end of the row above): have[k] != 0 where the sample is available; the others are filled in */
void oracle_intra_substitute(int32_t *val, const uint8_t *have, int nTbS, int bitDepthY);

/* turing/Measure.h:97-135 (measureSatd): PU SATD tiled in 8x8 / 4x4 / 2x2 Hadamards */
int oracle_pu_satd(const void *a, intptr_t stride_a, const void *b, intptr_t stride_b, int w, int h, int S);

/* turing/Rdoq.cpp:37-454 + Rdoq.h:163-187 (oracle/rdoq_oracle.c): rate-distortion optimised quantisation of one transform block.
 * `states` = 128 bytes of CABAC probability states in the layout of include/havoc_mi355x.h (HAVOC_RDOQ_CTX_*); lambdaQ16 / sdhFactor
 * from oracle_rdoq_lambda.  Returns the OR of the kept absolute levels. */
int oracle_rdoq(int16_t *dst, const int16_t *src, int log2Size, int cIdx, int scanIdx, int isIntra, int sdh, int quantScale, int quantShift,
                int invScale, int bitDepth, int32_t lambdaQ16, int32_t sdhFactor, const uint8_t *states);
void oracle_rdoq_lambda(double lambda, int invQuantScale, int32_t *lambdaQ16, int32_t *sdhFactor);
/* turing/ScanOrder.h:212-223 */
int oracle_scan_order(int log2BlockSize, int scanIdx, int sPos, int sComp);

/* turing/EncSao.h:62-283 and turing/sao.cpp:33-92 (oracle/sao_oracle.c): the statistics and the two filters of sample-adaptive offset */
void oracle_sao_stats(const void *src, intptr_t ss, const void *rec, intptr_t rs, int w, int h, int shift, int S, int64_t *out /* [105] */);
void oracle_sao_band_chroma(const void *src_u, const void *src_v, intptr_t ss, const void *rec_u, const void *rec_v, intptr_t rs, int w, int h, int shift, int S,
                            int64_t *out /* [65] */);
void oracle_sao_filter(void *dst, intptr_t ds, const void *src, intptr_t ss, int w, int h, int type, int eoClass, const int16_t *offsets, int bitDepth, int S);

#ifdef __cplusplus
}
#endif
#endif
