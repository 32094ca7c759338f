/*
 * havoc_mi355x.h -- C ABI of libhavoc_mi355x.so, the MI355X (gfx950) implementation of the `havoc`
 * primitive layer of the Turing HEVC encoder.
 *
 * Each batch entry point evaluates MANY independent calls of one reference primitive in one kernel launch.
 * A "job" carries what the reference passes per call as pointers (here: sample offsets into planes that
 * already live in HBM) and block geometry; strides and bit depth are per launch.  Results are bit-identical
 * to the reference function cited next to each entry point.
 *
 * Conventions
 *   - every `d_*` pointer is DEVICE memory (hipMalloc / torch.cuda); nothing is copied to or from the host
 *     unless the name says so (`*_h2d`, `*_d2h`, and the classic per-block API in havoc_classic.hpp);
 *   - `S` = bytes per sample: 1 (uint8_t, 8-bit) or 2 (uint16_t, 9/10-bit), the reference's `Sample` type;
 *   - offsets and strides are in SAMPLES (int16 elements for coefficient buffers), exactly like the
 *     reference's pointer arithmetic (havoc/sad.h:58 etc.); offsets are relative to the plane base pointer
 *     passed to the call and may address the 96-sample picture padding (turing/StatePictures.h:155-156);
 *   - launches are asynchronous on the context's HIP stream; call havoc_mi355x_sync() before reading results
 *     on the host.  Every function returns 0 on success or a negative hipError_t / HAVOC_MI355X_E* code;
 *     havoc_mi355x_last_error() gives the text.  The reference has no error channel: a
 *     failure here means the launch did not happen, never that a CPU path was taken.
 */
#ifndef HAVOC_MI355X_H
#define HAVOC_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HAVOC_MI355X_EINVAL (-10001) /* bad argument (S, taps, bit depth, size) */
#define HAVOC_MI355X_ENODEV (-10002) /* no gfx950 device / HIP runtime unavailable */
#define HAVOC_MI355X_EDEVICE (-10003) /* a kernel reported that it could not finish (havoc_mi355x_search_picture_uni: a wait gave up) */

typedef struct havoc_mi355x_ctx havoc_mi355x_ctx;

/* ---- context: replaces havoc_new_code / havoc_delete_code (havoc/havoc.h:138-147, havoc.cpp:144-155).
 * `stream` is a hipStream_t (NULL = the device's default stream; pass torch's current stream to order
 * launches with torch ops).  Fails with ENODEV when there is no GPU: there is no CPU fallback. */
#define HAVOC_MI355X_NEW_STREAM ((void *)(intptr_t)-1) /* create(): make and own a private non-blocking stream */
int havoc_mi355x_create(havoc_mi355x_ctx **ctx, int device, void *stream);
void havoc_mi355x_destroy(havoc_mi355x_ctx *ctx);
int havoc_mi355x_set_stream(havoc_mi355x_ctx *ctx, void *stream);
int havoc_mi355x_sync(havoc_mi355x_ctx *ctx);
/* the same wait for a caller that queued microseconds of work and needs it now (a table call of libhavoc_classic.so): the stream writes a sequence number into pinned
 * memory behind the queued work and the calling thread POLLS it (bounded: falls back to havoc_mi355x_sync after ~2 ms, and with forked lanes) */
int havoc_mi355x_sync_spin(havoc_mi355x_ctx *ctx);
const char *havoc_mi355x_last_error(void);
const char *havoc_mi355x_version(void);
/* device properties for roofline accounting: [0]=CUs [1]=clock kHz [2]=memory clock kHz [3]=bus width bits
 * [4]=L2 bytes [5]=wavefront size [6]=LDS bytes per workgroup [7]=total global memory MiB */
int havoc_mi355x_device_info(havoc_mi355x_ctx *ctx, int64_t info[8]);

/* plain device-memory helpers so that a C/C++ host (the encoder) needs no HIP headers */
int havoc_mi355x_malloc(havoc_mi355x_ctx *ctx, void **d_ptr, size_t bytes);
int havoc_mi355x_free(havoc_mi355x_ctx *ctx, void *d_ptr);
int havoc_mi355x_h2d(havoc_mi355x_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int havoc_mi355x_d2h(havoc_mi355x_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* timing on the context's stream with HIP events (used by bench.py for per-kernel durations) */
int havoc_mi355x_timer_start(havoc_mi355x_ctx *ctx);
int havoc_mi355x_timer_stop_ms(havoc_mi355x_ctx *ctx, float *ms); /* records, synchronises, returns elapsed */

/* pinned host memory the device can address: *h_ptr for the host, *d_ptr for kernels (results a host loop reads right after
 * havoc_mi355x_sync() without a copy); asynchronous and pitched copies on the context's stream */
int havoc_mi355x_host_alloc(havoc_mi355x_ctx *ctx, size_t bytes, void **h_ptr, void **d_ptr);
int havoc_mi355x_host_free(havoc_mi355x_ctx *ctx, void *h_ptr);
int havoc_mi355x_h2d_async(havoc_mi355x_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int havoc_mi355x_d2h_async(havoc_mi355x_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int havoc_mi355x_copy_2d(havoc_mi355x_ctx *ctx, void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t row_bytes, size_t rows,
                         int to_device);

/* HIP-graph capture of a fixed sequence of batch launches (one picture's launches are the same sequence with the same
 * buffers for every CTU row / picture of a size): begin, issue the launches (they are recorded, not run), end, then
 * replay with graph_launch.  Needs a context that owns its stream (HAVOC_MI355X_NEW_STREAM) or an explicit
 * non-default stream: the legacy default stream cannot be captured. */
typedef struct havoc_mi355x_graph havoc_mi355x_graph;
int havoc_mi355x_graph_begin(havoc_mi355x_ctx *ctx);
int havoc_mi355x_graph_end(havoc_mi355x_ctx *ctx, havoc_mi355x_graph **graph);
int havoc_mi355x_graph_launch(havoc_mi355x_ctx *ctx, havoc_mi355x_graph *graph);
void havoc_mi355x_graph_destroy(havoc_mi355x_graph *graph);

/* Fork / join: batches that do not depend on each other (integer ME of one list, sub-pel candidates of another PU
 * class, the intra stage, each transform size ...) are small, latency-bound launches; issued on separate HIP streams
 * they overlap on the 256 CUs.  fork(n) makes n lanes (lane 0 = the context's stream; lanes 1..n-1 are side streams
 * that start after everything already queued), lane(k) selects where the following launches go (launches on one
 * lane stay ordered), join() makes the context's stream wait for every lane.  Works inside graph capture: the graph
 * then holds the parallel branches. */
int havoc_mi355x_fork(havoc_mi355x_ctx *ctx, int nlanes); /* 1..8 */
int havoc_mi355x_lane(havoc_mi355x_ctx *ctx, int lane);
int havoc_mi355x_join(havoc_mi355x_ctx *ctx);

/* ------------------------------------------------------------------------------------------------------- */
/* job descriptors (plain 32-bit little-endian fields, no padding surprises: sizes asserted in the .cpp)    */
/* ------------------------------------------------------------------------------------------------------- */

/* two blocks of w x h samples: SAD(src, ref), SSD(a, b), SATD(a, b), residual(src, pred) */
typedef struct {
    int32_t a_off;   /* src / srcA */
    int32_t b_off;   /* ref / srcB */
    int32_t w, h;
} havoc_mi355x_pair_job; /* 16 bytes */

/* one source block against four candidate reference positions (havoc/sad.h:100) */
typedef struct {
    int32_t src_off;
    int32_t ref_off[4];
    int32_t w, h;
    int32_t reserved;
} havoc_mi355x_sad4_job; /* 32 bytes */

/* one source block against every integer candidate of a (2R+1) x (2R+1) window centred on ref_off */
typedef struct {
    int32_t src_off;
    int32_t ref_off;   /* candidate (dx, dy) = 0: the sample offset of the centre position in the reference plane */
    int32_t w, h;      /* w a multiple of 4 */
    int32_t out_off;   /* index of the surface's first int32 (dx = dy = -R) in d_out */
    int32_t reserved[3];
} havoc_mi355x_surface_job; /* 32 bytes */

/* one block against up to 16 candidate blocks (the 8 half- + 8 quarter-sample candidates of a PU and list) */
typedef struct {
    int32_t a_off;
    int32_t w, h;
    int32_t count;       /* 1..16 candidates used */
    int32_t b_off[16];
} havoc_mi355x_satd_multi_job; /* 80 bytes */

/* fractional-sample interpolation of one prediction block (havoc/pred_inter.h:35) */
typedef struct {
    int32_t dst_off;
    int32_t ref_off;  /* integer-sample position of the block inside the padded reference plane */
    int32_t w, h;
    int32_t xFrac, yFrac; /* quarter (8-tap luma) or eighth (4-tap chroma) sample phase */
    int32_t reserved[2];
} havoc_mi355x_pred_uni_job; /* 32 bytes */

/* bi-prediction from two positions of planes sharing one stride (havoc/pred_inter.h:63) */
typedef struct {
    int32_t dst_off;
    int32_t ref0_off, ref1_off;
    int32_t w, h;
    int32_t xFrac0, yFrac0, xFrac1, yFrac1;
    int32_t reserved[3];
} havoc_mi355x_pred_bi_job; /* 48 bytes */

/* dst = clip(2*src - pred) (havoc/pred_inter.h:87) */
typedef struct {
    int32_t dst_off, pred_off, src_off;
    int32_t w, h;
    int32_t reserved[3];
} havoc_mi355x_subtract_bi_job; /* 32 bytes */

/* one intra prediction block (havoc/pred_intra.h:32-52).  nb_off addresses neighbours[0] = p(0,-1):
 * neighbours[-1] is the corner, [-2 .. -1-2n] the left column top->bottom, [0 .. 2n-1] the top row
 * (havoc/pred_intra.cpp:43-51).  `edge` = (cIdx == 0): the DC / mode-10 / mode-26 edge filters are applied
 * when edge && log2 < 5, as Table::lookup does.  The block size is a launch parameter (the reference's table is
 * indexed by log2TrafoSize); `log2` here is informational and must equal it. */
typedef struct {
    int32_t dst_off;
    int32_t nb_off;
    int32_t log2;    /* 2..5 */
    int32_t mode;    /* 0 planar, 1 DC, 2..34 angular */
    int32_t edge;
    int32_t reserved[3];
} havoc_mi355x_intra_job; /* 32 bytes */

/* one transform unit; size and type are launch parameters (the reference picks one function per
 * (trType, log2TrafoSize), havoc/transform.h:72-84, :128-140).  Unused fields are ignored by each entry point. */
typedef struct {
    int32_t coef_off;  /* n*n contiguous int16 coefficients (index into the coefficient buffer) */
    int32_t res_off;   /* int16 residual block: forward input (row stride = stride_res argument), or n*n
                          contiguous output of inverse_transform / input of quantize_reconstruct */
    int32_t pred_off;  /* prediction samples (inverse+add input) */
    int32_t dst_off;   /* reconstructed samples (inverse+add output); may equal pred_off on the same plane */
} havoc_mi355x_tu_job; /* 16 bytes */

/* one (de)quantisation call over n contiguous int16 values (havoc/quantize.h:42,63) */
typedef struct {
    int32_t dst_off, src_off;
    int32_t n;        /* multiple of 16, as the reference requires */
    int32_t scale, shift, offset; /* offset ignored by the inverse quantiser */
    int32_t reserved[2];
} havoc_mi355x_quant_job; /* 32 bytes */

/* ------------------------------------------------------------------------------------------------------- */
/* picture-level helper next to the path                                                                     */
/* ------------------------------------------------------------------------------------------------------- */

/* Padding::padBlock<Sample> (turing/Padding.h:60-97; all four flags set = Padding::padImage :33-57): replicate the edge
 * samples of the width x height block whose sample (0, 0) sits at d_plane[origin_off] into a border of `pad` samples on
 * the requested sides, in place.  What TaskDeblock.cpp:151-159 does to a reconstructed picture before it is used
 * as a reference (and what the owner runs before the frame-parallel broadcast). */
int havoc_mi355x_pad_block(havoc_mi355x_ctx *ctx, int S, void *d_plane, int64_t origin_off, int width, int height,
                           intptr_t stride, int pad, int top, int bottom, int left, int right);

/* In-loop deblocking of a whole 4:2:0 picture, in place: LoopFilter::Picture::deblock<EDGE_VER> then
 * <EDGE_HOR> (turing/LoopFilter.h:229-400, 739-777) over every 8x8 region, which is what the reference's CTU-by-CTU order
 * (turing/TaskDeblock.cpp:105-127) amounts to.  d_luma / d_cb / d_cr point at sample (0, 0) of each plane.  The boundary
 * strengths and QPs -- the encoder's decisions (LoopFilter.h:480-737) -- come as the two arrays of LoopFilter::Block on its
 * grid of ((width + 63) / 64 * 8 + 1) x ((height + 63) / 64 * 8 + 1) regions: d_block_data[i] = (QpY << 1) | filter-disabled,
 * d_block_bs[i] = 2-bit strengths, bits 0-1 / 2-3 = the region's left edge rows 0-3 / 4-7, bits 4-5 / 6-7 = its top edge
 * columns 0-3 / 4-7 (chroma uses the first of each pair, as the reference does).  One slice: the offsets are per picture. */
int havoc_mi355x_deblock(havoc_mi355x_ctx *ctx, int S, int bitDepth, void *d_luma, intptr_t stride_luma, void *d_cb, void *d_cr, intptr_t stride_chroma,
                         int width, int height, const int8_t *d_block_data, const uint8_t *d_block_bs, int tc_offset_div2, int beta_offset_div2,
                         int cb_qp_offset, int cr_qp_offset);

/* The boundary strengths and QP bytes havoc_mi355x_deblock reads, DERIVED on the device from the picture's block structure
 * (LoopFilter::Picture::processCu / processPu / processTu / processRc, turing/LoopFilter.h:541-737, and sameMotion, :402-422) -- the
 * encoder's decisions as a map of 4x4 luma cells, which is what a device-side encoder holds after mode decision:
 *   a transform block edge on the 8-sample grid gets strength 2 where the block on either side is intra, 1 where either side is an
 *   inter block with coded luma coefficients; the left / top edge of an inter prediction unit gets 1 where the motion across it differs
 *   (different pictures, or a vector component 4 or more quarter samples apart, lists matched either way round); the maximum stands.
 * A cell outside the picture counts as "no motion" (what the reference's unavailable neighbour is); edges on the picture's left and top
 * boundary end with strength 0 (processCtu, LoopFilter.h:484-510, runs after the CTU's units; one slice per picture).  PCM units are not modelled (the
 * reference's encoder never emits them).  Writes the whole grid of ((width + 63) / 64 * 8 + 1) x ((height + 63) / 64 * 8 + 1) regions. */
typedef struct {
    int16_t mv[2][2];     /* [list][x, y], quarter samples */
    int8_t dpb_index[2];  /* which decoded picture each list predicts from; -1 = list unused (both -1: intra / no motion) */
    uint8_t flags;        /* HAVOC_CELL_* */
    int8_t qp_y;
    uint8_t tu_log2;      /* log2 size of the luma transform block holding the cell (2..5) */
    uint8_t reserved[3];
} havoc_mi355x_cell; /* 16 bytes */
enum {
    HAVOC_CELL_INTRA = 1,      /* CuPredMode == MODE_INTRA */
    HAVOC_CELL_CODED = 2,      /* cbf_luma of its transform block */
    HAVOC_CELL_NO_FILTER = 4,  /* cu_transquant_bypass_flag (Block::packData's disable bit) */
    HAVOC_CELL_PU_LEFT = 8,    /* the cell lies on the left edge of its prediction unit */
    HAVOC_CELL_PU_TOP = 16     /* ... on its top edge */
};
int havoc_mi355x_derive_bs(havoc_mi355x_ctx *ctx, const havoc_mi355x_cell *d_cells, intptr_t cells_stride, int width, int height, int8_t *d_block_data,
                           uint8_t *d_block_bs);

/* Device-side picture store + input upload.  A picture = the three planes of Picture<Sample>
 * (turing/Picture.cpp:91-125) in ONE HBM allocation: per plane `pad` samples of border (chroma: pad / 2), row stride rounded
 * up to `alignment` bytes (the reference: pad 96, alignment 32, turing/StatePictures.h:155-156; this library's kernels are
 * indifferent, 64 keeps rows on 64-byte boundaries).  4:2:0 only, as every configuration of BASELINE.json. */
typedef struct havoc_mi355x_picture havoc_mi355x_picture;
int havoc_mi355x_picture_create(havoc_mi355x_ctx *ctx, int S, int bit_depth, int width, int height, int pad, int alignment,
                                havoc_mi355x_picture **pic);
void havoc_mi355x_picture_destroy(havoc_mi355x_ctx *ctx, havoc_mi355x_picture *pic);
/* plane geometry: *d_base = the picture's allocation (one for the three planes), *origin_off = sample offset of sample (0, 0)
 * of plane cIdx from d_base (what a job's offsets are relative to), *stride in samples; any out pointer may be NULL */
int havoc_mi355x_picture_plane(havoc_mi355x_picture *pic, int cIdx, void **d_base, int64_t *origin_off, intptr_t *stride, int *width, int *height,
                               int *pad);
/* one input frame as the reference reads it (turing/encode.cpp:600-640): planar Y, U, V, tightly packed, src_S bytes per sample
 * (16-bit: little-endian words); stored << shift (encode.cpp:397: 8-bit input on the 16-bit path, shift 2); pad_after != 0
 * replicates the borders afterwards (Padding::padImage, turing/Padding.h:33-57) */
int havoc_mi355x_picture_upload_yuv(havoc_mi355x_ctx *ctx, havoc_mi355x_picture *pic, const void *h_yuv, int src_S, int shift, int pad_after);
/* one plane from / to a host Picture plane: h_origin = its sample (0, 0), h_stride in samples; with_padding != 0 moves the
 * `pad` border as well (the host plane must have one at least as wide).  download synchronises. */
int havoc_mi355x_picture_upload_plane(havoc_mi355x_ctx *ctx, havoc_mi355x_picture *pic, int cIdx, const void *h_origin, intptr_t h_stride,
                                      int with_padding);
int havoc_mi355x_picture_download_plane(havoc_mi355x_ctx *ctx, havoc_mi355x_picture *pic, int cIdx, void *h_origin, intptr_t h_stride,
                                        int with_padding);
int havoc_mi355x_picture_pad(havoc_mi355x_ctx *ctx, havoc_mi355x_picture *pic);
/* the 16 fractional-sample luma planes of the picture (havoc_mi355x_interp_planes; slot 0 = the luma plane), made on first
 * request after the picture changed and kept with it.  Phase k, sample (x, y):
 *   d_planes[k * plane_elems + (luma origin_off - *luma_first) + y * stride + x]   (luma plane's origin_off / stride). */
int havoc_mi355x_picture_phase_planes(havoc_mi355x_ctx *ctx, havoc_mi355x_picture *pic, void **d_planes, intptr_t *plane_elems, int64_t *luma_first);

/* ------------------------------------------------------------------------------------------------------- */
/* distortion metrics                                                                                        */
/* ------------------------------------------------------------------------------------------------------- */

/* havoc_sad<Sample> (havoc/sad.h:58, C reference havoc/sad.cpp:432-449): d_out[i] = SAD, 16-bit >> 2 */
int havoc_mi355x_sad(havoc_mi355x_ctx *ctx, int S, const void *d_src, intptr_t stride_src, const void *d_ref,
                     intptr_t stride_ref, const havoc_mi355x_pair_job *d_jobs, int njobs, int32_t *d_out);
/* havoc_sad_multiref<Sample>, ways = 4 (havoc/sad.h:100, havoc/sad.cpp:513-542): d_out[4*i + k] */
int havoc_mi355x_sad4(havoc_mi355x_ctx *ctx, int S, const void *d_src, intptr_t stride_src, const void *d_ref,
                      intptr_t stride_ref, const havoc_mi355x_sad4_job *d_jobs, int njobs, int32_t *d_out);
/* The same calls BY RUNS: a run = consecutive havoc_sad_multiref calls of one motion search (turing/Search.hpp:2224-2297 -> considerPattern :1447-1482: ~112 calls
 * sharing the source block and moving around one centre).  A workgroup stages the run's source block and the bounding box of all its candidates in LDS once and the
 * calls read them there; d_out as havoc_mi355x_sad4.  d_runs must tile the jobs the caller wants computed: the d_out entries of a job in no run are LEFT AS THEY ARE
 * (clear d_out first if that matters), and runs must not overlap (two workgroups would write the same entries).  havoc_mi355x_sad4_make_runs' output tiles [0, njobs).  Runs change the speed,
 * never a result: a run whose calls differ in source / size, whose box does not fit LDS or does not hold every candidate, is computed call by call.
 *   box_off / box_w / box_h: the rectangle of the reference plane (sample offset of its top-left from d_ref, width in samples, rows) that holds every candidate BLOCK of
 *   the run -- staged at once, each candidate then checked against it; box_w = 0: the kernel finds the box itself (one more pass over the run's jobs).
 * havoc_mi355x_sad4_make_runs (host code, no device) cuts a HOST copy of a job table into runs and gives them their boxes: consecutive jobs with equal src_off, w, h whose
 * box fits the kernel's window (16 KB x S), at most max_run calls (1 .. 128; <= 0: by block size -- 16 calls of a 64x64 block, 48 of a 32x32, 128 below: what keeps a
 * workgroup's work even, profiles/r05/sad4_run_policy.txt); stride_ref = the reference plane's row stride in samples (the boxes are rectangles of that plane).
 * Returns the number of runs written to `runs` (capacity njobs), or -1. */
typedef struct {
    int32_t first_job;
    int32_t count;
    int32_t box_off, box_w, box_h;
    int32_t reserved[3];
} havoc_mi355x_sad4_run; /* 32 bytes */
int havoc_mi355x_sad4_runs(havoc_mi355x_ctx *ctx, int S, const void *d_src, intptr_t stride_src, const void *d_ref,
                           intptr_t stride_ref, const havoc_mi355x_sad4_job *d_jobs, int njobs,
                           const havoc_mi355x_sad4_run *d_runs, int nruns, int32_t *d_out);
int havoc_mi355x_sad4_make_runs(const havoc_mi355x_sad4_job *jobs, int njobs, int max_run, intptr_t stride_ref, int S, havoc_mi355x_sad4_run *runs);
/* Full-pel SAD surface: the super-set serving the havoc_sad / havoc_sad_multiref calls of the integer motion search
 * (turing/Search.hpp:1447-1482 considerPattern, :1585-1623 bi grid, :2224-2290 star / raster / refinement) as look-ups.
 * d_out[out_off + (dy + range) * (2*range + 1) + (dx + range)] = havoc_sad(src, ref + dy*stride_ref + dx) for every
 * dx, dy in [-range, range], range 0..96 (the reference pads its pictures by 96 samples); every value is the one havoc_mi355x_sad returns for that candidate.
 * max_w / max_h: upper bounds on the block sizes of the batch.  Reads at most 3 bytes beyond a candidate row. */
int havoc_mi355x_sad_surface(havoc_mi355x_ctx *ctx, int S, int range, int max_w, int max_h, const void *d_src,
                             intptr_t stride_src, const void *d_ref, intptr_t stride_ref,
                             const havoc_mi355x_surface_job *d_jobs, int njobs, int32_t *d_out);
/* havoc_ssd<Sample> (havoc/ssd.h:33, havoc/ssd.cpp:28-43): uint32 accumulate, 16-bit >> 4 */
int havoc_mi355x_ssd(havoc_mi355x_ctx *ctx, int S, const void *d_a, intptr_t stride_a, const void *d_b,
                     intptr_t stride_b, const havoc_mi355x_pair_job *d_jobs, int njobs, uint32_t *d_out);
/* havoc_hadamard_satd<Sample> tiled over a w x h block exactly as measureSatd does (turing/Measure.h:97-135;
 * one 2x2 / 4x4 / 8x8 Hadamard = havoc/hadamard.cpp:58-98 when w == h == n).  max_w / max_h: upper bounds on the
 * block sizes of this batch (64, 64 always valid): they choose how many lanes share a job. */
int havoc_mi355x_satd(havoc_mi355x_ctx *ctx, int S, int max_w, int max_h, const void *d_a, intptr_t stride_a, const void *d_b,
                      intptr_t stride_b, const havoc_mi355x_pair_job *d_jobs, int njobs, int32_t *d_out);
/* The same measure for one block `a` against `count` candidate blocks `b` (costDistortionMv's candidates of one PU,
 * turing/Search.hpp:1965-1998): d_out[16*i + k], k < count, each equal to what havoc_mi355x_satd returns for that pair. */
int havoc_mi355x_satd_multi(havoc_mi355x_ctx *ctx, int S, int max_w, int max_h, const void *d_a, intptr_t stride_a,
                            const void *d_b, intptr_t stride_b, const havoc_mi355x_satd_multi_job *d_jobs, int njobs,
                            int32_t *d_out);
/* havoc_ssd_linear (havoc/diff.h:35, havoc/diff.cpp:29-39): one linear 8-bit run, *d_out = int32 sum */
int havoc_mi355x_ssd_linear(havoc_mi355x_ctx *ctx, const uint8_t *d_a, const uint8_t *d_b, int size, int32_t *d_out);

/* ------------------------------------------------------------------------------------------------------- */
/* inter prediction                                                                                          */
/* ------------------------------------------------------------------------------------------------------- */

/* HavocPredUni<Sample> (havoc/pred_inter.h:35, C reference havoc/pred_inter.cpp:76-202); taps = 8 | 4.
 * Writes exactly w x h samples per job (the JIT's licence to over-write to the right is not used).  max_w / max_h:
 * upper bounds on the block sizes of this batch -- the reference's table is indexed by width class
 * (havoc/pred_inter.h:47-50); 64, 64 is always valid.  Every job reads the full (w+taps-1) x (h+taps-1) window
 * around its block whatever the phase (the zero phase is evaluated with the {..,64,..} filter). */
int havoc_mi355x_pred_uni(havoc_mi355x_ctx *ctx, int S, int taps, int bitDepth, int max_w, int max_h, void *d_dst, intptr_t stride_dst,
                          const void *d_ref, intptr_t stride_ref, const havoc_mi355x_pred_uni_job *d_jobs, int njobs);
/* HavocPredBi<Sample> (havoc/pred_inter.h:63, havoc/pred_inter.cpp:1207-1252) */
int havoc_mi355x_pred_bi(havoc_mi355x_ctx *ctx, int S, int taps, int bitDepth, int max_w, int max_h, void *d_dst, intptr_t stride_dst,
                         const void *d_ref, intptr_t stride_ref, const havoc_mi355x_pred_bi_job *d_jobs, int njobs);
/* The same two primitives for a job table that holds ALL size classes, in ONE launch: the table is sorted by class -- count[0] jobs whose
 * larger side is <= 8, then count[1] jobs <= 16, count[2] <= 32, count[3] <= 64 -- the width class being a property of the job, as the
 * reference's havocGetPredUni / havocGetPredBi index it (havoc/pred_inter.h:47-50, 72-76).  Results identical to the per-class launches. */
int havoc_mi355x_pred_uni_classes(havoc_mi355x_ctx *ctx, int S, int taps, int bitDepth, void *d_dst, intptr_t stride_dst, const void *d_ref,
                                  intptr_t stride_ref, const havoc_mi355x_pred_uni_job *d_jobs, const int32_t count[4]);
int havoc_mi355x_pred_bi_classes(havoc_mi355x_ctx *ctx, int S, int taps, int bitDepth, void *d_dst, intptr_t stride_dst, const void *d_ref,
                                 intptr_t stride_ref, const havoc_mi355x_pred_bi_job *d_jobs, const int32_t count[4]);
/* havoc::SubtractBi<Sample> (havoc/pred_inter.h:87, havoc/pred_inter.cpp:2063-2080) */
int havoc_mi355x_subtract_bi(havoc_mi355x_ctx *ctx, int S, int bitDepth, void *d_dst, intptr_t stride_dst,
                             const void *d_pred, intptr_t stride_pred, const void *d_src, intptr_t stride_src,
                             const havoc_mi355x_subtract_bi_job *d_jobs, int njobs);

/* The 15 fractional-sample luma planes of a reference picture, computed once per picture: for every (xFrac, yFrac)
 * in 0..3 x 0..3 except (0,0) and every sample (x, y) of the rectangle [x0, x0+width) x [y0, y0+height) of the
 * padded picture,   d_planes[(4*yFrac + xFrac) * plane_elems + y*stride + x] = the sample HavocPredUni (8-tap,
 * havoc/pred_inter.cpp:113-202) produces at (x, y) for that phase.  A sub-pel candidate's prediction block is then a
 * block of one plane and its cost a plain havoc_mi355x_satd job against it (plane 0 is not written: use d_ref).
 * The rectangle must leave >= 4 rows and >= 12 samples of the allocation around it (the picture padding does). */
int havoc_mi355x_interp_planes(havoc_mi355x_ctx *ctx, int S, int bitDepth, void *d_planes, intptr_t plane_elems, const void *d_ref,
                               intptr_t stride, int x0, int y0, int width, int height);

/* One sub-pel motion candidate, fused: HavocPredUni of the PU at (ref_off, xFrac, yFrac) followed by measureSatd
 * against the source PU -- costDistortionMv (turing/Search.hpp:1965-1998 -> havoc/pred_inter.cpp:113-202 +
 * turing/Measure.h:97-135).  Jobs are havoc_mi355x_pred_uni_job with `dst_off` naming the SOURCE block in d_src;
 * d_cost[i] = the PU SATD (8x8 / 4x4 / 2x2 tiles by PU shape); the prediction is not written.  max_w / max_h are
 * upper bounds on the PU sizes in this batch (the reference's table is indexed by width class,
 * havoc/pred_inter.h:47-50); 64, 64 is always valid, tighter bounds pick a kernel with less LDS per PU. */
int havoc_mi355x_subpel_satd(havoc_mi355x_ctx *ctx, int S, int taps, int bitDepth, int max_w, int max_h, const void *d_src,
                             intptr_t stride_src, const void *d_ref, intptr_t stride_ref, const havoc_mi355x_pred_uni_job *d_jobs,
                             int njobs, int32_t *d_cost);

/* ------------------------------------------------------------------------------------------------------- */
/* intra prediction                                                                                          */
/* ------------------------------------------------------------------------------------------------------- */

/* havoc::intra::Function<Sample> via Table::lookup (havoc/pred_intra.h:32-52, pred_intra.cpp:20282-20401) */
int havoc_mi355x_intra(havoc_mi355x_ctx *ctx, int S, int bitDepth, int log2TrafoSize, void *d_dst, intptr_t stride_dst,
                       const void *d_neighbours, const havoc_mi355x_intra_job *d_jobs, int njobs);

/* The 35-mode luma SATD stage of one intra partition, fused: for every mode m = 0..34 the prediction
 * (havoc/pred_intra.cpp:20282-20401, from the filtered neighbour array when bit m of filt_lo/filt_hi is set, else the
 * unfiltered one -- the caller's filterFlag, turing/Reconstruct.cpp:659) is measured against the source block with
 * the Hadamard SATD (8x8 tiles; one 4x4 for 4x4 blocks) exactly as PredictIntraLumaBlock does
 * (turing/Reconstruct.cpp:630-701) inside the mode loop of searchIntraPartition (turing/Search.hpp:113-142).
 * d_cost[35*i + m] = SATD; no prediction is written to memory. */
typedef struct {
    int32_t src_off;   /* top-left of the source block in d_src */
    int32_t nb_off;    /* unfiltered neighbours[0] (same layout as havoc_mi355x_intra_job) in d_neighbours */
    int32_t nbf_off;   /* filtered neighbours[0] */
    uint32_t filt_lo;  /* bit m: mode m (0..31) predicts from the filtered array */
    uint32_t filt_hi;  /* bits 0..2: modes 32..34 */
    int32_t edge;      /* cIdx == 0 (edge filters apply when log2TrafoSize < 5) */
    int32_t reserved[2];
} havoc_mi355x_intra_search_job; /* 32 bytes */
int havoc_mi355x_intra_satd35(havoc_mi355x_ctx *ctx, int S, int bitDepth, int log2TrafoSize, const void *d_src, intptr_t stride_src,
                              const void *d_neighbours, const havoc_mi355x_intra_search_job *d_jobs, int njobs, int32_t *d_cost);

/* ------------------------------------------------------------------------------------------------------- */
/* residual, transforms, quantisation                                                                        */
/* ------------------------------------------------------------------------------------------------------- */

/* res = src - pred, int16 (turing/Reconstruct.cpp:258-260, 1274-1286).  job: a = src, b = pred;
 * the residual of job i is written at d_res + res_off[i] with row stride stride_res. */
int havoc_mi355x_residual(havoc_mi355x_ctx *ctx, int S, int16_t *d_res, intptr_t stride_res, const int32_t *d_res_off,
                          const void *d_src, intptr_t stride_src, const void *d_pred, intptr_t stride_pred,
                          const havoc_mi355x_pair_job *d_jobs, int njobs);
/* havoc::Transform via get_transform<bitDepth> (havoc/transform.h:117-140, transform.cpp:3087-3397) */
int havoc_mi355x_transform(havoc_mi355x_ctx *ctx, int bitDepth, int trType, int log2TrafoSize, int16_t *d_coeffs,
                           const int16_t *d_res, intptr_t stride_res, const havoc_mi355x_tu_job *d_jobs, int njobs);
/* havoc::inverse_transform (havoc/transform.h:33-57, transform.cpp:339-355): d_res n*n contiguous at res_off */
int havoc_mi355x_inverse_transform(havoc_mi355x_ctx *ctx, int bitDepth, int trType, int log2TrafoSize, int16_t *d_res,
                                   const int16_t *d_coeffs, const havoc_mi355x_tu_job *d_jobs, int njobs);
/* havoc::inverse_transform_add<Sample> (havoc/transform.h:61-84, transform.cpp:358-401); d_pred may be d_dst */
int havoc_mi355x_inverse_transform_add(havoc_mi355x_ctx *ctx, int S, int bitDepth, int trType, int log2TrafoSize,
                                       void *d_dst, intptr_t stride_dst, const void *d_pred, intptr_t stride_pred,
                                       const int16_t *d_coeffs, const havoc_mi355x_tu_job *d_jobs, int njobs);
/* The TU chain of the reference's reconstruction, fused on either side of the host's quantiser decision (at
 * speed=medium RDOQ, which is not havoc, runs between the two; turing/Reconstruct.cpp:258-353 and :766-856):
 *   tu_forward     : residual = src - pred (Reconstruct.cpp:258-260), then havoc::Transform -> d_coeffs[coef_off ..]
 *   tu_reconstruct : havoc_quantize_inverse(levels, scale, shift), then havoc::inverse_transform_add onto the
 *                    prediction -> d_rec[rec_off ..] (row stride stride_rec), then havoc_ssd(src, rec) -> d_ssd[i].
 * Residual and de-quantised coefficients are never written.  Bit-identical to calling the separate entry points. */
typedef struct {
    int32_t coef_off;  /* n*n contiguous int16: coefficients (forward) / levels (reconstruct) */
    int32_t src_off;   /* source block in d_src */
    int32_t pred_off;  /* prediction block in d_pred */
    int32_t rec_off;   /* reconstructed block in d_rec (may be the prediction's own location) */
} havoc_mi355x_tu_fused_job; /* 16 bytes */
int havoc_mi355x_tu_forward(havoc_mi355x_ctx *ctx, int S, int bitDepth, int trType, int log2TrafoSize, int16_t *d_coeffs, const void *d_src,
                            intptr_t stride_src, const void *d_pred, intptr_t stride_pred, const havoc_mi355x_tu_fused_job *d_jobs, int njobs);
/* The 35-mode stage of ONE intra partition in one launch (round 6; what libhavoc_classic.so's serve layer precomputes when a partition's first prediction / SATD /
 * transform call arrives -- turing/Search.hpp:113-142, Reconstruct.cpp:244-353): job i = mode slot i, its prediction block (pred_off, row stride stride_pred) against
 * the partition's source block (src_off); coef_off / rec_off = where its n*n coefficients / reconstructed samples (n x n, contiguous) go.  Per job, each bit-identical to
 * the entry point named:
 *   d_satd[i * tiles + t]    havoc_mi355x_satd of tile t (8x8 tiles in raster order; one 4x4 tile for a 4x4 partition), only if with_satd
 *   d_coeffs                 havoc_mi355x_tu_forward, DST-VII for 4x4 / DCT above (the luma rule, Reconstruct.cpp:263)
 *   d_coeffs_dct             4x4 only: havoc_mi355x_tu_forward with trType 0 (a 4x4 chroma block)
 *   d_rec0, d_ssd0[i]        havoc_mi355x_tu_reconstruct on a block of ZERO levels: the reconstruction (= the clipped prediction) and its SSD against the source */
int havoc_mi355x_intra_measure(havoc_mi355x_ctx *ctx, int S, int bitDepth, int log2TrafoSize, int16_t *d_coeffs, int16_t *d_coeffs_dct, int32_t *d_satd, void *d_rec0,
                               uint32_t *d_ssd0, const void *d_src, intptr_t stride_src, const void *d_pred, intptr_t stride_pred, const havoc_mi355x_tu_fused_job *d_jobs,
                               int njobs, int with_satd);
int havoc_mi355x_tu_reconstruct(havoc_mi355x_ctx *ctx, int S, int bitDepth, int trType, int log2TrafoSize, int scale, int shift, void *d_rec,
                                intptr_t stride_rec, const void *d_pred, intptr_t stride_pred, const void *d_src, intptr_t stride_src,
                                const int16_t *d_levels, const havoc_mi355x_tu_fused_job *d_jobs, int njobs, uint32_t *d_ssd);

/* What a rate estimate reads of a block of quantised levels, without the levels crossing the link: d_out[2 * i] = number of non-zero
 * levels, d_out[2 * i + 1] = sum of |level| of the block d_jobs[2 * i] (int16 offset, multiple of 2) of d_jobs[2 * i + 1] values (multiple
 * of 2).  Not a reference primitive: the reference's EstimateRate<residual_coding> (turing/EstimateRate.h) walks the levels on the host; a
 * batch client that chooses between transform-tree candidates (libhavoc_search.so: havoc_search_rqt) reads these instead. */
int havoc_mi355x_level_stats(havoc_mi355x_ctx *ctx, const int16_t *d_levels, const int32_t *d_jobs, int njobs, int32_t *d_out);

/* havoc_quantize (havoc/quantize.h:63, quantize.cpp:278-304): d_cbf[i] = OR of the job's outputs */
int havoc_mi355x_quantize(havoc_mi355x_ctx *ctx, int16_t *d_dst, const int16_t *d_src,
                          const havoc_mi355x_quant_job *d_jobs, int njobs, int32_t *d_cbf);
/* havoc_quantize_inverse (havoc/quantize.h:42, quantize.cpp:37-46) */
int havoc_mi355x_quantize_inverse(havoc_mi355x_ctx *ctx, int16_t *d_dst, const int16_t *d_src,
                                  const havoc_mi355x_quant_job *d_jobs, int njobs);
/* havoc_quantize_reconstruct (havoc/quantize.h:84, quantize.cpp:538-549), 8-bit only:
 * rec = clip8(pred + res); job: dst_off = rec, pred_off, res_off (n*n contiguous) */
int havoc_mi355x_quantize_reconstruct(havoc_mi355x_ctx *ctx, int log2TrafoSize, uint8_t *d_rec, intptr_t stride_rec,
                                      const uint8_t *d_pred, intptr_t stride_pred, const int16_t *d_res,
                                      const havoc_mi355x_tu_job *d_jobs, int njobs);

/* ------------------------------------------------------------------------------------------------------- */
/* sample-adaptive offset primitives                                                      */
/* ------------------------------------------------------------------------------------------------------- */
/* The statistics the encoder's SAO decision reads (turing/EncSao.h:111-283: edge_offset_stats_class0..3, band_offset_luma_stats)
 * over one block (a CTU of one colour component; the joint band statistics of Cb and Cr, EncSao.h:62-109: havoc_mi355x_sao_band_chroma).
 * d_out: 105 int64 per job -- for edge class c = 0..3 the sums of (original - reconstruction) per category at [10c .. 10c+4] and the
 * sample counts at [10c+5 .. 10c+9]; the sums per band at [40..71], the counts per band at [72..103]; [104] = the band position the
 * reference's function returns.  The block's outermost ring of samples is not counted, as in the reference (w, h >= 3). */
typedef struct {
    int32_t src_off, rec_off; /* the block in the original / reconstructed plane */
    int32_t w, h;
} havoc_mi355x_sao_stats_job; /* 16 bytes */
int havoc_mi355x_sao_stats(havoc_mi355x_ctx *ctx, int S, int bitDepth, const void *d_src, intptr_t stride_src, const void *d_rec, intptr_t stride_rec,
                           const havoc_mi355x_sao_stats_job *d_jobs, int njobs, int64_t *d_out);
/* band_offset_chroma_stats (turing/EncSao.h:62-109): the band statistics of the Cb and the Cr block of a CTU TOGETHER -- one histogram over
 * both interiors -- as the encoder's chroma SAO decision reads them.  d_src / d_rec: the planes holding both chroma components (offsets in
 * samples; one stride each).  d_out[65 * i]: E[32], count[32], the band position the reference's function returns. */
typedef struct {
    int32_t src_u, src_v;   /* the CTU's Cb / Cr block in the source chroma */
    int32_t rec_u, rec_v;   /* ... in the deblocked reconstruction */
    int32_t w, h;
    int32_t reserved[2];
} havoc_mi355x_sao_chroma_job; /* 32 bytes */
int havoc_mi355x_sao_band_chroma(havoc_mi355x_ctx *ctx, int S, int bitDepth, const void *d_src, intptr_t stride_src, const void *d_rec, intptr_t stride_rec,
                                 const havoc_mi355x_sao_chroma_job *d_jobs, int njobs, int64_t *d_out);
/* sao_filter_band / sao_filter_edge (turing/sao.h, sao.cpp:33-92) on one block: d_dst and d_src are different pictures; the edge
 * filter reads one sample beyond the block on every side.  type 0 copies (SAO off for the block), 1 = band offset with
 * offsets[] = the 32-entry table LoopFilter.h:994-1004 builds, 2 = edge offset with offsets[0..4] = SaoOffsetVal. */
typedef struct {
    int32_t dst_off, src_off, w, h;
    int32_t type, eo_class;
    int16_t offsets[32];
    int32_t reserved[2];
} havoc_mi355x_sao_job; /* 96 bytes */
int havoc_mi355x_sao_filter(havoc_mi355x_ctx *ctx, int S, int bitDepth, void *d_dst, intptr_t stride_dst, const void *d_src, intptr_t stride_src,
                            const havoc_mi355x_sao_job *d_jobs, int njobs);

/* ------------------------------------------------------------------------------------------------------- */
/* rate-distortion optimised quantisation                                                 */
/* ------------------------------------------------------------------------------------------------------- */
/* Rdoq::runQuantisation (turing/Rdoq.cpp:37-454), the quantiser the reference's speed=medium uses between the forward
 * transform and the reconstruction (turing/Reconstruct.cpp:289-312 intra, :794-812 inter).  One job = one transform block of
 * the launch's size.  The CABAC probability states are read, never updated, so the entropy coder is a frozen snapshot of
 * state bytes (ContextModel::state, turing/ContextModel.h:29-31) in this layout, 128 bytes per snapshot: */
enum {
    HAVOC_RDOQ_CTX_ROOT_CBF = 0,   /* rqt_root_cbf [1] */
    HAVOC_RDOQ_CTX_CBF_LUMA = 1,   /* cbf_luma [2] */
    HAVOC_RDOQ_CTX_CBF_CHROMA = 3, /* cbf_cb / cbf_cr [4] */
    HAVOC_RDOQ_CTX_LAST_X = 8,     /* last_sig_coeff_x_prefix [18] */
    HAVOC_RDOQ_CTX_LAST_Y = 26,    /* last_sig_coeff_y_prefix [18] */
    HAVOC_RDOQ_CTX_CSBF = 44,      /* coded_sub_block_flag [4] */
    HAVOC_RDOQ_CTX_SIG = 48,       /* sig_coeff_flag [44] */
    HAVOC_RDOQ_CTX_GREATER1 = 92,  /* coeff_abs_level_greater1_flag [24] */
    HAVOC_RDOQ_CTX_GREATER2 = 116, /* coeff_abs_level_greater2_flag [6] */
    HAVOC_RDOQ_CTX_BYTES = 128
};
typedef struct {
    int32_t dst_off, src_off;          /* n*n contiguous int16 levels (out) / coefficients (in); multiples of 4 (8-byte rows) */
    int32_t quant_scale, quant_shift;  /* runQuantisation's quantiserScale / quantiserShift */
    int32_t inv_scale;                 /* the constructor's invQuantScale */
    int32_t lambda_q16, sdh_factor;    /* from havoc_mi355x_rdoq_lambda */
    int32_t ctx_index;                 /* which 128-byte snapshot of d_states */
    uint8_t c_idx, scan_idx;           /* rc.cIdx (0..2), scanIdx (0 diagonal, 1 horizontal, 2 vertical) */
    uint8_t is_intra, sdh;             /* isIntra, isSdhEnabled */
    int32_t reserved[3];
} havoc_mi355x_rdoq_job; /* 48 bytes */
/* the two integers the Rdoq constructor derives from the floating-point lambda (turing/Rdoq.h:163-167); host-side, no device work */
void havoc_mi355x_rdoq_lambda(double lambda, int inv_scale, int32_t *lambda_q16, int32_t *sdh_factor);
/* d_cbf[i] = runQuantisation's return value (OR of the kept absolute levels).  d_dst and d_src are different buffers, as in the
 * reference (quantizedCoefficients vs coefficients, Reconstruct.cpp:276-277); every level of a job's block is written. */
/* d_work: havoc_mi355x_rdoq_workspace(njobs) bytes of device scratch, 16-byte aligned, private to the launch until it completes
 * (per block: which 4x4 groups are non-zero, the block's energy; the order the blocks are walked in: densest first, so that
 * the 64 blocks a wavefront walks finish together). */
size_t havoc_mi355x_rdoq_workspace(int njobs);
int havoc_mi355x_rdoq(havoc_mi355x_ctx *ctx, int bitDepth, int log2TrafoSize, int16_t *d_dst, const int16_t *d_src, const uint8_t *d_states,
                      const havoc_mi355x_rdoq_job *d_jobs, int njobs, int32_t *d_cbf, void *d_work, size_t work_bytes);

/* The same two steps with the scan of the coefficients done where they are produced (16x16 / 32x32 blocks; round 3):
 *   tu_forward_scan  = tu_forward + the first pass of the device RDOQ (which 4x4 groups hold a rounded level, the block's energy -> d_work;
 *                      the level block of d_rdoq_jobs[i].dst_off zeroed).  d_rdoq_jobs[i] describes the same block as d_jobs[i]
 *                      (src_off = the job's coef_off).
 *   rdoq_prescanned  = the rest of havoc_mi355x_rdoq on that workspace: levels and flags identical to havoc_mi355x_rdoq on the coefficients.
 * Saves a kernel that re-reads every coefficient per block size (2 x 47 MB and 47 us of launches per 1080p picture). */
int havoc_mi355x_tu_forward_scan(havoc_mi355x_ctx *ctx, int S, int bitDepth, int log2TrafoSize, int16_t *d_coeffs, const void *d_src, intptr_t stride_src,
                                 const void *d_pred, intptr_t stride_pred, const havoc_mi355x_tu_fused_job *d_jobs, int njobs,
                                 const havoc_mi355x_rdoq_job *d_rdoq_jobs, int16_t *d_levels, void *d_work, size_t work_bytes);
int havoc_mi355x_rdoq_prescanned(havoc_mi355x_ctx *ctx, int bitDepth, int log2TrafoSize, int16_t *d_dst, const int16_t *d_src, const uint8_t *d_states,
                                 const havoc_mi355x_rdoq_job *d_jobs, int njobs, int32_t *d_cbf, void *d_work, size_t work_bytes);

/* ---- the intra decisions that are data-parallel (one lane per partition), taken on the device between the launches they separate ----
 * Not reference primitives: the reference takes them inline in searchIntraPartition (turing/Search.hpp:40-255).  A batch client that
 * refines a picture's intra partitions (libhavoc_search.so: havoc_search_intra_device) chains
 *     intra_satd35 -> intra_order -> intra_expand -> intra -> tu_forward -> rdoq -> tu_reconstruct -> level_stats -> intra_decide -> tu_reconstruct
 * so that neither the 35 costs per partition nor the job records per candidate cross the link. */
#define HAVOC_MI355X_INTRA_MAX_ORDER 12
typedef struct
{
    int32_t cand_mode_list[3];     /* candModeList (Search.hpp:55-98) */
    int32_t neighbour_modes;       /* how many of them are forced into the refinement if not selected (Search.hpp:170-180) */
    int32_t max_refine;            /* selections before that happens; max_refine + neighbour_modes <= HAVOC_MI355X_INTRA_MAX_ORDER */
    int32_t reserved;
    int64_t rate_a_minus_c, rate_b_minus_c;   /* Q16 rate offsets of candModeList[0] / [1], [2] relative to a mode outside the list */
} havoc_mi355x_intra_mpm;          /* 40 bytes; = havoc_search_intra_ctx */
typedef struct
{
    int32_t mode, index, evaluated, reserved;
    int64_t cost;                  /* Q16: mode rate + residual rate + ssd * reciprocal lambda */
    int32_t cbf;
    uint32_t ssd;
    int32_t nonzero, sum_abs;
} havoc_mi355x_intra_choice;       /* 40 bytes; = havoc_intra_rd_result */
/* Search.hpp:55-98, 143-190: costs = rate offsets + lambda_q16 * d_satd35[35 * i + mode] (64-bit), then the order the modes are refined in:
 * d_order[HAVOC_MI355X_INTRA_MAX_ORDER * i + k], k < d_count[i]; d_slot[i] = the partition's first candidate slot;
 * d_total[0] = slots handed out; d_total[1]: bit 0 = some partition wanted more than HAVOC_MI355X_INTRA_MAX_ORDER (its order is cut), bit 1 = some record was out of
 * range (a mode outside 0..34, neighbour_modes outside 0..3, max_refine < 1: brought into range, nothing is written outside the partition's slots) -- results are
 * then not to be used. */
int havoc_mi355x_intra_order(havoc_mi355x_ctx *ctx, const int32_t *d_satd35, const havoc_mi355x_intra_mpm *d_mpm, int n, int32_t lambda_q16, int32_t *d_order,
                             int32_t *d_count, int32_t *d_slot, int32_t *d_total);
/* the job records of every candidate c = d_slot[i] + k: prediction into piece c (n x n, stride n) from the filtered / unfiltered neighbours as
 * the partition's mask says, residual + forward transform, Rdoq (intra, scan by mode: Global.h:1212-1227), level statistics; d_owner[c] = i */
int havoc_mi355x_intra_expand(havoc_mi355x_ctx *ctx, const havoc_mi355x_intra_search_job *d_parts, const int32_t *d_order, const int32_t *d_count, const int32_t *d_slot,
                              const int32_t *d_ctx_index, int n, int log2TrafoSize, int quant_scale, int quant_shift, int inv_scale, int lambda_q16, int sdh_factor, int sdh,
                              havoc_mi355x_intra_job *d_intra_jobs, havoc_mi355x_tu_fused_job *d_tu_jobs, havoc_mi355x_rdoq_job *d_rdoq_jobs, int32_t *d_stat_jobs,
                              int32_t *d_owner);
/* Search.hpp:143-255: the first candidate with the smallest   mode rate + (1 + (cbf ? 2 * nonzero + sum_abs : 0) << 16) + reciprocal_lambda_q16 * ssd.
 * The RATE here is a stand-in, not the reference's: its RD stage charges every mode -- candModeList[0] included -- the bits EstimateRateLuma measures in the CABAC state of
 * that moment (prev_intra_luma_pred_flag, mpm_idx / rem_intra_luma_pred_mode, the residual), where this uses the first stage's offsets (rate_a_minus_c is 0 rate for
 * candModeList[0]) and a count of levels.  The order of evaluation, the Q16 arithmetic and the strict comparison are the reference's (pinned with the encoder's own rates:
 * tests/test_trace_pin.py); a caller with an entropy coder supplies real rates through the host form (search/tu_decision.hpp: decideIntraRd with a rate functor);
 * d_final[i] = the champion's tu job with rec_off = i << (2 * log2TrafoSize) (the block it reconstructs into) */
int havoc_mi355x_intra_decide(havoc_mi355x_ctx *ctx, const havoc_mi355x_intra_mpm *d_mpm, const int32_t *d_order, const int32_t *d_count, const int32_t *d_slot,
                              const int32_t *d_cbf, const uint32_t *d_ssd, const int32_t *d_stats, const havoc_mi355x_tu_fused_job *d_tu_jobs, int n, int log2TrafoSize,
                              int32_t reciprocal_lambda_q16, havoc_mi355x_intra_choice *d_out, havoc_mi355x_tu_fused_job *d_final);

/* ---- the transform-tree decision of inter units and the picture's block structure on the device (round 4; csrc/kernels_decide.hip) ----
 * reconstructInter's choice between one transform block and four (turing/Reconstruct.cpp:1296-1428; turingcodec_amd/search/tu_decision.hpp: decideRqt) from the outcomes of a
 * unit's five candidate blocks, evaluated by the caller's chain tu_forward -> rdoq -> tu_reconstruct (into pieces) -> level_stats: the split tree first; none of its blocks
 * coded -> the unit stays unsplit without residual and depth 0 is never considered; else the cheaper of  rate + ssd * reciprocal_lambda  wins, depth 0 on `<` (rate = the
 * stand-in of tu_decision.hpp: (1 + (cbf ? 2 * nonzero + sum_abs : 0)) << 16 per block).  Candidate j of size s has its outcome at sizes[s - 2].d_cbf / d_ssd / d_stats [j] and its
 * job at d_jobs[j]; unit i's depth-0 candidate is d_zero_at[i] of its size, its four depth-1 candidates d_one_at[i] .. + 3 of the next smaller size.  d_final[j] = the job that
 * reconstructs candidate j again: into the picture (rec_off = rec_origin + y * rec_stride + x) when it belongs to the chosen tree -- a unit without residual through its four
 * depth-1 blocks -- else at dump_off (any block-sized area of the same allocation nobody reads: the launches after the decision have a fixed size). */
typedef struct { int32_t x0, y0, log2_size, ctx_index; } havoc_mi355x_rqt_unit;                 /* = havoc_rqt_cu */
typedef struct { int32_t cbf; uint32_t ssd; int32_t nonzero, sum_abs; } havoc_mi355x_tu_outcome;   /* = havoc_tu_outcome */
typedef struct { int32_t depth, tried_zero; havoc_mi355x_tu_outcome zero, one[4]; int64_t cost_zero, cost_one; } havoc_mi355x_rqt_choice;      /* 104 bytes; = havoc_rqt_result */
typedef struct { const int32_t *d_cbf; const uint32_t *d_ssd; const int32_t *d_stats; const havoc_mi355x_tu_fused_job *d_jobs; havoc_mi355x_tu_fused_job *d_final; } havoc_mi355x_rqt_size;
int havoc_mi355x_rqt_decide(havoc_mi355x_ctx *ctx, const havoc_mi355x_rqt_unit *d_units, int n, const int32_t *d_zero_at, const int32_t *d_one_at, const havoc_mi355x_rqt_size sizes[4],
                            int64_t rec_origin, intptr_t rec_stride, int32_t dump_off, int32_t reciprocal_lambda_q16, havoc_mi355x_rqt_choice *d_out);
/* the 4x4 cells havoc_mi355x_derive_bs reads, made from the decisions: every unit one inter 2Nx2N prediction unit from list 0 (decoded picture dpb_index0) at the vector d_field
 * (int16 [2][height / 4][width / 4][2]) holds at its origin, coded flags and transform sizes as decided; cells outside the units: no motion coded, qp, tu_log2 = 2 */
int havoc_mi355x_block_cells(havoc_mi355x_ctx *ctx, int width, int height, int qp, int dpb_index0, const int16_t *d_field, const havoc_mi355x_rqt_unit *d_units,
                             const havoc_mi355x_rqt_choice *d_decisions, int n, havoc_mi355x_cell *d_cells);
/* ... the cells of n more units into cells that hold other units' already (a picture decided band by band: havoc_mi355x_block_cells with n = 0 blanks them once) */
int havoc_mi355x_block_cells_add(havoc_mi355x_ctx *ctx, int width, int height, int qp, int dpb_index0, const int16_t *d_field, const havoc_mi355x_rqt_unit *d_units,
                                 const havoc_mi355x_rqt_choice *d_decisions, int n, havoc_mi355x_cell *d_cells);

/* ---- an intra picture's partitions with their real dependencies (round 4; csrc/kernels_decide.hip) ----
 * A partition predicts from the reconstruction of what precedes it (turing/Reconstruct.cpp:609-615) and takes candModeList from its neighbours' decided modes
 * (turing/CandModeList.h:33-95).  A client that runs the batch chain over the partitions level by level (every partition of a level has all its neighbours final)
 * gets what a level needs from the picture's running state here:
 *   intra_gather: per partition the 4n + 1 reference samples from the reconstruction picture -- a sample is there when it lies inside the picture and the partition
 *     owning its 4x4 cell (d_owner, one int32 per cell) precedes this one in coding order (`index`); HEVC 8.4.4.2.2 fills in the others -- written unfiltered and
 *     [1 2 1]-filtered (turing/IntraReferenceSamples.h:373-421) at the job's nb_off / nbf_off, and cand_mode_list / neighbour_modes of its havoc_mi355x_intra_mpm record
 *     from d_modes (one byte per cell) left of and above it (above only inside the same CTU row);
 *   intra_commit: the champions' blocks (block i = n x n samples at i * n * n, as havoc_search_intra_device leaves them) into the picture, their modes into d_modes. */
typedef struct { int32_t x0, y0, log2, index; } havoc_mi355x_intra_chain_part;        /* 16 bytes */
typedef struct { int32_t pic_width, pic_height, stride, pad, cells_per_row, bit_depth, ctb_log2, strong_intra_smoothing; } havoc_mi355x_intra_chain_layout;      /* 32 bytes;
 * strong_intra_smoothing = the sequence's strong_intra_smoothing_enabled_flag (turing/Encoder.cpp:688 sets it): flat 32x32 blocks then take the bi-linear filter of
 * IntraReferenceSamples.h:382-402 */
int havoc_mi355x_intra_gather(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_intra_chain_layout *layout, const void *d_rec, const int32_t *d_owner, const uint8_t *d_modes,
                              const havoc_mi355x_intra_chain_part *d_parts, int n, const havoc_mi355x_intra_search_job *d_jobs, void *d_neighbours, havoc_mi355x_intra_mpm *d_mpm);
int havoc_mi355x_intra_commit(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_intra_chain_layout *layout, void *d_rec, uint8_t *d_modes, const havoc_mi355x_intra_chain_part *d_parts,
                              int n, const void *d_blocks, const int32_t *d_mode, int mode_stride);      /* mode of partition i = d_mode[i * mode_stride]: a plain array (1) or
                                                                                                            havoc_mi355x_intra_choice records (10) */
/* after intra_expand: the candidate slots [d_total[0], capacity) that no partition was given become copies of slot 0's records writing into their own slots, so the chain
 * (intra -> tu_forward -> rdoq -> tu_reconstruct -> level_stats) can run over `capacity` = n * HAVOC_MI355X_INTRA_MAX_ORDER jobs WITHOUT the host waiting for the count */
int havoc_mi355x_intra_fill_spare(havoc_mi355x_ctx *ctx, const int32_t *d_total, int capacity, int log2TrafoSize, havoc_mi355x_intra_job *d_intra_jobs,
                                  havoc_mi355x_tu_fused_job *d_tu_jobs, havoc_mi355x_rdoq_job *d_rdoq_jobs, int32_t *d_stat_jobs, int32_t *d_owner);

/* ---- job tables made on the device from the decided motion field (round 4; csrc/kernels_decide.hip) ----
 * Not reference primitives: the reference builds its prediction calls inline from the vectors it has just decided (predictInter, turing/Search.hpp:1659-1706;
 * searchMergeModes :1754-1768).  A batch client whose searches leave the motion field in device memory (havoc_mi355x_search_picture_uni: d_field) gets the
 * job tables of the next launches here, so nothing of the field crosses the link.  Pictures: the planes of one component lie at multiples of *_plane_elems in
 * ONE allocation -- luma: source, list 0, list 1; chroma: Cb source, list 0, list 1, Cr source, list 0, list 1 -- with *_pad samples of border (>= range + 4). */
typedef struct {
    int32_t pic_width, pic_height;
    int32_t range;                 /* vectors are limited so that a block stays within `range` samples of the picture (LimitFullPelMv: CtbSizeY = 64) */
    int32_t field_cw;              /* cells per row of d_field = (pic_width + 3) / 4 */
    int32_t luma_stride, luma_pad, luma_plane_elems;
    int32_t chroma_stride, chroma_pad, chroma_plane_elems;
    int32_t reserved[2];
} havoc_mi355x_field_layout;       /* 48 bytes */
/* the five spatial merge candidates (HEVC 8.5.3.2.3 positions A1, B1, B0, A0, B2; the vectors of both lists decided there, zero outside the picture; Mvp.h's
 * pruning / temporal candidate are the caller's) of n square units of one size as HavocPredBi jobs in three planes: job 5 * i + k, dst_off = job * size^2
 * (chroma: (size / 2)^2), d_vectors[job][list] = (x, y). */
int havoc_mi355x_merge_jobs(havoc_mi355x_ctx *ctx, const havoc_mi355x_field_layout *layout, const int16_t *d_field, const int32_t *d_x0, const int32_t *d_y0, int n, int log2_size,
                            havoc_mi355x_pred_bi_job *d_luma_jobs, havoc_mi355x_pred_bi_job *d_cb_jobs, havoc_mi355x_pred_bi_job *d_cr_jobs, int16_t *d_vectors);
/* the merge decision of n units from the SATDs of their 5 candidates in three planes (job 5 * i + k): d_cost[5 * i + k] = (min(k + 1, 4) << 16) + (satdY + satdCb + satdCr)
 * * reciprocal_sqrt_lambda_q16 (measurePuCost, turing/Search.hpp:1659-1706, with a stand-in for the CABAC rate of the merge index), d_best[i] = the first of the cheapest. */
int havoc_mi355x_merge_decide(havoc_mi355x_ctx *ctx, const int32_t *d_satd_y, const int32_t *d_satd_cb, const int32_t *d_satd_cr, int n, int64_t reciprocal_sqrt_lambda_q16,
                              int64_t *d_cost, int32_t *d_best);
/* HavocPredUni jobs of n square units at the vector decided for `list` at their origin: plane 0 = luma, 1 / 2 = Cb / Cr (half size, eighth-sample phases) */
int havoc_mi355x_pred_jobs(havoc_mi355x_ctx *ctx, const havoc_mi355x_field_layout *layout, const int16_t *d_field, int list, const int32_t *d_x0, const int32_t *d_y0, int n,
                           int log2_size, int plane, const int32_t *d_dst_off, havoc_mi355x_pred_uni_job *d_jobs);

/* ---- a picture's uni-directional motion searches with the decision loops ON THE DEVICE (csrc/kernels_search.hip) ----
 * The reference runs searchMotionUni (turing/Search.hpp:1317-1355: integer search :2060-2336, sub-sample refinement :2010-2061, 2340-2358) per
 * prediction unit and list through the havoc_sad / havoc_sad_multiref / HavocPredUni / hadamard_satd tables.  Here the same loops (restated in
 * turingcodec_amd/search/decision.hpp, compiled for gfx950) run inside the kernel with those primitives computed by the workgroup's lanes; the
 * picture's PUs are walked in the encoder's order with the dependencies of turingcodec_amd/search/picture_order.hpp (derived predictors,
 * mvPreviousInteger2Nx2N along the CTU row, CTU (x, y) after (x + 1, y - 1): TaskEncodeSubstream.cpp:71-95), one launch per wavefront step.
 * The records are those of turingcodec_amd/search/search_abi.h: d_pus = havoc_picture_pu[] (CTU by CTU), d_ctu_first[ctus + 1],
 * d_out = havoc_search_result[2 * n] (index 2 * pu + list), d_field = int16 [2][height / 4][width / 4][x, y] (the decided vectors),
 * d_work = havoc_mi355x_search_workspace(width, height) bytes.  step_launches = 0: ONE launch, a workgroup per (CTU row, list) that waits in
 * the kernel for the row above to be two CTUs ahead; a wait that does not end gives up (nothing hangs) and leaves a non-zero int32 in the last
 * 4 bytes of d_work: the results are then invalid.  step_launches = 1: one launch per wavefront step (no waiting inside a kernel).
 * d_out_bi (optional, havoc_search_result[2 * n_pus]): the bi-directional refinement of searchBi (Search.hpp:1796-1827) after a PU's two
 * uni-directional searches, unless nPbW + nPbH == 12: list 0 against the prediction from list 1's vector, then list 1 against the prediction
 * from list 0's refined vector (searchMotionBi, Search.hpp:1498-1657: the ideal second predictor clip(2 * source - other prediction) built in LDS,
 * an 11 x 11 integer grid, two sub-sample steps); mv, mvd, mvp_flag, calls and cost_subpel (= the cost) are filled.  Nothing of the refinement
 * feeds the walk (the motion field keeps the uni-directional vectors), so it is two more launches after it -- a workgroup per PU, list 0 then list 1.  d_phase: the 16 fractional-sample planes of each reference picture
 * (havoc_mi355x_interp_planes; plane 0 = the picture), which must reach ctb_size + 20 samples beyond the picture on every side; origins are the
 * sample offsets of sample (0, 0).  Everything stays on the device: nothing is uploaded or downloaded by this call.
 * The SOURCE plane d_src is read in whole CTUs (the kernel stages a CTU's ctb_size x ctb_size source block with 16-byte loads): when the picture's width / height are
 * not multiples of ctb_size it must be READABLE up to the next multiple -- i.e. own a border of at least ctb_size - 8 samples to the right and below (the picture store's
 * planes have 96; the values there reach no result: only samples inside a prediction unit are measured). */
typedef struct
{
    int32_t pic_width, pic_height, ctb_size, concurrent_frames;
    int32_t met, small_search_window, bi_small_search_window, half_pel, quarter_pel;
    int32_t bit_depth;
    double reciprocal_sqrt_lambda;
} havoc_mi355x_search_params;      /* 48 bytes; = havoc_search_params */
/* n INDEPENDENT searches (searchMotionUni per record) in ONE launch, a workgroup per search: d_pus = havoc_search_pu[n] (search_abi.h: geometry, the two
 * predictors and their rates, the previous 2Nx2N vector -- inputs here, not derived), one reference picture (d_ref / d_phase), d_out =
 * havoc_search_result[n].  The PUs must lie inside the picture with w, h multiples of 4 in 4..64 and the planes must reach ctb_size + 20 samples beyond it. */
int havoc_mi355x_search_motion_uni(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_search_params *params, const void *d_src, int64_t src_origin, intptr_t src_stride,
                                   const void *d_ref, int64_t ref_origin, intptr_t ref_stride, const void *d_phase, intptr_t plane_elems, int64_t phase_origin,
                                   const void *d_pus, int n, void *d_out);
/* n INDEPENDENT bi-directional refinements (searchMotionBi, turing/Search.hpp:1498-1657) in ONE launch, a workgroup per refinement: list
 * d_pus[i].ref_list's vector is refined around d_start[2 * i .. 2 * i + 1] (quarter samples) against the prediction the OTHER list's vector d_pus[i].mv_other
 * gives (read from d_phase_other, the other reference picture's 16 phase planes); d_ref / d_phase = the refined list's picture.  d_out[i]: mv, mvd, mvp_flag,
 * calls, cost_subpel (= the cost).  The list form of the refinement launches of havoc_mi355x_search_picture_uni; mvp_rate[] >= 0. */
int havoc_mi355x_search_motion_bi(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_search_params *params, const void *d_src, int64_t src_origin, intptr_t src_stride,
                                  const void *d_ref, int64_t ref_origin, intptr_t ref_stride, const void *d_phase, intptr_t plane_elems, int64_t phase_origin,
                                  const void *d_phase_other, int64_t phase_other_origin, const void *d_pus, const int16_t *d_start, int n, void *d_out);
size_t havoc_mi355x_search_workspace(int width, int height);
int havoc_mi355x_search_picture_uni(havoc_mi355x_ctx *ctx, int S, const havoc_mi355x_search_params *params, const int64_t mvp_rate[2], const void *d_src,
                                    int64_t src_origin, intptr_t src_stride, const void *d_ref, const int64_t ref_origin[2], intptr_t ref_stride, const void *d_phase,
                                    intptr_t plane_elems, const int64_t phase_origin[2], const void *d_pus, const int32_t *d_ctu_first, int ctus_x, int ctus_y,
                                    int n_pus, void *d_out, void *d_out_bi, int16_t *d_field, void *d_work, int step_launches);
/* Searching into reference pictures that are STILL ARRIVING (the reference: turing/TaskEncodeSubstream.cpp:71-95 -- a CTU starts when its reference picture is
 * reconstructed three CTU rows below it; TaskDeblock.cpp:151-167 publishes the deblocked, padded rows).  d_rows_ready = two device int32: entry l = how many luma rows
 * of reference list l, counted from picture row 0, are final in d_ref AND in all 16 planes of d_phase -- raised (never lowered) by whatever delivers the picture, on
 * ANOTHER stream, while the search kernel runs; the last band raises it to at least pic_height + ctb_size + 8 (bottom border included).  The havoc_mi355x_search_picture_uni
 * launches of this context that follow make CTU row r of list l wait until d_rows_ready[l] >= min((r + 2) * ctb_size, pic_height + ctb_size + 8): with
 * concurrent_frames > 1 (required) the search limits a CTU row's vectors to blocks that end above row (r + 2) * ctb_size - 15 (the reference's LimitFullPelMv), so
 * nothing below is read for a result.  Results are those of the ungated call.  A wait that outlasts ~4 s gives up (HAVOC_MI355X_EDEVICE from the next sync of the
 * caller).  d_rows_ready = NULL removes the gate.  The delivering stream must make its writes visible before it raises the counter (a kernel boundary does), and it
 * must be able to RUN while the search kernel waits: HIP multiplexes the streams of one priority onto GPU_MAX_HW_QUEUES hardware queues (default 4), and a waiting kernel
 * blocks what is queued behind it -- give the delivering stream another priority (hipStreamCreateWithPriority: fine for a few such streams; the high-priority pool is
 * small) or, for many, raise GPU_MAX_HW_QUEUES to about the device's two dozen and keep one priority (bench.py --vr-bands; profiles/r05/banded_pipeline.txt). */
int havoc_mi355x_search_gate(havoc_mi355x_ctx *ctx, const int32_t *d_rows_ready);
/* The PRODUCER's side of the same rule (turing/TaskDeblock.cpp:151-167 deblocks and publishes a picture's rows while the rows below are still being encoded): a launch on
 * THIS context's stream that ends when CTU rows 0 .. ctu_row of both lists of the havoc_mi355x_search_picture_uni running over d_work (its workspace; one-launch form) on
 * ANOTHER stream are done -- whatever is queued behind it (the band's predictions, transform trees, deblocking, padding, the counter a dependent picture's search_gate
 * polls) runs while the rows below are searched.  The two streams must not share a hardware queue (give them different priorities).  The workspace must have been zeroed
 * since the previous picture before anything waits on it.  A wait that outlasts ~8 s, or a search that gave up, sets *d_gave_up and lets the stream through. */
int havoc_mi355x_search_wait_rows(havoc_mi355x_ctx *ctx, const void *d_work, int pic_width, int pic_height, int ctu_row, int32_t *d_gave_up);

#ifdef __cplusplus
}
#endif
#endif
