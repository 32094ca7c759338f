/*
 * havoc_classic_ext.h -- what libhavoc_classic.so offers BESIDE the reference's table API (the headers under include/havoc): the one hook
 * a host encoder adds to make the per-block table calls fast (SURVEY.md 7.1-A "precompute and serve").
 *
 * The reference's table entries take raw host pointers and return per block (havoc/sad.h:58, pred_inter.h:35,
 * hadamard.h:32).  Once the encoder has told the library which host planes are PICTURES -- the input picture when it
 * arrives (turing/TaskEncodeInput.cpp:134-250), a reconstructed picture when it is complete and padded
 * (turing/TaskDeblock.cpp:151-167) -- a pointer identifies (picture, x, y), and the library answers
 *   havoc_sad / havoc_sad_multiref   from a full-pel SAD surface of the (PU, reference) pair: ONE launch on the first
 *                                    call of a search, look-ups afterwards (turing/Search.hpp:1447-1482, 2060-2336);
 *   HavocPredUni (8-tap)             as a strided copy out of the reference's 16 fractional-sample planes, interpolated
 *                                    once at registration and mirrored in pinned host memory (Search.hpp:1976-1979);
 *   havoc_hadamard_satd              from the tile SATDs of all 49 quarter-sample positions around the vector being
 *                                    refined: ONE launch per (PU, list) (Search.hpp:1963-2061, Measure.h:97-135).
 * Round 5, keyed on CONTENT (no registration needed beyond the input picture), per calling thread:
 *   havoc::intra::Function           every mode of a partition from the reference-sample array a call names: ONE launch (35 modes + the edge-filtered forms of
 *                                    DC / 10 / 26), look-ups while the array's content stays the same (turing/Reconstruct.cpp:244-246, 672-674);
 *   havoc_hadamard_satd (intra)      every mode's tiles against the source block, and every mode's forward transform, in ONE more launch (Reconstruct.cpp:684-701);
 *   havoc::Transform                 an intra candidate's: from that launch, when the residual handed in IS source - prediction;
 *   inverse_transform_add, havoc_ssd an intra candidate's: computed in the wait of its havoc_quantize_inverse call (prediction and source are on the device);
 *                                    an inter block's two havoc_ssd calls: in the wait of its inverse_transform_add (the source block is residual + prediction);
 *   every such answer is given only if the operands the call names hold exactly the samples the precomputed value was made from (compared on the host).
 * Round 6 (per calling thread, same rule -- the operands a call names must hold what the value was made from):
 *   havoc_quantize_inverse           from a table of the call's (scale, shift) pair: havoc_mi355x_quantize_inverse of all 65 536 int16 levels, ONE launch the first time a
 *                                    pair is seen (the de-quantiser is element-wise, havoc/quantize.cpp:37-46); an intra candidate WITH levels keeps its launch (its
 *                                    inverse transform + add + SSD ride in the same wait), one WITHOUT any is reconstructed already: the 35-mode stage also makes
 *                                    every mode's reconstruction from a block of zero levels and its SSD (two thirds of the reference encoder's de-quantiser calls at QP 32);
 *   the 35-mode stage itself         in the wait of the partition's FIRST intra call, against the block this thread's partitions of that size have been walking towards
 *                                    (coding order; the source picture is on the device, nothing of the caller's is read) -- used if the first SATD call names that block;
 *   the Cb / Cr candidates of a unit (predict -> residual -> transform, no SATD call: Reconstruct.cpp:244-353) measured at their first `transform` call against
 *                                    residual + prediction -- the source block, exactly -- every mode's transform and zero-level reconstruction in that one wait;
 *   havoc::Transform of an inter unit all blocks of the unit (luma, Cb, Cr: the encoder has subtracted the whole unit before it walks the transform tree,
 *                                    Reconstruct.cpp:1246-1285) in the wait of the first: the library learns which residual blocks followed a first block the last time,
 *                                    reads them when it comes again, and answers the following calls if their residuals hold exactly what was transformed.  (The one
 *                                    place it reads memory the current call does not name: blocks that WERE operands of this thread's earlier `transform` calls, in the
 *                                    encoder's per-thread residual buffer; a host that frees that buffer while encoding must not use this library.)
 *   a one-job call                   packs its operands into pinned memory the device addresses directly and waits ONCE (before: copy + wait, launch, copy + wait).
 * It helps to register all three planes of an input picture as HAVOC_PICTURE_SOURCE (chroma tile SATDs are then batched like luma's).
 * Every served value is what the batch kernel computed on the GPU, bit-identical to the per-call value.  A call the
 * precomputed data cannot answer (unregistered planes, other primitives) takes the one-job launch path -- never a CPU
 * path.  Registered planes must not change until they are unregistered (the encoder's input pictures and completed
 * reference pictures do not).
 */
#ifndef HAVOC_CLASSIC_EXT_H
#define HAVOC_CLASSIC_EXT_H

#include <stddef.h>
#include <stdint.h>

#include "havoc/havoc.h"

#ifdef __cplusplus
extern "C" {
#endif

#define HAVOC_PICTURE_SOURCE 0     /* an input picture: the `src` operand of SAD / SATD calls */
#define HAVOC_PICTURE_REFERENCE 1  /* a reconstructed, padded picture: the `ref` operand; its phase planes are made now */

/* origin = host pointer of luma sample (0, 0); stride in samples; pad = samples of border around the picture that belong
 * to the plane (0 for the reference's input pictures, 96 for its reconstructed pictures, turing/StatePictures.h:155-156);
 * S = bytes per sample.  Uploads the plane (and for a reference interpolates and mirrors its phase planes): call it from
 * the thread that completed the picture.  Returns 0, or a negative havoc_mi355x error code. */
int havoc_classic_register_picture(havoc_code code, const void *origin, intptr_t stride, int width, int height, int pad, int S, int bit_depth,
                                   int role);
int havoc_classic_unregister_picture(havoc_code code, const void *origin);

/* counters since havoc_new_code: [0] table calls answered from precomputed data, [1] table calls that took the one-job
 * launch path, [2] kernel launches issued for table calls (both kinds), [3] SAD surfaces launched, [4] tile-SATD batches
 * launched, [5] pictures registered, [6] bytes uploaded at registration, [7] bytes mirrored back at registration */
void havoc_classic_stats(havoc_code code, int64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif
