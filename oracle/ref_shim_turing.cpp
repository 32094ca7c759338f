// TEST INFRASTRUCTURE ONLY (see ref_shim.cpp).  The few self-contained template functions of the reference's turing/
// directory that border the havoc path, compiled from the reference headers where they lie (oracle/Makefile, target
// `ref`): Padding::padBlock / padImage (/root/reference/turing/Padding.h:33-97).  turing/Picture.h is included first
// because Padding.h's padPicture template names Picture<Sample>; nothing of either header is copied.
#include "turing/Picture.h"
#include "turing/Padding.h"

#include <cstdint>

extern "C" {

void ref_pad_block_u8(uint8_t *p, int w, int h, intptr_t stride, int pad, int top, int bottom, int left, int right)
{
    Padding::padBlock<uint8_t>(p, w, h, stride, pad, top != 0, bottom != 0, left != 0, right != 0);
}
void ref_pad_block_u16(uint16_t *p, int w, int h, intptr_t stride, int pad, int top, int bottom, int left, int right)
{
    Padding::padBlock<uint16_t>(p, w, h, stride, pad, top != 0, bottom != 0, left != 0, right != 0);
}
void ref_pad_image_u8(uint8_t *p, int w, int h, int stride, int pad) { Padding::padImage<uint8_t>(p, w, h, stride, pad); }
void ref_pad_image_u16(uint16_t *p, int w, int h, int stride, int pad) { Padding::padImage<uint16_t>(p, w, h, stride, pad); }

} // extern "C"
