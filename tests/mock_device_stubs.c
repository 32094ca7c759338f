/* TEST INFRASTRUCTURE ONLY (see mock_device.c): C-ABI entry points the host libraries link but the host-logic tests never
 * reach.  Kept apart from mock_device.c because they are deliberately declared without their real prototypes. */
#include <stdio.h>
#include <stdlib.h>

/* entry points libhavoc_classic.so links but the host-logic tests never reach: loud, not silent */
#define NOT_IN_MOCK(name) int name() { fprintf(stderr, "mock device: " #name " is not implemented (host-logic tests only)\n"); abort(); }
NOT_IN_MOCK(havoc_mi355x_ssd)
NOT_IN_MOCK(havoc_mi355x_ssd_linear)
NOT_IN_MOCK(havoc_mi355x_pred_bi)
NOT_IN_MOCK(havoc_mi355x_transform)
NOT_IN_MOCK(havoc_mi355x_inverse_transform)
NOT_IN_MOCK(havoc_mi355x_inverse_transform_add)
NOT_IN_MOCK(havoc_mi355x_quantize)
NOT_IN_MOCK(havoc_mi355x_quantize_inverse)
NOT_IN_MOCK(havoc_mi355x_quantize_reconstruct)
