// cand_mode_list.hpp -- the three most probable intra modes of a partition from the modes of its left (A) and above (B) neighbours (HEVC 8.4.2 as turing/CandModeList.h:57-95
// applies it; A / B are already DC where the neighbour is not there, not intra, or -- B -- lies in the CTU row above, CandModeList.h:37-55).  Data-only, compiled for the
// device (csrc/kernels_decide.hip: k_intra_gather) and for the host.  PINNED: the traced reference encoder records A, B and the list it made for every searchIntraPartition
// (the HAVOC_TRACE_INTRA_BEGIN trace point); tests/test_trace_pin.py requires the same list and number of neighbour modes from the recorded A and B (0 differ).
#pragma once

#if defined(__HIPCC__)
#define HAVOC_CML_HD __host__ __device__ __attribute__((always_inline))
#else
#define HAVOC_CML_HD
#endif

namespace havoc_search {

// cand[0..2] = candModeList, returns CandModeList::neighbourModes (1: A == B, 2: they differ)
HAVOC_CML_HD inline int candModeListOf(int a, int b, int cand[3])
{
    if (a == b)
    {
        if (a < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }      // planar, DC, vertical
        else { cand[0] = a; cand[1] = ((a + 29) % 32) + 2; cand[2] = ((a - 1) % 32) + 2; }      // the angle and its two neighbours
        return 1;
    }
    cand[0] = a;
    cand[1] = b;
    cand[2] = (a != 0 && b != 0) ? 0 : ((a != 1 && b != 1) ? 1 : 26);
    return 2;
}

} // namespace havoc_search
