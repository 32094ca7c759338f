"""Frame-parallel sharding of the hot path over the GPUs of one node.

The primitive layer itself has no cross-block dependency; what ties pictures together is that inter prediction reads
RECONSTRUCTED reference pictures.  The unit of sharding is therefore the picture: one process per GPU, pictures dealt
to ranks in coding order, and ONE exchange step per reference picture -- the owner broadcasts its padded
reconstruction (Y, Cb, Cr planes) to every rank's mirror of the decoded-picture buffer with an RCCL broadcast over
xGMI (backend "nccl" on ROCm; the CPU tests run the same code over gloo).  Non-reference pictures (the RASL_N /
TRAIL_N leaves of the hierarchy) are never sent.  No reduction collective exists on this path.

The structure-of-pictures below restates the reference's 8-picture hierarchical-B docket
(turing/InputQueue.cpp:370-379: coding order 8 4 2 1 3 6 5 7; reference flags nutR/nutN; reference deltas).
"""
from dataclasses import dataclass
from typing import List, Tuple

# (poc offset inside the SOP, is_reference, reference deltas)   -- turing/InputQueue.cpp:325-379
SOP8: List[Tuple[int, bool, Tuple[int, ...]]] = [
    (8, True, (-8,)),
    (4, True, (-4, 4)),
    (2, True, (-2, 2, 6)),
    (1, False, (-1, 1, 3, 7)),
    (3, False, (-1, 1, -3, 5)),
    (6, True, (-2, 2, -6)),
    (5, False, (-1, 1, 3, -5)),
    (7, False, (-1, 1, -7)),
]


@dataclass(frozen=True)
class Picture:
    index: int          # position in coding order (0 = the IDR)
    poc: int
    is_reference: bool
    refs: Tuple[int, ...]   # POCs of the pictures it predicts from


def coding_order(n_sops: int) -> List[Picture]:
    """IDR (POC 0) followed by n_sops hierarchical-B SOPs of 8 pictures, in coding order"""
    pics = [Picture(0, 0, True, ())]
    last = 8 * n_sops
    for s in range(n_sops):
        base = 8 * s
        for off, is_ref, deltas in SOP8:
            poc = base + off
            refs = tuple(sorted({poc + d for d in deltas if 0 <= poc + d <= last}))
            pics.append(Picture(len(pics), poc, is_ref, refs))
    return pics


def owner(pic_index: int, world: int) -> int:
    """pictures are dealt round-robin in coding order: picture i is encoded by rank i % world"""
    return pic_index % world


def dependency_ready_step(pics: List[Picture], world: int) -> List[int]:
    """earliest lock-step `step` at which each picture may start: one after the step of its latest reference.
    With world ranks in lock step, step s encodes pictures [s*world, (s+1)*world); a picture whose reference sits in
    the same step waits for the reference's rows as the reference encoder does (turing/TaskEncodeSubstream.cpp:71-95),
    which the lock-step model approximates by the next step.  Used by tests to check the plan is a valid schedule."""
    by_poc = {p.poc: p for p in pics}
    ready = []
    for p in pics:
        r = 0
        for q in p.refs:
            r = max(r, by_poc[q].index // world + 1)
        ready.append(r)
    return ready


class ReferenceExchange:
    """Mirror of the decoded-picture buffer on every rank + the broadcast step.

    `recon_luma` / `recon_chroma` are this rank's reconstruction planes (1-D tensors in the padded picture layout).
    Every DPB slot is ONE flat buffer (luma then chroma), so a reference picture costs one broadcast, not one per
    plane.  After every rank finished step `step`:

      stage(step)  the owner of a reference picture copies its planes into slot (poc/2 % slots) of its own mirror --
                   from then on the reconstruction planes may be overwritten by the next picture;
      send(step)   one broadcast per reference picture of the step, owner -> every rank's mirror.

    `exchange(step)` = stage + send.  On the GPU both are enqueued on the caller's current stream (the broadcasts on
    RCCL's own stream behind it), so a caller that orders its compute stream only after `stage` overlaps the
    broadcasts of picture i with the computation of picture i+1 (bench.py)."""

    def __init__(self, dist, rank: int, world: int, recon_luma, recon_chroma, slots: int = 6, n_sops: int = 64,
                 single_rank_broadcast: bool = False, recon_chroma2=None):
        import torch
        self._torch = torch
        self.dist, self.rank, self.world = dist, rank, world
        self.single_rank_broadcast = single_rank_broadcast   # exercise the collective even when world == 1
        self.recon_luma, self.recon_chroma = recon_luma, recon_chroma
        self.pics = coding_order(n_sops)
        self.slots = slots
        self.recon_chroma2 = recon_chroma2                   # second chroma plane (Cr), when the caller keeps one
        nl, nc = recon_luma.numel(), recon_chroma.numel()
        nc2 = recon_chroma2.numel() if recon_chroma2 is not None else 0
        self.dpb = [recon_luma.new_zeros(nl + nc + nc2) for _ in range(slots)]
        self.dpb_luma = [b[:nl] for b in self.dpb]          # views
        self.dpb_chroma = [b[nl:nl + nc] for b in self.dpb]
        self.dpb_chroma2 = [b[nl + nc:] for b in self.dpb]
        self.sent_bytes = 0

    def picture_of(self, step: int, rank: int) -> Picture:
        return self.pics[(step * self.world + rank) % len(self.pics)]

    @staticmethod
    def slot_of(pic: Picture, slots: int) -> int:
        return (pic.poc // 2) % slots   # reference pictures have even POC inside a SOP (8 4 2 6) or are the IDR

    def stage(self, step: int, planes=None):
        """`planes`: (luma, chroma[, chroma2]) of the picture being staged when the caller keeps several pictures in
        flight; default: the tensors given at construction"""
        pic = self.picture_of(step, self.rank)
        if pic.is_reference:
            slot = self.slot_of(pic, self.slots)
            if planes is None:
                planes = (self.recon_luma, self.recon_chroma) + ((self.recon_chroma2,) if self.recon_chroma2 is not None else ())
            dst, src = [self.dpb_luma[slot], self.dpb_chroma[slot]], list(planes)
            if len(src) > 2:
                dst.append(self.dpb_chroma2[slot])
            self._torch._foreach_copy_(dst, src)   # one launch for the planes (this sits between two pictures' kernels)

    def send(self, step: int):
        for src in range(self.world):
            pic = self.picture_of(step, src)
            if not pic.is_reference:
                continue
            buf = self.dpb[self.slot_of(pic, self.slots)]
            if src == self.rank:
                self.sent_bytes += buf.numel() * buf.element_size() * (self.world - 1)
            if self.world > 1 or self.single_rank_broadcast:
                self.dist.broadcast(buf, src=src)

    def exchange(self, step: int):
        self.stage(step)
        self.send(step)
