"""Frame-parallel sharding of the hot path over the GPUs of one node.

The primitive layer itself has no cross-block dependency; what ties pictures together is that inter prediction reads
RECONSTRUCTED reference pictures.  The unit of sharding is therefore the picture: one process per GPU, and ONE exchange
step per reference picture -- the owner broadcasts its padded reconstruction (Y, Cb, Cr planes) to every rank's mirror
of the decoded-picture buffer with an RCCL broadcast over xGMI (backend "nccl" on ROCm; the CPU tests run the same code
over gloo).  Non-reference pictures (the RASL_N / TRAIL_N leaves of the hierarchy) are never sent.  No reduction
collective exists on this path.

Who may start when follows the reference: the structure-of-pictures below restates its 8-picture hierarchical-B docket
(turing/InputQueue.cpp:370-379: coding order 8 4 2 1 3 6 5 7; reference flags nutR/nutN; reference deltas), and a picture
starts only when every picture it predicts from is complete and has arrived in the local DPB mirror -- the picture-level
form of the reference's wait-for-reference rule (turing/TaskEncodeSubstream.cpp:71-95 blocks a CTU until the reference
picture is reconstructed 4 CTUs to the right / 3 rows below; across GPUs the granule is the whole picture).
`DagSchedule` is the resulting list schedule: per time slot at most one picture per rank, earliest coding-order
picture first among the ready ones.  With 8 ranks it settles into the level-skewed pipeline -- per slot the anchor of SOP
k+3, POC 4 of SOP k+2, POCs 2 and 6 of SOP k+1 and POCs 1, 3, 5, 7 of SOP k -- i.e. eight pictures in flight, which is
what ">= 6x at 8 GPUs" needs.  The anchor chain (POC 8k -> 8k + 8) is kept on rank 0, where it is a local
dependency, and with 8+ ranks every other reference is scheduled two slots ahead of its users (`lag`), so that no broadcast
sits on a slot's critical path.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

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

    @property
    def l0(self) -> Optional[int]:
        """POC searched as list 0: the nearest earlier picture of the reference set (refIdx 0, turing/Search.hpp:1883)"""
        past = [q for q in self.refs if q < self.poc]
        return max(past) if past else (min(self.refs) if self.refs else None)

    @property
    def l1(self) -> Optional[int]:
        """POC searched as list 1: the nearest later picture; a picture without one (the SOP anchors) uses L0 again"""
        fut = [q for q in self.refs if q > self.poc]
        return min(fut) if fut else self.l0


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


class DagSchedule:
    """List schedule of a hierarchical-B sequence on `world` ranks in lock-step time slots.

    slot(t) -> [Picture or None] * world: the picture each rank works on in slot t.  A picture is READY in slot t when
    every picture it references was scheduled in a slot < t (its reconstruction is broadcast at the end of its slot) and,
    for a reference picture, a DPB mirror slot is free.  Among ready pictures the earliest in coding order goes first;
    run-ahead is bounded by `window` pictures beyond the oldest unscheduled one (DPB capacity).  Deterministic: every
    rank computes the same plan.  dpb_slot[poc] = mirror slot a reference picture lives in until its last user is done.

    n_sops = None: an endless sequence (the steady state of the weak-scaling bench); an int: IDR + n_sops SOPs, after
    which slot() returns all-None (the strong-scaling run over a fixed sequence)."""

    def __init__(self, world: int, dpb_slots: Optional[int] = None, n_sops: Optional[int] = None, window: Optional[int] = None,
                 lag: Optional[int] = None):
        """lag: a picture may start `lag` slots after the last reference it gets FROM ANOTHER RANK was scheduled.  1 = the
        tightest schedule: that reference's broadcast (end of its slot) sits on the critical path of the next slot.  2 gives
        every broadcast a whole slot to complete underneath other pictures' kernels, at the price of a deeper pipeline (more
        pictures in the window, more mirror slots held).  The anchor chain POC 8k -> 8k + 8 always runs on rank 0 with lag 1:
        that dependency is local.  With 8 ranks lag 2 still schedules 8 pictures per slot (one SOP per slot, seven levels of
        skew); with 2-4 ranks a SOP spans several slots and the extra latency costs more than it hides, so the default is
        lag 2 for world >= 8 on an endless sequence and 1 otherwise (a short finite sequence is fill / drain bound)."""
        if lag is None:
            lag = 2 if (world >= 8 and n_sops is None) else 1
        dpb_slots = (24 if lag == 1 else 40) if dpb_slots is None else dpb_slots
        window = (32 if lag == 1 else 64) if window is None else window
        assert world >= 1 and dpb_slots >= 6 and lag >= 1
        self.world, self.n_slots_dpb, self.n_sops, self.window, self.lag = world, dpb_slots, n_sops, window, lag
        self.pics: List[Picture] = [Picture(0, 0, True, ())]
        self._sops = 0
        self.by_poc: Dict[int, Picture] = {0: self.pics[0]}
        self.done_slot: Dict[int, int] = {}          # poc -> slot it was scheduled in
        self.dpb_slot: Dict[int, int] = {}           # poc -> DPB mirror slot (reference pictures)
        self.rank_of: Dict[int, int] = {}            # poc -> rank that works on it
        self._free = list(range(dpb_slots))
        self._held: Dict[int, int] = {}              # poc -> mirror slot still held
        self._next = 0                               # oldest unscheduled coding-order index
        self._scheduled = set()
        self._plan: List[List[Optional[Picture]]] = []
        self._stalled = 0                            # consecutive slots that scheduled nothing while pictures remained

    # ---- sequence generation ---------------------------------------------------------------------------------------
    def _grow(self, upto_index: int):
        while len(self.pics) <= upto_index and (self.n_sops is None or self._sops < self.n_sops):
            base = 8 * self._sops
            last = None if self.n_sops is None else 8 * self.n_sops
            for off, is_ref, deltas in SOP8:
                poc = base + off
                refs = tuple(sorted({poc + d for d in deltas if poc + d >= 0 and (last is None or poc + d <= last)}))
                p = Picture(len(self.pics), poc, is_ref, refs)
                self.pics.append(p)
                self.by_poc[poc] = p
            self._sops += 1

    def _users_done(self, poc: int) -> bool:
        """every picture that references `poc` has been scheduled (users live in the SOP of poc and the next one)"""
        sop = (poc - 1) // 8 if poc > 0 else -1
        self._grow(1 + 8 * (sop + 2))                # SOPs sop and sop + 1 exist (or the sequence ends before)
        lo, hi = max(0, 1 + 8 * sop), min(len(self.pics), 1 + 8 * (sop + 2))
        return all(p.index in self._scheduled for p in self.pics[lo:hi] if poc in p.refs)

    def total_pictures(self) -> Optional[int]:
        return None if self.n_sops is None else 1 + 8 * self.n_sops

    # ---- the schedule ----------------------------------------------------------------------------------------------
    def slot(self, t: int) -> List[Optional[Picture]]:
        while len(self._plan) <= t:
            self._plan.append(self._make_slot(len(self._plan)))
        return self._plan[t]

    def _make_slot(self, t: int) -> List[Optional[Picture]]:
        # release the mirror slots whose picture has no unscheduled user left (users ran in slots < t)
        for poc in [q for q in self._held if self._users_done(q)]:
            # the sequence's last pictures keep their slots (nothing else needs them)
            if self.n_sops is None or poc < 8 * self.n_sops:
                self._free.append(self._held.pop(poc))
        self._free.sort()
        self._grow(self._next + self.window)
        chosen: List[Picture] = []
        i = self._next
        while len(chosen) < self.world and i < len(self.pics) and i < self._next + self.window:
            p = self.pics[i]
            i += 1
            if p.index in self._scheduled:
                continue
            # the anchor chain (POC 8k -> 8k + 8) stays on rank 0, so that dependency is local: no broadcast to wait for
            lag = 1 if (p.poc % 8 == 0 and self.world > 1) else self.lag
            if not all(self.done_slot.get(q, t) + lag <= t for q in p.refs):
                continue
            if p.is_reference:
                if not self._free:
                    continue
                s = self._free.pop(0)
                self.dpb_slot[p.poc] = s
                self._held[p.poc] = s
            chosen.append(p)
        # a slot that schedules nothing while pictures remain must be waiting for a reference that ran in the previous slot(s)
        # (`lag`) -- otherwise nothing can ever change again (too few mirror slots for this lag / world): fail, do not spin
        if not chosen and self._next < len(self.pics):
            self._stalled += 1
            if self._stalled > self.lag + 1:
                raise RuntimeError(f"DagSchedule cannot make progress at slot {t}: {len(self._free)} free of {self.n_slots_dpb} mirror slots, "
                                   f"{len(self._held)} held (lag {self.lag}, world {self.world}); give it more dpb_slots or a larger window")
        else:
            self._stalled = 0
        for p in chosen:
            self._scheduled.add(p.index)
            self.done_slot[p.poc] = t
        while self._next < len(self.pics) and self._next in self._scheduled:
            self._next += 1
        # rank 0 takes the slot's anchor picture (if any), the other ranks the rest in coding order; a rank without one idles
        chosen.sort(key=lambda p: (p.poc % 8 != 0, p.index))
        for r, p in enumerate(chosen):
            self.rank_of[p.poc] = r
        return [chosen[r] if r < len(chosen) else None for r in range(self.world)]

    def finished(self, t: int) -> bool:
        """True when the (finite) sequence is completely scheduled in slots < t"""
        if self.n_sops is None:
            return False
        self.slot(t)
        return self._next >= 1 + 8 * self.n_sops and all(p is None for p in self._plan[t])

    def slots_for_sequence(self) -> int:
        """number of slots the finite sequence takes on `world` ranks"""
        assert self.n_sops is not None
        t = 0
        while not self.finished(t):
            t += 1
        return t


def owner(pic_index: int, world: int) -> int:
    """round-robin dealing in coding order (the round-1 plan; kept for comparison in tests -- it ignores dependencies)"""
    return pic_index % world


class BandPlan:
    """A padded picture cut into bands of whole CTU rows, the granule of the reference's own wait-for-reference rule (turing/TaskEncodeSubstream.cpp:71-95: a CTU
    starts when the reference picture is reconstructed 4 CTUs to the right / 3 rows below; TaskDeblock.cpp:151-167 publishes the deblocked, padded rows).  Band b
    holds the picture's CTU rows [b * band_ctu_rows, (b + 1) * band_ctu_rows); the first band also carries the top border, the last one the bottom border, so the
    bands tile the padded planes exactly.  Chroma rows are the luma rows halved (4:2:0)."""

    def __init__(self, height: int, pad: int, luma_stride: int, chroma_stride: int, band_ctu_rows: int = 4, ctb: int = 64, reach_ctu_rows: int = 3):
        if height <= 0 or pad < 0 or pad % 2 or band_ctu_rows < 1 or luma_stride <= 0 or chroma_stride <= 0:
            raise ValueError("band plan: sizes")
        self.height, self.pad, self.ls, self.cs, self.ctb, self.reach = height, pad, luma_stride, chroma_stride, ctb, reach_ctu_rows
        self.ctu_rows = (height + ctb - 1) // ctb
        self.band_ctu_rows = band_ctu_rows
        self.n_bands = (self.ctu_rows + band_ctu_rows - 1) // band_ctu_rows
        self.luma_rows_total = height + 2 * pad
        self.chroma_rows_total = height // 2 + pad

    def luma_rows(self, b: int) -> Tuple[int, int]:
        """[first, last) rows of band b in the padded luma plane"""
        lo = 0 if b == 0 else self.pad + b * self.band_ctu_rows * self.ctb
        hi = self.luma_rows_total if b == self.n_bands - 1 else self.pad + (b + 1) * self.band_ctu_rows * self.ctb
        return lo, hi

    def chroma_rows(self, b: int) -> Tuple[int, int]:
        lo, hi = self.luma_rows(b)
        return lo // 2, (self.chroma_rows_total if b == self.n_bands - 1 else hi // 2)

    def pieces(self, b: int, n_luma: int, n_chroma: int) -> List[Tuple[int, int]]:
        """band b as [first, last) element ranges of a flat (luma | Cb | Cr) mirror buffer"""
        (l0, l1), (c0, c1) = self.luma_rows(b), self.chroma_rows(b)
        return [(l0 * self.ls, l1 * self.ls), (n_luma + c0 * self.cs, n_luma + c1 * self.cs), (n_luma + n_chroma + c0 * self.cs, n_luma + n_chroma + c1 * self.cs)]

    def band_of_ctu_row(self, row: int) -> int:
        return min(max(row, 0), self.ctu_rows - 1) // self.band_ctu_rows

    def bands_needed(self, ctu_row: int) -> int:
        """how many bands (0 .. n) of a reference must have arrived before CTU row `ctu_row` of a picture predicting from it may start: the rows down to
        `reach` CTU rows below it (the reference's rule), i.e. everything up to the band holding that row"""
        return self.band_of_ctu_row(ctu_row + self.reach) + 1

    def rows_ready(self, bands_arrived: int) -> int:
        """CTU rows of a dependent picture that may start when the first `bands_arrived` bands of its reference are in the mirror"""
        if bands_arrived >= self.n_bands:
            return self.ctu_rows
        return max(0, bands_arrived * self.band_ctu_rows - self.reach)


class ReferenceExchange:
    """Mirror of the decoded-picture buffer on every rank + the broadcast step of a DagSchedule.

    Every DPB mirror slot is ONE flat buffer (luma, Cb, Cr in the padded picture layout), so a reference picture costs
    one broadcast, not one per plane.  Per time slot t:

      stage(t, planes)  the rank that encoded a reference picture in slot t copies its reconstruction into the picture's
                        mirror slot (schedule.dpb_slot[poc]) -- from then on the reconstruction planes may be reused;
      send(t)           one broadcast per reference picture of the slot, owner -> every rank's mirror, issued by all
                        ranks in rank order.

    On the GPU both are enqueued on the caller's current stream (the broadcasts on RCCL's own stream behind it), so a
    caller that orders its compute stream only after `stage` overlaps the broadcasts of slot t with the computation of
    slot t + 1 (bench.py).  `refs(pic)` returns the mirror buffers a picture predicts from: a picture's kernels read
    their references FROM THE MIRROR, whoever encoded them."""

    def __init__(self, dist, rank: int, schedule: DagSchedule, n_luma: int, n_chroma: int, like, single_rank_broadcast: bool = False):
        import torch
        self._torch = torch
        self.dist, self.rank, self.world, self.schedule = dist, rank, schedule.world, schedule
        self.single_rank_broadcast = single_rank_broadcast   # exercise the collective even when world == 1
        self.nl, self.nc = n_luma, n_chroma
        self.dpb = [like.new_zeros(n_luma + 2 * n_chroma) for _ in range(schedule.n_slots_dpb)]
        self.dpb_luma = [b[:n_luma] for b in self.dpb]          # views
        self.dpb_cb = [b[n_luma:n_luma + n_chroma] for b in self.dpb]
        self.dpb_cr = [b[n_luma + n_chroma:] for b in self.dpb]
        self.sent_bytes = 0
        self.broadcasts = 0
        self.plan: Optional[BandPlan] = None
        self.arrived: Dict[int, int] = {}       # poc -> bands of it in the local mirror (band mode)

    def picture_of(self, t: int, rank: Optional[int] = None) -> Optional[Picture]:
        return self.schedule.slot(t)[self.rank if rank is None else rank]

    def slot_of(self, poc: int) -> int:
        return self.schedule.dpb_slot[poc]

    def refs(self, pic: Picture):
        """(L0 mirror buffer index, L1 mirror buffer index) of a picture, None for the IDR"""
        if not pic.refs:
            return None
        return self.slot_of(pic.l0), self.slot_of(pic.l1)

    def stage(self, t: int, planes):
        """planes = (luma, cb, cr) reconstruction of this rank's picture of slot t"""
        pic = self.picture_of(t)
        if pic is not None and pic.is_reference:
            s = self.slot_of(pic.poc)
            self._torch._foreach_copy_([self.dpb_luma[s], self.dpb_cb[s], self.dpb_cr[s]], list(planes))   # one launch

    # ---- the same exchange in CTU-row bands: a band leaves when its rows are deblocked and padded, the rest of the picture still computing ----
    def set_bands(self, plan: BandPlan):
        if plan.luma_rows_total * plan.ls > self.nl or plan.chroma_rows_total * plan.cs > self.nc:
            raise ValueError("band plan does not fit the mirror planes")
        self.plan = plan

    def stage_band(self, t: int, b: int, planes):
        """band b of this rank's reconstruction of slot t into the picture's mirror slot; planes = (luma, cb, cr) flat padded planes"""
        pic = self.picture_of(t)
        if pic is not None and pic.is_reference:
            buf = self.dpb[self.slot_of(pic.poc)]
            bases = (0, self.nl, self.nl + self.nc)
            for (lo, hi), plane, base in zip(self.plan.pieces(b, self.nl, self.nc), planes, bases):
                buf[lo:hi].copy_(plane[lo - base:hi - base])

    def send_band(self, t: int, b: int, async_op: bool = False):
        """band b of every reference picture of slot t, owner -> all; every rank calls it for the same (t, b) in the same order.  Returns the work handles when
        async_op (the caller waits before the rows that need the band start: rows_ready)."""
        works = []
        for src in range(self.world):
            pic = self.picture_of(t, src)
            if pic is None or not pic.is_reference:
                continue
            buf = self.dpb[self.slot_of(pic.poc)]
            for lo, hi in self.plan.pieces(b, self.nl, self.nc):
                piece = buf[lo:hi]
                if src == self.rank:
                    self.sent_bytes += piece.numel() * piece.element_size() * (self.world - 1)
                if self.world > 1 or self.single_rank_broadcast:
                    w = self.dist.broadcast(piece, src=src, async_op=async_op)
                    if async_op:
                        works.append(w)
                    self.broadcasts += 1
            self.arrived[pic.poc] = b + 1
        return works

    def rows_ready(self, poc: int) -> int:
        """CTU rows of a picture predicting from `poc` that may start now (band mode)"""
        return self.plan.rows_ready(self.arrived.get(poc, 0))

    def send(self, t: int):
        for src in range(self.world):
            pic = self.picture_of(t, src)
            if pic is None or not pic.is_reference:
                continue
            buf = self.dpb[self.slot_of(pic.poc)]
            if src == self.rank:
                self.sent_bytes += buf.numel() * buf.element_size() * (self.world - 1)
            if self.world > 1 or self.single_rank_broadcast:
                self.dist.broadcast(buf, src=src)
                self.broadcasts += 1
