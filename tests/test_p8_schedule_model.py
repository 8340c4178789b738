"""CPU: the hazard argument of the persistent 8-phase GEMM loop (surya_amd/csrc/gemm.h, gemm_nt_p8p_kernel) as an executable model.

DESIGN.md section 4b states two properties of the schedule -- every LDS fragment read is separated by a workgroup barrier from the
`s_waitcnt vmcnt` with which EVERY wave retired its share of that half-tile (RAW), and a slot is re-requested only after a barrier that all
readers of its previous occupant have passed (WAR) -- "properties of the table, not of clean runs". This file transcribes the kernel's
request / read / wait / barrier sequence (prologue, first two K-tiles of a tile without counted waits, middle pairs, the single-path tail
with the next tile's look-ahead requests, the groups' re-alignment around the wave-private epilogue, the stagger at the top of a tile) into
per-wave-group event lists and checks both properties for every request and read, across the two wave groups that run one barrier apart
and across tile seams. The GPU race screen (tools/microbench/p8_screen.py) and the bit-identity tests are the measurement; this is the
argument, kept honest by three perturbed schedules that the checker must reject.

What the model knows about the hardware (MI355X_MICROARCH.md / cdna_hip_programming.md): `s_barrier` is workgroup-wide, so the k-th
barrier call of every wave is the same event; `vmcnt(N)` returns when all but the wave's N most recent vector-memory operations (loads
AND stores, in issue order) are complete; an LDS-DMA load may land at any time between its issue and that wait; nothing else orders a
`ds_read` against another wave's pending LDS-DMA.
"""
from dataclasses import dataclass, field

import pytest

SLOT = {"X0": 0, "W0": 1, "X1": 2, "W1": 3}          # + 4 for odd K-tiles: eight 16 KiB slots = two K-tiles


def slot_of(half, kt):
    return (kt & 1) * 4 + SLOT[half]


@dataclass
class Wave:
    """Event list of one wave group. epoch = number of barrier calls before the event."""
    ops: list = field(default_factory=list)
    epoch: int = 0
    issued: int = 0                                   # vector-memory operations issued so far (requests: 2 each)
    pending: list = field(default_factory=list)       # [first_seq, n, key] of requests not yet known complete

    def bar(self):
        self.epoch += 1

    def req(self, tile, half, kt):
        key = (tile, half, kt)
        self.ops.append(("req", slot_of(half, kt), key, self.epoch))
        self.pending.append([self.issued, 2, key])
        self.issued += 2

    def other(self, n):                               # bias / residual / rotary loads, output stores: counted by vmcnt, touch no slot
        self.issued += n

    def vm(self, n):
        done = [p for p in self.pending if p[0] + p[1] <= self.issued - n]
        for p in done:
            self.ops.append(("retired", None, p[2], self.epoch))
        self.pending = [p for p in self.pending if p not in done]

    def read(self, tile, half, kt):
        self.ops.append(("read", slot_of(half, kt), (tile, half, kt), self.epoch))


def phase(w, reads, request, vm, tile):
    """SP_PHASE: fragment reads, one half-tile request, the counted wait, barrier, 8 MFMAs, barrier."""
    for half, kt in reads:
        w.read(tile, half, kt)
    if request is not None:
        w.req(*request)
    if vm is not None:
        w.vm(vm)
    w.bar()
    w.bar()


def program(group, tiles, nk, residual_loads=0, stores=16, stagger=True, realign=True, tail_wait=8, seam_waits=True):
    """One wave group's events over `tiles` tiles of `nk` (even, >= 4) K-tiles, transcribed from gemm_nt_p8p_kernel.
    The keyword arguments past `stores` switch OFF pieces of the schedule for the negative tests."""
    assert nk % 2 == 0 and nk >= 4
    w = Wave()
    # prologue: six half-tiles of tile 0, vmcnt(0), one barrier -- before the groups are staggered
    for half, kt in (("X0", 0), ("W0", 0), ("X1", 0), ("W1", 0), ("X0", 1), ("W0", 1)):
        w.req(0, half, kt)
    w.vm(0)
    w.bar()
    for tile in range(tiles):
        have_next = tile + 1 < tiles
        if group == 1 and stagger:
            w.bar()                                   # the second wave of every SIMD runs one barrier behind
        T = tile
        # first two K-tiles: phases 0..3 without a counted wait
        phase(w, [("X0", 0), ("W0", 0)], (T, "X1", 1), None, T)
        phase(w, [("X1", 0)],            (T, "W1", 1), None, T)
        phase(w, [("W1", 0)],            (T, "X0", 2), None, T)
        phase(w, [("X0", 1)],            (T, "W0", 2), None, T)
        phase(w, [("W0", 1)],            (T, "X1", 2), 8, T)
        phase(w, [("X1", 1)],            (T, "W1", 2), 8, T)
        phase(w, [("W1", 1)],            (T, "X0", 3), 8, T)
        phase(w, [("X0", 2)],            (T, "W0", 3), 8, T)
        for pi in range(1, nk // 2 - 1):
            t = 2 * pi
            phase(w, [("W0", t)],     (T, "X1", t + 1), 8, T)
            phase(w, [("X1", t)],     (T, "W1", t + 1), 8, T)
            phase(w, [("W1", t)],     (T, "X0", t + 2), 8, T)
            phase(w, [("X0", t + 1)], (T, "W0", t + 2), 8, T)
            phase(w, [("W0", t + 1)], (T, "X1", t + 2), 8, T)
            phase(w, [("X1", t + 1)], (T, "W1", t + 2), 8, T)
            phase(w, [("W1", t + 1)], (T, "X0", t + 3), 8, T)
            phase(w, [("X0", t + 2)], (T, "W0", t + 3), 8, T)
        # last two K-tiles: ONE path; with a next tile its first six half-tiles take the ring's next six requests
        t = nk - 2
        phase(w, [("W0", t)], (T, "X1", t + 1), 8, T)
        phase(w, [("X1", t)], (T, "W1", t + 1), 8, T)
        look = [("X0", 0), ("W0", 0), ("X1", 0), ("W1", 0), ("X0", 1), ("W0", 1)]
        reads = [[("W1", t)], [("X0", t + 1)], [("W0", t + 1)], [("X1", t + 1)], [("W1", t + 1)], []]
        for i in range(6):
            if i == 5:
                w.other(1)                            # the bias pair: one more operation in the request stream, older than phase 7's request
            if have_next:
                phase(w, reads[i], (T + 1, *look[i]), tail_wait if (seam_waits or i < 4) else None, T)
            else:
                phase(w, reads[i], None, (6, 4, 2, 0, None, None)[i], T)
        if group == 0 and realign:
            w.bar()                                   # the leading half meets the trailing half's last barrier
        # wave-private epilogue: residual rows, ONE vmcnt(0), then staging + stores only (no barrier inside)
        w.other(residual_loads)
        w.vm(0)
        w.other(stores)
    return w


def check(tiles, nk, **kw):
    """-> list of hazards (empty = the schedule is sound)."""
    g = [program(0, tiles, nk, **kw), program(1, tiles, nk, **kw)]
    if g[0].epoch != g[1].epoch:
        return [f"barrier counts differ between the groups: {g[0].epoch} vs {g[1].epoch} (a hang, or waves in different barriers)"]
    bad = []
    issue = [{}, {}]
    retire = [{}, {}]
    order = {}                                        # slot -> keys in request order (the same in both groups by construction)
    for gi, w in enumerate(g):
        for kind, slot, key, ep in w.ops:
            if kind == "req":
                issue[gi][key] = ep
                if gi == 0:
                    order.setdefault(slot, []).append(key)
            elif kind == "retired":
                retire[gi][key] = ep
    reads = {}
    for gi, w in enumerate(g):
        for kind, slot, key, ep in w.ops:
            if kind != "read":
                continue
            reads.setdefault(key, []).append((gi, ep))
            # RAW: every wave group's share retired in an EARLIER epoch (= a barrier in between)
            for gj in (0, 1):
                if key not in issue[gj]:
                    bad.append(f"group {gi} reads {key} which group {gj} never requested")
                elif retire[gj].get(key, 10 ** 9) >= ep:
                    bad.append(f"RAW: group {gi} reads {key} in epoch {ep}, group {gj} retires its share in epoch {retire[gj].get(key)}")
            # the slot holds what is read: no younger request to the slot issued before this read could have landed
            ks = order[slot]
            nxt = ks.index(key) + 1
            if nxt < len(ks):
                for gj in (0, 1):
                    if issue[gj][ks[nxt]] <= ep:
                        bad.append(f"WAR: {ks[nxt]} requested by group {gj} in epoch {issue[gj][ks[nxt]]}, group {gi} still reads {key} in epoch {ep}")
    # every requested half-tile of the tiles that run is consumed exactly once per group (a transcription check on the model itself)
    for slot, ks in order.items():
        for key in ks:
            n = reads.get(key, [])
            if sorted(gi for gi, _ in n) != [0, 1]:
                bad.append(f"{key}: read by groups {[gi for gi, _ in n]}")
    return bad


@pytest.mark.parametrize("nk", [4, 6, 10, 20, 128])
@pytest.mark.parametrize("tiles", [1, 2, 4])
@pytest.mark.parametrize("residual_loads", [0, 16])
def test_schedule_is_hazard_free(nk, tiles, residual_loads):
    assert check(tiles, nk, residual_loads=residual_loads) == []


def test_store_count_of_the_epilogue_does_not_matter():
    """Stores sit in the same vmcnt queue as the next tile's requests: more of them make phase 4's counted wait stricter, never weaker."""
    for stores in (0, 4, 64):
        assert check(3, 20, stores=stores) == []


def test_without_the_stagger_the_schedule_is_still_sound_but_groups_collide():
    """(The stagger is a throughput device -- one wave of a SIMD multiplies while the other loads -- not a correctness device.)"""
    assert check(2, 20, stagger=False, realign=False) == []


def test_checker_rejects_a_missing_realignment():
    """Staggered groups that are never re-aligned call different numbers of barriers per tile."""
    assert any("barrier counts differ" in h for h in check(2, 20, realign=False))


def test_checker_rejects_a_wait_that_is_too_loose():
    """vmcnt(10) in the tail leaves the half-tile the next phase reads possibly in flight."""
    assert any(h.startswith("RAW") for h in check(2, 20, tail_wait=10))


def test_checker_rejects_a_seam_without_the_last_two_counted_waits():
    """The next tile's phase 0 reads X0'(0) and W0'(0) BEFORE any barrier of that tile for the leading group -- in the same barrier interval
    as the other waves' epilogue vmcnt(0). What makes that read safe is the tail: the two half-tiles are requested in its phases 2 and 3 and
    retired by the counted waits of phases 6 and 7, with barriers behind them. Without those two waits the model must object."""
    hazards = check(2, 20, seam_waits=False)
    assert any(h.startswith("RAW") and "'X0', 0" in h for h in hazards) and any(h.startswith("RAW") and "'W0', 0" in h for h in hazards)
    assert check(1, 20, seam_waits=False) == []          # (a single tile has no seam)
