"""CPU validation of the persistent kernel's scheduler: the host-built task list must cover every
tile and every finished block row exactly once, and every wait must target a smaller ticket."""
import ctypes

import numpy as np
import pytest

TASK = np.dtype([("type", "u1"), ("q", "u1"), ("j", "u1"), ("S", "u1"), ("b", "<u2"), ("pa", "u1"), ("pb", "u1"),
                 ("slot", "<u4"), ("ctr", "<u4")])
PART, DIAG, OFF = 0, 1, 2
TYPE_MASK, CHAIN, NOSOLVE, WAITNEXT, FUSED = 0x0F, 0x10, 0x20, 0x40, 0x80


def plan(B, P, workers):
    """``P``: block rows of every matrix (uniform batch) or a list with one entry per matrix."""
    from psoap_amd import _lib
    L = _lib.load()
    n, slots, ctrs = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
    first = (ctypes.c_uint32 * 9)()
    if np.ndim(P) == 0:
        call = lambda out, cap: L.psoap_dag_plan(B, P, workers, out, cap, ctypes.byref(n), ctypes.byref(slots),   # noqa: E731
                                                 ctypes.byref(ctrs), first)
    else:
        Ps = (ctypes.c_int * B)(*P)
        call = lambda out, cap: L.psoap_dag_plan_multi(B, Ps, workers, out, cap, ctypes.byref(n),                 # noqa: E731
                                                       ctypes.byref(slots), ctypes.byref(ctrs), first)
    assert call(None, 0) == 0
    tasks = np.zeros(n.value, dtype=TASK)
    assert call(tasks.ctypes.data_as(ctypes.c_void_p), n.value) == 0
    plan.queue_first = list(first)
    plan.chain = (tasks["type"] & CHAIN) != 0
    plan.flags = tasks["type"].copy()
    tasks["type"] &= TYPE_MASK
    return tasks, slots.value, ctrs.value


@pytest.mark.parametrize("B,P,workers", [(1, 1, 512), (1, 2, 512), (1, 47, 512), (3, 5, 512), (7, 16, 512),
                                         (32, 47, 512), (32, 64, 512), (5, 20, 8), (200, 4, 512),
                                         # heterogeneous batches: matrices of several chunks in one launch
                                         (4, [3, 9, 1, 16], 512), (12, [16, 13, 11, 16, 20, 7, 16, 13, 11, 16, 20, 7], 512),
                                         (40, [47, 32] * 20, 512), (9, [2, 40, 2, 2, 2, 2, 2, 2, 5], 64)])
def test_plan_is_complete_and_deadlock_free(B, P, workers):
    tasks, n_slots, n_ctrs = plan(B, P, workers)
    chain = plan.chain
    Ps = [P] * B if np.ndim(P) == 0 else list(P)
    n_tiles = sum(p * (p + 1) // 2 for p in Ps)
    assert TASK.itemsize == 16
    # latency scheme (chained partial sums) while the fullest queue holds at most ~92 block rows (round 6; 150 before) or a
    # single matrix; gathered otherwise
    # (queues in use, dag_queue_count: of 8, 4, 2, 1 the one that leaves the fullest queue relatively emptiest -- 12 matrices:
    # 4 x 3, 9 matrices: one list -- the larger on a tie; up to 8 matrices a queue each)
    nq = 8 if B <= 8 else min((8, 4, 2, 1), key=lambda n: ((B + n - 1) // n * n, -n))
    q_rows = [sum(Ps[g::nq]) for g in range(nq)]
    q_count = [len(Ps[g::nq]) for g in range(nq)]
    want_chain = max(q_rows) * nq <= 92 * 8 or max(q_count) <= 1
    # ... and within that, the following scheme (scheme 2) for up to eight matrices (24 small ones): the strip solves follow the
    # factorisation (DAG_WAITNEXT on OFF finals), the diagonal task solves nothing
    following = want_chain and (B <= 8 or (B <= 24 and max(Ps) <= 20))
    assert np.all(chain[tasks["S"] > 1] == want_chain) if (tasks["S"] > 1).any() else True
    # one queue per XCD: matrix b lives in queue b mod nq, queues are contiguous ranges of the list
    first = plan.queue_first
    assert first[0] == 0 and first[8] == len(tasks) and all(first[g] <= first[g + 1] for g in range(8))
    for g in range(8):
        assert np.all(tasks["b"][first[g]:first[g + 1]] % nq == g)
    finals = {}
    covered = {}           # (b, q, j) -> list of (pa, pb)
    part_done_ticket = {}  # ctr -> list of tickets of its PARTs
    slots_seen = set()
    pair_owner = {}        # chained partial sums: slot pair -> arrival counter of the chain
    diag_final_ticket = {}
    row_final_last_ticket = {}   # (b, q) -> max ticket of the row's finals
    for t, k in enumerate(tasks):
        key = (int(k["b"]), int(k["q"]), int(k["j"]))
        assert k["b"] < B and k["j"] >= k["q"] and k["q"] < Ps[int(k["b"])] and k["j"] < Ps[int(k["b"])]
        assert k["pa"] <= k["pb"] <= k["q"]
        covered.setdefault(key, []).append((int(k["pa"]), int(k["pb"])))
        if k["type"] == PART:
            assert k["slot"] < n_slots and k["ctr"] < n_ctrs
            if chain[t]:
                # a chain ping-pongs between the two slots of an even/odd pair that no other chain uses
                pair = int(k["slot"]) >> 1
                assert pair_owner.setdefault(pair, int(k["ctr"])) == int(k["ctr"])
                assert int(k["slot"]) & 1 == int(k["S"]) & 1
            else:
                assert int(k["slot"]) not in slots_seen and (int(k["slot"]) >> 1) not in pair_owner
            slots_seen.add(int(k["slot"]))
            part_done_ticket.setdefault(int(k["ctr"]), []).append(t)
        else:
            assert key not in finals, "one final task per tile"
            finals[key] = t
            assert (k["type"] == DIAG) == (k["j"] == k["q"])
            if k["type"] == DIAG:
                diag_final_ticket[(key[0], key[1])] = t
            rk = (key[0], key[1])
            row_final_last_ticket[rk] = max(row_final_last_ticket.get(rk, -1), t)
    # every upper tile of every matrix has exactly one final
    assert len(finals) == n_tiles
    for b in range(B):
        assert sum(1 for key in finals if key[0] == b) == Ps[b] * (Ps[b] + 1) // 2
    assert all(sl < n_slots for sl in slots_seen)
    for key, ranges in covered.items():
        b, q, j = key
        ranges.sort()
        # the parts tile [0, q) exactly
        pos = 0
        for pa, pb in ranges:
            assert pa == pos
            pos = pb
        assert pos == q
    # dependency order: every wait targets a smaller ticket
    flags = plan.flags
    for t, k in enumerate(tasks):
        b, q, j = int(k["b"]), int(k["q"]), int(k["j"])
        # the update over block rows [pa, pb) needs rows < pb complete: all finals of those rows are earlier
        # -- except a diagonal final of the latency scheme (WAITNEXT), which reads of the last row only the
        # tile right of its diagonal, solved by that row's DIAG task itself (FUSED)
        follows = bool(flags[t] & WAITNEXT) and k["type"] == OFF      # on an OFF final the bit means "follow the factorisation"
        wait_next = bool(flags[t] & WAITNEXT) and not follows
        if follows:
            assert following and diag_final_ticket[(b, q)] < t      # behind the task whose progress it polls
            if q >= 1:
                # the last panel of its update follows the two tiles of the row above it reads -- both delivered row block
                # by row block (FUSED) by following strip solves with smaller tickets
                for src in (finals[(b, q - 1, q)], finals[(b, q - 1, j)]):
                    assert src < t and flags[src] & WAITNEXT and flags[src] & FUSED
        two_panel_diag = k["type"] == DIAG and bool(flags[t] & NOSOLVE) and q >= 3   # (a following diagonal task: two panels)
        if wait_next:
            assert want_chain and k["type"] == DIAG and int(k["pb"]) == q
            assert int(k["pb"]) - int(k["pa"]) == (2 if two_panel_diag else min(q, 1))
        if wait_next and q >= 1:
            prev = diag_final_ticket[(b, q - 1)]
            if following:
                # the tile right of the diagonal above is solved by its own (following) task, which announces it
                solver = finals[(b, q - 1, q)]
                assert solver < t and (flags[solver] & NOSOLVE) and (flags[solver] & WAITNEXT) and not flags[prev] & FUSED
            else:
                assert prev < t and flags[prev] & FUSED
        if want_chain:
            for m in range(int(k["pb"]) - (1 if wait_next else 0)):
                assert row_final_last_ticket[(b, m)] < t
        else:
            # scheme 0 (round 4), tile-level dependencies: an update over [pa, pb) reads the tiles (m, q) and (m, j), m < pb,
            # whose finals are earlier; an OWNED diagonal final (NOSOLVE) sits right behind the final of the tile above
            # its diagonal (FUSED) and is run by that task's workgroup; before it writes its W tile (three in turn), block
            # row q - 3 is through
            for m in range(int(k["pb"])):
                assert finals[(b, m, q)] < t and finals[(b, m, j)] < t
            if k["type"] == DIAG and q >= 1:
                assert flags[t] & NOSOLVE and not flags[t] & (WAITNEXT | FUSED)
                assert finals[(b, q - 1, q)] == t - 1 and flags[t - 1] & FUSED and int(k["pb"]) == q
                for m in range(q - 2):
                    assert row_final_last_ticket[(b, m)] < t
            elif k["type"] == DIAG:
                assert not flags[t] & (NOSOLVE | WAITNEXT | FUSED)
            if flags[t] & FUSED:
                assert k["type"] == OFF and j == q + 1 and q + 1 < Ps[b] and diag_final_ticket[(b, q + 1)] == t + 1
        # the second level of the following scheme: the strip solve of tile (q, q+1), q >= 2, delivers its tile row block
        # by row block (FUSED on a following OFF final) to the diagonal task of block q+1 (NOSOLVE on a DIAG final), which
        # must be one the out-of-line fast path takes: chained, a one-panel final with a running sum to start from
        xpub = follows and bool(flags[t] & FUSED)
        xdiag = want_chain and k["type"] == DIAG and bool(flags[t] & NOSOLVE)
        assert xpub == follows           # every following strip solve delivers its tile progressively
        if xpub and j == q + 1 and q + 1 < Ps[b]:
            nxt = diag_final_ticket[(b, q + 1)]
            assert flags[t] & NOSOLVE and flags[nxt] & NOSOLVE and t < nxt
        if xdiag:
            assert following and q >= 1 and wait_next and chain[t] and k["S"] >= 2
            src = finals[(b, q - 1, q)]
            assert src < t and flags[src] & FUSED and flags[src] & WAITNEXT
        if flags[t] & FUSED and not xpub and want_chain:
            # DIAG(q) also solves tile (q, q+1): its update-only task comes earlier in the list
            assert k["type"] == DIAG and q + 1 < Ps[b]
            upd = finals[(b, q, q + 1)]
            assert upd < t and flags[upd] & NOSOLVE
        if flags[t] & NOSOLVE and not xdiag and want_chain:
            assert k["type"] == OFF and j == q + 1
            assert bool(flags[diag_final_ticket[(b, q)]] & FUSED) == (not follows)
        if following and k["type"] == DIAG:
            # (every diagonal task is the fused fast one that publishes its steps -- it starts from the running sum of a
            # chain, which for blocks 0 and 1 is a PART over no panels: the covariance tile)
            assert not flags[t] & FUSED and wait_next and chain[t] and k["S"] >= 2
        elif want_chain and k["type"] == DIAG:
            assert bool(flags[t] & FUSED) == (q + 1 < Ps[b]) and wait_next == (q >= 1)
        if not want_chain:
            assert not flags[t] & WAITNEXT
        if k["type"] != PART:
            if k["S"] > 1:
                parts = part_done_ticket[int(k["ctr"])]
                assert len(parts) == k["S"] - 1 and max(parts) < t
                assert parts == sorted(parts)
                got = [int(tasks[p]["slot"]) for p in parts]
                for p in parts:
                    assert (int(tasks[p]["b"]), int(tasks[p]["q"]), int(tasks[p]["j"])) == (b, q, j)
                    assert chain[p] == chain[t]
                if chain[t]:
                    # chained: PART s has index S == s and waits for its predecessor (a smaller ticket);
                    # the final reads the last PART's slot and covers only the last panel -- the one
                    # that depends on the block row above
                    assert [int(tasks[p]["S"]) for p in parts] == list(range(len(parts)))
                    assert got == [(got[0] & ~1) + (i & 1) for i in range(len(parts))]     # even/odd ping-pong
                    assert int(k["slot"]) == got[-1]
                    # (a following strip solve takes the last TWO panels: its chain then needs the row before the row
                    # above only, and the hand-over of the partial tile is off the row-to-row path)
                    assert int(k["pb"]) - int(k["pa"]) == (2 if follows or two_panel_diag else min(q, 1))
                else:
                    # gathered: PARTs wait for nothing, the final reads all of them from the first slot
                    assert all(int(tasks[p]["S"]) == 0 for p in parts)
                    assert got == list(range(got[0], got[0] + len(parts)))                 # consecutive slots
                    assert int(k["slot"]) == got[0]
            if k["type"] == OFF and not flags[t] & NOSOLVE:
                assert diag_final_ticket[(b, q)] < t       # potrf(q): a smaller ticket (scheme 0: the record behind tile (q-1, q))


def test_sparse_rows_are_split_and_diagonal_is_preaccumulated():
    tasks, _, _ = plan(32, 47, 512)
    off = tasks[tasks["type"] == OFF]
    # full rows are not split, the last block rows are
    assert off[off["q"] == 10]["S"].max() == 1
    assert off[off["q"] == 45]["S"].min() >= 4
    diag = tasks[tasks["type"] == DIAG]
    # from block row 2 on the diagonal final only covers the last finished row
    late = diag[diag["q"] >= 2]
    assert np.all(late["pb"] - late["pa"] == 1) and np.all(late["S"] >= 2)


@pytest.mark.parametrize("B,nq", [(1, 1), (5, 5), (8, 8), (9, 1), (10, 2), (12, 4), (13, 1), (16, 8), (17, 1), (20, 4), (24, 8),
                                  (25, 1), (28, 4), (31, 1), (32, 8), (256, 8)])
def test_ticket_queues_in_use_follow_the_batch_size(B, nq):
    """dag_queue_count: of 8, 4, 2, 1 queues the number that leaves the fullest queue relatively emptiest (ceil(B / n) x n),
    the larger on a tie; up to eight matrices a queue each."""
    plan(B, 6, 512)
    first = plan.queue_first
    assert sum(first[g + 1] > first[g] for g in range(8)) == nq


def test_chain_bound_batches_get_one_workgroup_per_compute_unit():
    """dag_pick_workers: one persistent workgroup per compute unit -- the kernels compiled for one wave per SIMD -- while the
    batch's algorithmic flops stay below 3.3e9 x the block rows of its largest matrix (single evaluations, small batches of
    small matrices), all the device admits otherwise; measured crossovers of DESIGN.md 3.3 and round 4's."""
    from psoap_amd import _lib
    L = _lib.load()

    def pick(Ps, Mt=0, cus=256, mx=512):
        arr = (ctypes.c_int * len(Ps))(*Ps)
        w = ctypes.c_int(0)
        assert L.psoap_dag_pick_workers(len(Ps), arr, Mt, cus, mx, ctypes.byref(w)) == 0
        return w.value

    assert pick([47]) == 256 and pick([47] * 2) == 256 and pick([47] * 3) == 512 and pick([47] * 32) == 512   # N = 6000
    assert pick([16] * 8) == 256 and pick([16] * 32) == 512                                                    # N = 2000
    assert pick([32] * 2) == 256 and pick([32] * 4) == 256 and pick([32] * 6) == 512                           # N = 4096
    assert pick([64]) == 256 and pick([64] * 2) == 512                                                         # N = 8192
    assert pick([64], Mt=24) == 512                      # predict at the retrieve shape: the appended columns are work
    assert pick([5], cus=256, mx=256) == 256             # a device that admits one workgroup per CU anyway
    assert pick([47], cus=304, mx=608) == 304


@pytest.mark.parametrize("P,Mt,Ms,scheme", [(1, 2, 2, 1), (2, 3, 3, 1), (9, 4, 4, 1), (64, 24, 24, 1), (47, 5, 5, 0),
                                            (16, 6, 3, 1), (20, 24, 24, -1)])
def test_schur_tiles_of_the_augmented_launch(P, Mt, Ms, scheme):
    """predict: the Ms x Ms upper tiles of Sigma = A - W^T W as tasks of the factorisation's own launch.  Every tile is a
    chain of PARTs over [0, P-1) plus a final over the last panel; the chain ping-pongs between a slot pair of its own;
    every wait (the predecessor in the chain, the block row a part needs) is satisfied by a smaller ticket."""
    from psoap_amd import _lib
    L = _lib.load()
    SCHUR = 3
    n, slots, ctrs = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
    first = (ctypes.c_uint32 * 9)()
    assert L.psoap_dag_plan_aug(P, Mt, Ms, 512, scheme, None, 0, ctypes.byref(n), ctypes.byref(slots), ctypes.byref(ctrs), first) == 0
    tasks = np.zeros(n.value, dtype=TASK)
    assert L.psoap_dag_plan_aug(P, Mt, Ms, 512, scheme, tasks.ctypes.data_as(ctypes.c_void_p), n.value, ctypes.byref(n),
                                ctypes.byref(slots), ctypes.byref(ctrs), first) == 0
    base = np.zeros(0, dtype=TASK)
    nb = ctypes.c_longlong()
    assert L.psoap_dag_plan_aug(P, Mt, 0, 512, scheme, None, 0, ctypes.byref(nb), None, None, None) == 0
    ty = tasks["type"] & TYPE_MASK
    chain = (tasks["type"] & CHAIN) != 0
    schur = (tasks["q"] >= P)
    assert schur.sum() == n.value - nb.value            # nothing else changes in the list
    assert np.all(chain[schur]) and np.all(tasks["j"][schur] >= tasks["q"][schur]) and np.all(tasks["j"][schur] < P + Ms)
    assert (ty[schur] == SCHUR).sum() == Ms * (Ms + 1) // 2 and not np.any(ty[~schur] == SCHUR)
    # the last ticket that finishes a block row: a task over [pa, pb) may only be handed out behind the tasks of rows < pb
    row_last = {}
    for t, k in enumerate(tasks):
        if k["q"] < P and ty[t] != PART:
            row_last[int(k["q"])] = max(row_last.get(int(k["q"]), -1), t)
    seen_pairs = {}
    for i in range(Ms):
        for j in range(i, Ms):
            sel = np.where(schur & (tasks["q"] == P + i) & (tasks["j"] == P + j))[0]
            parts = [t for t in sel if ty[t] == PART]
            fin = [t for t in sel if ty[t] == SCHUR]
            assert len(fin) == 1 and all(t < fin[0] for t in parts)
            assert parts == sorted(parts) and [int(tasks["S"][t]) for t in parts] == list(range(len(parts)))
            assert int(tasks["S"][fin[0]]) == len(parts) + 1 and int(tasks["pa"][fin[0]]) == P - 1 and int(tasks["pb"][fin[0]]) == P
            # the parts tile [0, P-1) without gaps, every one at least one panel long
            edges = [(int(tasks["pa"][t]), int(tasks["pb"][t])) for t in parts]
            assert all(a < b for a, b in edges)
            assert [a for a, _ in edges] == ([0] + [b for _, b in edges[:-1]] if edges else [])
            assert (edges[-1][1] if edges else 0) == P - 1
            # one arrival counter per tile, one even/odd slot pair per chain
            assert len({int(tasks["ctr"][t]) for t in sel}) == 1
            if parts:
                pair = int(tasks["slot"][parts[0]]) >> 1
                assert seen_pairs.setdefault(pair, (i, j)) == (i, j)
                for t in parts:
                    assert int(tasks["slot"][t]) >> 1 == pair and (int(tasks["slot"][t]) & 1) == (int(tasks["S"][t]) & 1)
                assert int(tasks["slot"][fin[0]]) == int(tasks["slot"][parts[-1]]) and int(tasks["slot"][fin[0]]) < slots.value
            # block rows a task reads are finished by smaller tickets
            for t in list(parts) + fin:
                for row in range(int(tasks["pa"][t]), int(tasks["pb"][t])):
                    assert row_last[row] < t, (i, j, t, row)
    # no other chain shares a Schur tile's slot pair
    others = np.where(~schur & chain & (ty == PART))[0]
    assert not ({int(tasks["slot"][t]) >> 1 for t in others} & set(seen_pairs))


@pytest.mark.parametrize("P", [3, 16, 32, 47])
def test_following_scheme_task_flags_and_ticket_order(P):
    """Scheme 2: the diagonal task solves nothing (no DAG_FUSED), every strip solve follows it (DAG_WAITNEXT on an OFF
    final) and therefore sits BEHIND it in the ticket order, and the strip solve of tile (q, q+1) -- flagged DAG_NOSOLVE,
    here "publishes next_done" -- sits in front of the next diagonal task, which follows it."""
    from psoap_amd import _lib
    L = _lib.load()
    n, slots, ctrs = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
    first = (ctypes.c_uint32 * 9)()
    assert L.psoap_dag_plan_aug(P, 0, 0, 256, 2, None, 0, ctypes.byref(n), ctypes.byref(slots), ctypes.byref(ctrs), first) == 0
    tasks = np.zeros(n.value, dtype=TASK)
    assert L.psoap_dag_plan_aug(P, 0, 0, 256, 2, tasks.ctypes.data_as(ctypes.c_void_p), n.value, ctypes.byref(n),
                                ctypes.byref(slots), ctypes.byref(ctrs), first) == 0
    ty = tasks["type"] & TYPE_MASK
    flags = tasks["type"]
    diag_ticket, off_tickets = {}, {}
    for t, k in enumerate(tasks):
        q, j = int(k["q"]), int(k["j"])
        if ty[t] == DIAG:
            diag_ticket[q] = t
            assert not flags[t] & FUSED, (q, hex(flags[t]))
            assert bool(flags[t] & NOSOLVE) == (q >= 1), (q, hex(flags[t]))       # follows the strip solve of tile (q-1, q)
        elif ty[t] == OFF:
            off_tickets.setdefault(q, {})[j] = t
            assert flags[t] & WAITNEXT, (q, j)                                    # follows
            assert bool(flags[t] & NOSOLVE) == (j == q + 1), (q, j)                 # the tile the next diagonal waits for
            assert flags[t] & FUSED, (q, j)                                        # delivers its tile row block by row block
    assert sorted(diag_ticket) == list(range(P))
    for q in range(P):
        for j, t in off_tickets.get(q, {}).items():
            assert diag_ticket[q] < t, (q, j)                  # a follower waits on its leader's progress
        if q + 1 < P:
            assert off_tickets[q][q + 1] < diag_ticket[q + 1]  # what DIAG(q+1) follows comes from a smaller ticket
    # every tile exactly once, as in the other schemes
    assert sum(len(v) for v in off_tickets.values()) == P * (P - 1) // 2


@pytest.mark.parametrize("P,scheme", [(1, 0), (2, 0), (16, 0), (47, 0), (64, 0), (16, 1), (47, 1), (16, 2), (32, 2)])
def test_lane_plan_of_a_stream(P, scheme):
    """psoap_stream_plan: the task list every lane of a stream runs (one matrix).  It depends on P and the scheme only
    (not on the lane count: results are bit-identical for every lane count); DagTask::b carries the burst marks -- the
    last ticket of a block row's worth of tasks; scheme 0: tile-level dependencies with owned diagonal finals."""
    from psoap_amd import _lib
    L = _lib.load()

    def lane_plan(lanes):
        n, slots, ctrs, sch = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_int(-5)
        assert L.psoap_stream_plan(P, lanes, 511, scheme, None, 0, ctypes.byref(n), ctypes.byref(slots), ctypes.byref(ctrs),
                                   ctypes.byref(sch)) == 0
        tasks = np.zeros(n.value, dtype=TASK)
        assert L.psoap_stream_plan(P, lanes, 511, scheme, tasks.ctypes.data_as(ctypes.c_void_p), n.value, ctypes.byref(n),
                                   ctypes.byref(slots), ctypes.byref(ctrs), ctypes.byref(sch)) == 0
        assert sch.value == scheme
        return tasks, slots.value, ctrs.value

    tasks, n_slots, n_ctrs = lane_plan(32)
    for lanes in (1, 8, 64):
        other, s2, c2 = lane_plan(lanes)
        assert np.array_equal(other, tasks) and (s2, c2) == (n_slots, n_ctrs)
    ty = tasks["type"] & TYPE_MASK
    flags = tasks["type"]
    marks = tasks["b"]
    assert set(np.unique(marks)) <= {0, 0x8000} and marks[-1] == 0x8000
    # every tile exactly once
    finals = {(int(k["q"]), int(k["j"])): t for t, k in enumerate(tasks) if ty[t] != PART}
    assert len(finals) == P * (P + 1) // 2
    ends = np.where(marks == 0x8000)[0]
    if scheme == 0:
        # a burst = what was emitted for one block row: the parts that pre-accumulate DIAG(q+1), the row's strip solves with
        # their parts, the owned DIAG(q+1) behind the strip solve of tile (q, q+1)
        section = np.where((tasks["q"] == tasks["j"]) & (tasks["q"] > 0), tasks["q"].astype(int) - 1, tasks["q"].astype(int))
        assert np.all(np.diff(section) >= 0)
        assert list(ends) == [t for t in range(len(tasks)) if t + 1 == len(tasks) or section[t + 1] != section[t]]
        for q in range(1, P):
            t = finals[(q, q)]
            assert flags[t] & NOSOLVE and finals[(q - 1, q)] == t - 1 and flags[t - 1] & FUSED
        for t, k in enumerate(tasks):
            q, j = int(k["q"]), int(k["j"])
            for m in range(int(k["pb"])):
                assert finals[(m, q)] < t and finals[(m, j)] < t          # the tiles an update reads: smaller tickets
            if ty[t] == OFF:
                assert finals[(q, q)] < t                                  # potrf(q)
    else:
        # schemes 1, 2 (lists in order of readiness, short tasks): the lanes are served ticket by ticket in turn
        assert list(ends) == list(range(len(tasks)))


# ---------------------------------------------------------------------------------------------- ready-only hand-out (round 5)
def plan_pool(Ps, workers, Mt=0, Ms=0, scheme=-1):
    from psoap_amd import _lib
    L = _lib.load()
    B = len(Ps)
    arr = (ctypes.c_int * B)(*Ps)
    n, ctrs = ctypes.c_longlong(), ctypes.c_longlong()
    first, n_main = (ctypes.c_uint32 * 9)(), (ctypes.c_uint32 * 8)()
    has = ctypes.c_int(0)
    assert L.psoap_dag_plan_pool(B, arr, workers, Mt, Ms, scheme, None, 0, ctypes.byref(n), None, None, n_main, first,
                                 ctypes.byref(has), ctypes.byref(ctrs)) == 0
    tasks = np.zeros(n.value, dtype=TASK)
    order = np.zeros(n.value, dtype=np.uint32)
    dep = np.zeros(n.value, dtype=np.uint32)
    assert L.psoap_dag_plan_pool(B, arr, workers, Mt, Ms, scheme, tasks.ctypes.data_as(ctypes.c_void_p), n.value, ctypes.byref(n),
                                 order.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                 dep.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), n_main, first, ctypes.byref(has),
                                 ctypes.byref(ctrs)) == 0
    return tasks, order, dep, list(n_main), list(first), bool(has.value), ctrs.value


def _simulate_pool(tasks, order, dep, n_main, first, Ps, Pt, workers, rng, window=64):
    """The hand-out of dag_kernel.hpp: pool_take played through with `workers` workgroups in rounds.  A final that has been
    handed out completes once everything it waits for is complete (its chain, the finals of the block rows its panels read,
    the diagonal final of its own row); a part is only ever TAKEN when it is ready and completes in the next round.
    Returns the number of rounds; raises when a round passes with work left, nothing running to completion and nothing
    taken (a deadlock of the hand-out)."""
    NONE = 0xFFFFFFFF
    ty = tasks["type"] & TYPE_MASK
    n = len(tasks)
    done = np.zeros(n, dtype=bool)
    taken = np.zeros(n, dtype=bool)              # by position in order[]
    nxt = [0] * 8
    lo = [0] * 8
    row_finals = {}                              # (b, q) -> task ids of the row's finals (not SCHUR)
    diag_of = {}
    parts_of = {}                                # ctr -> part task ids
    for t in range(n):
        b, q = int(tasks["b"][t]), int(tasks["q"][t])
        if ty[t] == PART:
            parts_of.setdefault(int(tasks["ctr"][t]), []).append(t)
        else:
            if q < Ps[b]:
                row_finals.setdefault((b, q), []).append(t)
            if ty[t] == DIAG:
                diag_of[(b, q)] = t
    rows_done = lambda b, m: all(done[x] for x in row_finals[(b, m)])      # noqa: E731

    pos_of = {int(order[p]): p for p in range(n)}

    def part_ready(t):
        b = int(tasks["b"][t])
        if not all(rows_done(b, m) for m in range(int(tasks["pb"][t]))):
            return False
        if (tasks["type"][t] & CHAIN) and tasks["S"][t] > 0:
            return bool(taken[int(dep[pos_of[t]])])         # its predecessor has been handed out
        return True

    def part_can_finish(t):
        if (tasks["type"][t] & CHAIN) and tasks["S"][t] > 0:
            return sum(done[x] for x in parts_of[int(tasks["ctr"][t])]) >= int(tasks["S"][t])
        return True

    def final_can_finish(t):
        b, q = int(tasks["b"][t]), int(tasks["q"][t])
        if tasks["S"][t] > 1 and not all(done[x] for x in parts_of[int(tasks["ctr"][t])]):
            return False
        if not all(rows_done(b, m) for m in range(min(int(tasks["pb"][t]), Ps[b]))):
            # (a diagonal final of the latency schemes reads only the tile right of the diagonal above: weaker than this)
            if not (ty[t] == DIAG and all(rows_done(b, m) for m in range(int(tasks["pb"][t]) - 1)) and
                    done[[x for x in row_finals[(b, q - 1)] if tasks["j"][x] == q][0]]):
                return False
        if ty[t] == OFF and not (tasks["type"][t] & NOSOLVE and not tasks["type"][t] & WAITNEXT) and q < Ps[b]:
            d = diag_of[(b, q)]
            # (a strip solve needs its row's factorisation; in the following scheme it FOLLOWS it: the diagonal task must
            # have been handed out, and -- scheme 1 -- the fused diagonal task in turn waits for the update-only task)
            if not (done[d] or running_set.__contains__(d)):
                return False
        return True

    running = []                                 # task ids
    running_set = set()
    pendings = []                                # finals drawn ahead of their turn: (position in order[]) held by a workgroup
    rounds = 0
    burst = max(2, workers // 8)                 # how many workgroups see the same head final before its ticket counter moves

    def runnable(pos):
        d = int(dep[pos])
        return d == NONE or taken[d]

    while not done.all():
        rounds += 1
        progressed = False
        still = []
        for t in running:
            if (ty[t] == PART and part_can_finish(t)) or (ty[t] != PART and final_can_finish(t)):
                done[t] = True
                running_set.discard(t)
                progressed = True
            else:
                still.append(t)
        running = still
        # workgroups that hold a final ahead of its turn: run it once its chain's last part has been taken
        keep = []
        for pos in pendings:
            if runnable(pos):
                running.append(int(order[pos]))
                running_set.add(int(order[pos]))
                progressed = True
            else:
                keep.append(pos)
        pendings = keep
        idle = workers - len(running) - len(pendings)
        helpers = len(pendings)                  # ... and take ready parts meanwhile (one each per round)
        queues = list(range(8))
        seen_head = {}
        for w in range(idle + helpers):
            helper = w >= idle
            got = None
            rng.shuffle(queues)
            for g in queues:
                f0, n_all = first[g], first[g + 1] - first[g]
                if n_all == 0:
                    continue
                if not helper:
                    # the head as this workgroup sees it: up to `burst` workgroups read the counter before any of their
                    # fetch-adds lands, so all of them draw when the head may be handed out -- the overshoot of pool_take
                    m_seen = seen_head.setdefault(g, [nxt[g], 0])
                    if m_seen[1] >= burst:
                        m_seen[0], m_seen[1] = nxt[g], 0
                    m_seen[1] += 1
                    m = m_seen[0]
                    if m < n_main[g] and runnable(f0 + m):
                        tk = nxt[g]
                        nxt[g] += 1
                        if tk < n_main[g]:
                            if runnable(f0 + tk):
                                got = int(order[f0 + tk])
                            else:
                                pendings.append(f0 + tk)
                                got = -1
                            break
                p0, p1 = f0 + n_main[g], f0 + n_all
                while lo[g] < p1 - p0 and taken[p0 + lo[g]]:
                    lo[g] += 1
                cur = p0 + lo[g]
                base = cur & ~31
                for pos in range(max(base, cur), min(base + window, p1)):
                    if not taken[pos] and part_ready(int(order[pos])):
                        taken[pos] = True
                        got = int(order[pos])
                        break
                if got is not None:
                    break
            if got is None:
                continue
            progressed = True
            if got >= 0:
                running.append(got)
                running_set.add(got)
        assert progressed, (f"hand-out deadlock after {rounds} rounds: {int((~done).sum())} tasks left, running {running[:8]}, "
                            f"held {pendings[:8]}")
        assert rounds < 100000
    assert not pendings
    return rounds


@pytest.mark.parametrize("Ps,workers,Mt,Ms,scheme", [([47], 256, 0, 0, -1), ([47], 256, 0, 0, 1), ([16], 256, 0, 0, -1),
                                                     ([64], 256, 0, 0, -1), ([1], 256, 0, 0, -1), ([2], 256, 0, 0, 2), ([3], 4, 0, 0, 2),
                                                     ([47] * 8, 512, 0, 0, -1), ([16] * 24, 512, 0, 0, -1), ([47] * 16, 512, 0, 0, 1),
                                                     ([16, 13, 11, 16, 20, 7], 512, 0, 0, -1), ([47], 1, 0, 0, -1), ([20], 3, 0, 0, 1),
                                                     ([64], 512, 24, 24, -1), ([9], 256, 4, 4, 1), ([16], 8, 6, 3, 2)])
def test_ready_only_hand_out_is_complete_and_cannot_deadlock(Ps, workers, Mt, Ms, scheme):
    """DagPool (dag_kernel.hpp): finals in list order -- a final with a chain only once the chain's last part is taken --
    and parts taken only when ready.  The orders cover every task once, a final's `dep` is its chain's last part, a chain's
    parts appear in chain order, and the hand-out played through with many, few and ONE workgroup (random queue order)
    always completes."""
    tasks, order, dep, n_main, first, has, n_ctrs = plan_pool(Ps, workers, Mt, Ms, scheme)
    assert has, "the latency schemes build the hand-out orders"
    ty = tasks["type"] & TYPE_MASK
    n = len(tasks)
    assert sorted(order.tolist()) == list(range(n))
    NONE = 0xFFFFFFFF
    for g in range(8):
        f0, f1 = first[g], first[g + 1]
        main = order[f0:f0 + n_main[g]]
        pool = order[f0 + n_main[g]:f1]
        assert np.all(ty[main] != PART) and np.all(ty[pool] == PART)
        assert np.all(np.diff(main.astype(np.int64)) > 0) and np.all(np.diff(pool.astype(np.int64)) > 0)     # list order kept
        assert np.all((main >= f0) & (main < f1)) and np.all((pool >= f0) & (pool < f1))
        for m in range(f0, f0 + n_main[g]):
            t = int(order[m])
            if tasks["S"][t] > 1:
                d = int(dep[m])
                assert f0 + n_main[g] <= d < f1
                last = int(order[d])
                assert ty[last] == PART and tasks["ctr"][last] == tasks["ctr"][t]
                mine = [int(x) for x in pool if tasks["ctr"][x] == tasks["ctr"][t]]
                assert mine[-1] == last and len(mine) == tasks["S"][t] - 1
                if tasks["type"][t] & CHAIN:
                    assert [int(tasks["S"][x]) for x in mine] == list(range(len(mine)))
            else:
                assert int(dep[m]) == NONE
        # a chained part's dep: the position of its predecessor in the chain
        for p in range(f0 + n_main[g], f1):
            t = int(order[p])
            if (tasks["type"][t] & CHAIN) and tasks["S"][t] > 0:
                pr = int(order[int(dep[p])])
                assert int(dep[p]) < p and ty[pr] == PART and tasks["ctr"][pr] == tasks["ctr"][t] and tasks["S"][pr] == tasks["S"][t] - 1
            else:
                assert int(dep[p]) == NONE
    Pt = [p + Mt for p in Ps]
    rng = np.random.default_rng(5)
    for w in sorted({workers, 1, 7}):
        _simulate_pool(tasks, order, dep, n_main, first, Ps, Pt, w, rng)


def test_throughput_scheme_has_no_hand_out_orders():
    _, _, _, n_main, _, has, _ = plan_pool([47] * 32, 512)
    assert not has and n_main == [0] * 8
