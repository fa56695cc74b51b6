"""A happens-before model of the persistent kernel's FOLLOWING scheme (scheme 2, psoap_amd/csrc/dag_kernel.hpp), built
from the host's own task list: test infrastructure (tests/test_protocol_hb.py), never imported by the product.

Round 6 found two places in that scheme where nothing but TIME ordered two tasks (the accumulator chain of the diagonal
tasks; the single monotone progress words potrf_done / rows_done, whose readers take "value >= q + 1" to mean "everything up to
q") -- both through rare wrong values on the device, one in 20,000 ... 400,000 matrices.  This model is the check that would
have found them on the CPU: every task is a sequence of events (waits, reads, writes, publications) taken from the kernel's
code paths (dag_special / dag_diag_fast / dag_pss / dag_update_following / the generic PART path; file:function in the
comments below), every wait adds the happens-before edges the hardware guarantees for it and NO others, and every shared
object's accesses are checked: a read of version v of an object must happen after the write of v and before the write of
v + 1; a write of v + 1 after the write of v.

What a wait guarantees.  A flag word holds the maximum its publishers have stored so far (they are monotone), so a wait for
`flag >= v` may be satisfied by ANY publication of a value >= v: it orders the waiter after an event E only if E happens
before EVERY such publication (an AND node).  A wait for `counter >= n` of a counter that A tasks add to is satisfied by
whichever n adds come first: it orders the waiter after E only if E happens before at least A - n + 1 of the adds (a threshold
node).  A publication that itself happens AFTER the wait (it depends, through other hand-offs, on what the waiter does next)
cannot be the one that satisfies it: such candidates are removed, and the relation recomputed, until nothing changes -- every
pass only uses facts established with MORE candidates, i.e. fewer edges, so every pass is sound.  Program order inside a task is a chain; the waves of a workgroup are not modelled apart (their joint progress is
what the kernel's barriers make it).

Options reproduce the two historic holes, so that the test can show the model SEES them:
  inorder=False    potrf_done / rows_done published without waiting for the predecessor's value (rounds 3-5)
  acc_chain=True   one accumulator record per matrix, read-modify-written by every diagonal task (rounds 1-5)
  rv_wait=False    a strip solve's update of the right-hand side without the wait for the row above's (round 3, until its
                   last day: "only the timing said so")
"""
from __future__ import annotations

import ctypes
from collections import defaultdict

import numpy as np

TASK = np.dtype([("type", "u1"), ("q", "u1"), ("j", "u1"), ("S", "u1"), ("b", "<u2"), ("pa", "u1"), ("pb", "u1"),
                 ("slot", "<u4"), ("ctr", "<u4")])
PART, DIAG, OFF, SCHUR = 0, 1, 2, 3
TYPE_MASK, CHAIN, NOSOLVE, WAITNEXT, FUSED = 0x0F, 0x10, 0x20, 0x40, 0x80


def aug_plan(P: int, Mt: int, Ms: int, scheme: int = 2):
    """the task list of predict's launch: one matrix of P block rows with Mt appended column tiles and the Ms x Ms upper tiles
    of their Schur complement (psoap_dag_plan_aug)"""
    from psoap_amd import _lib
    L = _lib.load()
    n, slots, ctrs = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong()
    first = (ctypes.c_uint32 * 9)()
    assert L.psoap_dag_plan_aug(P, Mt, Ms, 512, scheme, None, 0, ctypes.byref(n), ctypes.byref(slots), ctypes.byref(ctrs), first) == 0
    tasks = np.zeros(n.value, dtype=TASK)
    assert L.psoap_dag_plan_aug(P, Mt, Ms, 512, scheme, tasks.ctypes.data_as(ctypes.c_void_p), n.value, ctypes.byref(n),
                                ctypes.byref(slots), ctypes.byref(ctrs), first) == 0
    return tasks


def lane_plan(P: int, scheme: int = 2):
    """the task list ONE matrix of P block rows gets under `scheme` (psoap_stream_plan: what every stream lane runs; a single
    evaluation per launch gets the same structure)"""
    from psoap_amd import _lib
    L = _lib.load()
    n, slots, ctrs, sch = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_int()
    assert L.psoap_stream_plan(P, 8, 511, scheme, None, 0, ctypes.byref(n), ctypes.byref(slots), ctypes.byref(ctrs),
                               ctypes.byref(sch)) == 0
    tasks = np.zeros(n.value, dtype=TASK)
    assert L.psoap_stream_plan(P, 8, 511, scheme, tasks.ctypes.data_as(ctypes.c_void_p), n.value, ctypes.byref(n),
                               ctypes.byref(slots), ctypes.byref(ctrs), ctypes.byref(sch)) == 0
    assert sch.value == scheme
    return tasks


class Graph:
    """events, happens-before edges, AND / threshold nodes; reach(src) = everything guaranteed to happen after src"""

    def __init__(self):
        self.succ = defaultdict(list)
        self.need = {}            # threshold node -> number of predecessors that must be reached
        self.names = []
        self._reach = {}

    def node(self, name) -> int:
        self.names.append(name)
        return len(self.names) - 1

    def edge(self, a: int, b: int):
        self.succ[a].append(b)

    def threshold(self, name, preds, need) -> int:
        n = self.node(name)
        self.need[n] = max(int(need), 1)
        for p in preds:
            self.succ[p].append(n)
        return n

    def reach(self, src: int) -> set:
        got = self._reach.get(src)
        if got is not None:
            return got
        seen = {src}
        hits = defaultdict(int)
        work = [src]
        while work:
            u = work.pop()
            for v in self.succ.get(u, ()):
                if v in seen:
                    continue
                if v in self.need:
                    hits[v] += 1
                    if hits[v] < self.need[v]:
                        continue
                seen.add(v)
                work.append(v)
        self._reach[src] = seen
        return seen

    def hb(self, a: int, b: int) -> bool:
        return a == b or b in self.reach(a)


class Model:
    def __init__(self, P: int, inorder: bool = True, acc_chain: bool = False, rv_wait: bool = True, Mt: int = 0, Ms: int = 0):
        self.P, self.inorder, self.acc_chain, self.rv_wait = P, inorder, acc_chain, rv_wait
        self.Pt = P + Mt               # column tiles: the matrix's + the appended ones (predict: [B | Cx^T])
        self.g = Graph()
        self.tasks = aug_plan(P, Mt, Ms, 2) if Mt else lane_plan(P, 2)
        self.flags = self.tasks["type"].copy()
        self.kind = self.tasks["type"] & TYPE_MASK
        self.pubs = defaultdict(list)        # flag name -> [(value, event)]
        self.adds = defaultdict(list)        # counter name -> [event]
        self.waits = []                      # (flag or counter name, target, event after the wait, is_counter)
        self.writes = defaultdict(dict)      # object -> {version: event}
        self.reads = []                      # (object, version, event, who)
        self.last = {}                       # ticket -> its last event (the completion count of its row)
        self.launch = self.g.node("launch")  # the state the launch starts from (flags zero, r = fl - mu, ...)
        self.n_parts = defaultdict(int)
        for t, k in enumerate(self.tasks):
            if self.kind[t] == PART:
                self.n_parts[int(k["ctr"])] += 1
        self.build()

    # ---- events of one task, in program order
    class Seq:
        def __init__(self, model, ticket, label):
            self.m, self.t, self.label = model, ticket, label
            self.cur = model.g.node(f"{label}:start")
            model.g.edge(model.launch, self.cur)

        def step(self, what) -> int:
            n = self.m.g.node(f"{self.label}:{what}")
            self.m.g.edge(self.cur, n)
            self.cur = n
            return n

        def wait(self, flag, target, counter=False):
            if target <= 0:
                return
            n = self.step(f"wait {flag}>={target}")
            self.m.waits.append((flag, int(target), n, counter))

        def publish(self, flag, value):
            self.m.pubs[flag].append((int(value), self.step(f"publish {flag}={value}")))

        def add(self, counter):
            self.m.adds[counter].append(self.step(f"add {counter}"))

        def read(self, obj, version):
            self.m.reads.append((obj, version, self.step(f"read {obj} v{version}"), self.label))

        def write(self, obj, version):
            self.m.writes[obj][version] = self.step(f"write {obj} v{version}")

    def read_tiles(self, s, rows, cols):
        for r in rows:
            for c in set(cols):
                for b in range(8):
                    s.read(("tile", r, c, b), 0)

    def rv_version_before(self, q, j):
        """version of the right-hand-side block j that the strip solve of tile (q, j) reads: one per earlier row that applied
        its contribution itself (dag_pss: the tile right of the diagonal is skip_rv -- the diagonal task applies it)"""
        return sum(1 for r in range(q) if j >= r + 2)

    def build(self):
        P, tasks, flags = self.P, self.tasks, self.flags
        row_members = defaultdict(list)
        for t, k in enumerate(tasks):
            q, j, S, pa, pb = int(k["q"]), int(k["j"]), int(k["S"]), int(k["pa"]), int(k["pb"])
            ctr, slot = int(k["ctr"]), int(k["slot"])
            f = int(flags[t])
            chain = bool(f & CHAIN)
            if self.kind[t] == PART:
                # the generic path of k_chol_dag: dag_update (look-ahead form), the wait for the chain's predecessor BEHIND the
                # update, dag_sub_partials, dag_store_updated into the slot, dag_drain, release, arrive += 1
                s = self.Seq(self, t, f"PART({q},{j})#{S}")
                assert chain
                if pb > pa:
                    if pb - pa > 1:
                        s.wait("rows_done", pb - 1)
                        self.read_tiles(s, range(pa, pb - 1), (q, j))
                    s.wait("rows_done", pb)
                    self.read_tiles(s, (pb - 1,), (q, j))
                if S > 0:
                    s.wait(("arrive", ctr), S, counter=True)
                    s.read(("slot", ctr, (S - 1) & 1), S - 1)
                s.write(("slot", ctr, S & 1), S)
                s.add(("arrive", ctr))
                self.last[t] = s.cur
                continue
            n_wait = S - 1
            if self.kind[t] == SCHUR:
                # a tile of Sigma = A - W^T W (the generic path: the chain's sum, the last panel behind rows_done, the store
                # into DagAug::S; nobody inside the launch reads it, it is not counted in any row)
                s = self.Seq(self, t, f"SCHUR({q},{j})")
                if n_wait > 0:
                    s.wait(("arrive", ctr), n_wait, counter=True)
                    s.read(("slot", ctr, (n_wait - 1) & 1), n_wait - 1)
                assert pb - pa == 1 and pb == P
                s.wait("rows_done", pb)
                self.read_tiles(s, (pb - 1,), (q, j))
                s.write(("sigma", q, j), 0)
                self.last[t] = s.cur
                continue
            if self.kind[t] == DIAG:
                # dag_special mode 0 -> dag_diag_fast -> potrf_spine_fused (spine / worker, SpinePub, SpineFollow)
                assert chain and (f & WAITNEXT) and n_wait >= 1 and not (f & FUSED)
                xfollow = bool(f & NOSOLVE)
                two = pb - pa == 2
                s = self.Seq(self, t, f"DIAG({q})")
                s.wait(("arrive", ctr), n_wait, counter=True)
                s.read(("slot", ctr, (n_wait - 1) & 1), n_wait - 1)          # `part`, read before wait_dep()
                if not xfollow:
                    s.wait("next_done", q)
                elif two:
                    s.wait("rows_done", q - 1)
                s.read(("rv", q), self.rv_version_before(q, q))              # the right-hand side blocks (worker<W>)
                if xfollow:
                    if two:
                        self.read_tiles(s, (q - 2,), (q,))
                    for b in range(8):
                        s.wait(("xcol", q), 8 * (q - 1) + b + 1)
                        s.read(("tile", q - 1, q, b), 0)
                        s.read(("mb", (q - 1) & 1, b), (q - 1) // 2)          # x_zblock: z_b of the factorisation above
                elif q > 0:
                    self.read_tiles(s, (q - 1,), (q,))
                for b in range(8):
                    s.write(("mb", q & 1, b), q // 2)                        # pub_block
                    s.publish("step_w", 8 * q + b + 1)                       # pub_flag
                for b in range(8):
                    s.write(("tile", q, q, b), 0)
                s.write(("rv", q), self.rv_version_before(q, q) + 1)         # z_q
                if self.acc_chain:
                    if q > 0:
                        s.read(("acc",), q - 1)
                    s.write(("acc",), q)
                else:
                    s.write(("acc", q), 0)
                if self.inorder and q > 0:
                    s.wait("potrf_done", q)                                  # dag_spin_ge
                s.publish("potrf_done", q + 1)
                self.last[t] = s.cur
                row_members[q].append(t)
                continue
            # a following strip solve: dag_special mode 1 (dag_update_following, dag_pss)
            assert self.kind[t] == OFF and (f & WAITNEXT) and chain == (S > 1)
            xlink, pubnext = bool(f & FUSED), bool(f & NOSOLVE)
            assert xlink, "the second level of following is what the model describes (PSOAP_XFOLLOW default)"
            s = self.Seq(self, t, f"OFF({q},{j})")
            if n_wait > 0:
                s.wait(("arrive", ctr), n_wait, counter=True)
                s.read(("slot", ctr, (n_wait - 1) & 1), n_wait - 1)
            if q >= 1:
                if pb - pa > 1:
                    s.wait("rows_done", pb - 1)
                    self.read_tiles(s, range(pa, pb - 1), (q, j))
                for b in range(8):                                           # dag_update_following: stage b = row block b
                    s.wait(("xcol", q), 8 * (q - 1) + b + 1)
                    s.wait(("xcol", j), 8 * (q - 1) + b + 1)
                    s.read(("tile", q - 1, q, b), 0)
                    s.read(("tile", q - 1, j, b), 0)
            for b in range(8):                                               # dag_pss
                s.wait("step_w", 8 * q + b + 1)
                s.read(("mb", q & 1, b), q // 2)
                s.write(("tile", q, j, b), 0)                                # xpub: row block b goes out ...
                s.publish(("xcol", j), 8 * q + b + 1)                        # ... and its flag rises behind it
            if not (xlink and pubnext):                                      # (skip_rv: tile (q, q+1), taken by DIAG(q+1))
                s.wait("potrf_done", q + 1)
                if q > 0 and self.rv_wait:
                    s.wait(("rvrow", j), q)
                v = self.rv_version_before(q, q)
                s.read(("rv", q), v + 1)                                     # z_q
                if j < P:                                                    # (an appended column has no right-hand side block)
                    vj = self.rv_version_before(q, j)
                    s.read(("rv", j), vj)
                    s.write(("rv", j), vj + 1)
            if pubnext:
                s.publish("next_done", q + 1)
            s.publish(("rvrow", j), q + 1)
            self.last[t] = s.cur
            row_members[q].append(t)
        # rows_done: published by whichever task of the row counts last (dag_task_done) -- after all of them, and, in
        # order, after the row above has been published
        g = self.g
        for q in range(P):
            members = [self.last[t] for t in row_members[q]]
            assert len(members) == self.Pt - q
            # (a JOIN: the publication comes after ALL of them, so it comes after whichever one an event precedes)
            n = g.node(f"rows_done={q + 1}")
            for mbr in members:
                g.edge(mbr, n)
            if self.inorder and q > 0:
                g.edge(self.pubs["rows_done"][-1][1], n)
            self.pubs["rows_done"].append((q + 1, n))
        # the report: behind every task of the matrix (stream_retire's count / the kernel boundary) -- a join as well
        rep = g.node("report")
        for ev in self.last.values():
            g.edge(ev, rep)
        if self.acc_chain:
            self.reads.append((("acc",), P - 1, rep, "report"))
        else:
            for q in range(P):
                self.reads.append((("acc", q), 0, rep, "report"))
        # the waits: AND over every publication that could satisfy them / threshold over the adders
        self.wait_nodes = []                 # [node, event after the wait, candidate events, n]: n = 0 for a flag (every
                                             # candidate may be the satisfier), else the count the counter has to reach
        for flag, target, ev, counter in self.waits:
            cands = self.adds[flag] if counter else [e for v, e in self.pubs[flag] if v >= target]
            assert len(cands) >= (target if counter else 1), f"nobody ever brings {flag} to {target}"
            n = g.node(f"{flag}>={target}")
            g.edge(n, ev)
            self.wait_nodes.append([n, ev, set(cands), target if counter else 0])
        self.refine()

    def refine(self):
        """a publication (an add) that happens after the wait it would satisfy does not satisfy it: iterate to the fixed point"""
        g = self.g
        base = {k: list(v) for k, v in g.succ.items()}
        while True:
            g.succ = defaultdict(list, {k: list(v) for k, v in base.items()})
            g._reach = {}
            for n, ev, cands, count in self.wait_nodes:
                # a flag: ordered after E iff E precedes EVERY candidate; a counter that must reach `count`: iff E precedes
                # all but count - 1 of the candidate adds
                g.need[n] = len(cands) if count == 0 else len(cands) - count + 1
                for c in cands:
                    g.succ[c].append(n)
            changed = False
            for w in self.wait_nodes:
                n, ev, cands, count = w
                if len(cands) <= max(count, 1):
                    continue
                after = g.reach(ev)
                drop = {c for c in cands if c in after}
                if drop and len(cands) - len(drop) >= max(count, 1):
                    cands -= drop
                    changed = True
            if not changed:
                return

    # ---- the check
    def races(self, limit: int = 20):
        g, out = self.g, []
        for obj, vers in self.writes.items():
            order = sorted(vers)
            for a, b in zip(order, order[1:]):
                if not g.hb(vers[a], vers[b]):
                    out.append(f"write {obj} v{a} is not ordered before the write of v{b}: {g.names[vers[a]]} || {g.names[vers[b]]}")
        for obj, v, ev, who in self.reads:
            if v == 0 and obj[0] in ("rv",):
                w = self.launch            # the right-hand side as the launch starts with it
            else:
                w = self.writes[obj].get(v)
                if w is None and obj[0] == "tile":
                    out.append(f"{who} reads {obj} that nobody writes")
                    continue
            if w is not None and not g.hb(w, ev):
                out.append(f"{who}: the read of {obj} v{v} is not ordered after its write: {g.names[w]} || {g.names[ev]}")
            nxt = self.writes[obj].get(v + 1)
            if nxt is not None and not g.hb(ev, nxt):
                out.append(f"{who}: the read of {obj} v{v} is not ordered before the next write: {g.names[ev]} || {g.names[nxt]}")
            if len(out) >= limit:
                break
        return out


class Model01(Model):
    """The same check for the task lists of scheme 1 (latency: the row-to-row chain inside fused diagonal tasks) and scheme 0
    (throughput: tile-level dependencies, the strip solve of tile (q, q+1) continuing into diagonal task q+1), whose tasks run
    the kernel's generic path (dag_update / dag_store_updated / potrf_blocked / dag_trsm) or, for scheme 1's chained diagonal
    tasks, dag_diag_fast with its fused strip solve.  A tile has two versions here: 0 = updated (in memory between the update
    and the strip solve), 1 = final."""

    def __init__(self, P: int, scheme: int, inorder: bool = True):
        self.scheme = scheme
        self.P, self.inorder, self.acc_chain, self.rv_wait = P, inorder, False, True
        self.g = Graph()
        self.tasks = lane_plan(P, scheme)
        self.flags = self.tasks["type"].copy()
        self.kind = self.tasks["type"] & TYPE_MASK
        self.pubs, self.adds, self.waits = defaultdict(list), defaultdict(list), []
        self.writes, self.reads, self.last = defaultdict(dict), [], {}
        self.launch = self.g.node("launch")
        self.build01()

    def read_final(self, s, rows, cols):
        for r in rows:
            for c in set(cols):
                s.read(("tile", r, c), 1)

    def update(self, s, q, j, pa, pb, wait_next=False):
        """dag_update: rows < pb - 1 first, the last panel behind its own wait (scheme 0: the two column tiles' progress words
        instead of whole rows)"""
        if pb <= pa:
            return
        td = self.scheme == 0

        def wait(v):
            if td:
                s.wait(("rvrow", q), v)
                if j != q:
                    s.wait(("rvrow", j), v)
            else:
                s.wait("next_done" if wait_next else "rows_done", v)
        if pb - pa > 1:
            if td:
                s.wait(("rvrow", q), pb - 1)
                if j != q:
                    s.wait(("rvrow", j), pb - 1)
            else:
                s.wait("rows_done", pb - 1)
            self.read_final(s, range(pa, pb - 1), (q, j))
        wait(pb)
        self.read_final(s, (pb - 1,), (q, j))

    def build01(self):
        P, tasks, flags, sch = self.P, self.tasks, self.flags, self.scheme
        row_members = defaultdict(list)
        counts_two = set()
        prev_end = None
        n_parts = defaultdict(int)
        for t, k in enumerate(tasks):
            if self.kind[t] == PART:
                n_parts[int(k["ctr"])] += 1
        for t, k in enumerate(tasks):
            q, j, S, pa, pb = int(k["q"]), int(k["j"]), int(k["S"]), int(k["pa"]), int(k["pb"])
            ctr, slot = int(k["ctr"]), int(k["slot"])
            f = int(flags[t])
            chain = bool(f & CHAIN)
            wt = ("wt", q % 3 if sch == 0 else q & 1)
            wt_v = q // 3 if sch == 0 else q // 2
            if self.kind[t] == PART:
                s = self.Seq(self, t, f"PART({q},{j})#{S if chain else slot}")
                self.update(s, q, j, pa, pb)
                if chain:
                    if S > 0:
                        s.wait(("arrive", ctr), S, counter=True)
                        s.read(("slot", ctr, (S - 1) & 1), S - 1)
                    s.write(("slot", ctr, S & 1), S)
                else:
                    s.write(("gslot", slot), 0)
                s.add(("arrive", ctr))
                self.last[t] = s.cur
                prev_end = s.cur
                continue
            n_wait = S - 1
            preload = chain and n_wait > 0

            def gather(s):
                if n_wait > 0:
                    s.wait(("arrive", ctr), n_wait, counter=True)
                    if chain:
                        s.read(("slot", ctr, (n_wait - 1) & 1), n_wait - 1)
                    else:
                        for u in range(n_wait):
                            s.read(("gslot", slot + u), 0)

            def factor(s):
                """potrf_blocked / potrf_spine_fused outputs + the in-order publication"""
                s.write(("tile", q, q), 1)
                s.write(wt, wt_v)
                s.write(("rv", q), q + 1)                        # z_q
                s.write(("acc", q), 0)
                if self.inorder and q > 0:
                    s.wait("potrf_done", q)
                s.publish("potrf_done", q + 1)

            def trsm(s, jj):
                """dag_trsm on tile (q, jj): W, the updated tile, z_q; the solved tile; the right-hand side block jj"""
                s.read(wt, wt_v)
                s.read(("tile", q, jj), 0)
                s.read(("rv", q), q + 1)
                s.write(("tile", q, jj), 1)
                s.read(("rv", jj), q)
                s.write(("rv", jj), q + 1)

            if self.kind[t] == DIAG:
                s = self.Seq(self, t, f"DIAG({q})")
                if sch == 0 and (f & NOSOLVE):
                    # owned: run by the workgroup that ran the strip solve of tile (q-1, q), right behind it (k_chol_dag, CONT)
                    assert prev_end is not None and int(tasks[t - 1]["q"]) == q - 1 and int(tasks[t - 1]["j"]) == q
                    self.g.edge(prev_end, s.cur)
                fast = sch == 1 and preload and (f & WAITNEXT) and pb - pa == 1
                if fast:
                    # dag_diag_fast, xfollow = false: the chain's sum, then next_done, then the strip above
                    gather(s)
                    s.wait("next_done", q)
                    s.read(("rv", q), q)
                    self.read_final(s, (q - 1,), (q,))
                else:
                    if preload:
                        gather(s)
                    self.update(s, q, q, pa, pb, wait_next=bool(f & WAITNEXT))
                    if not preload:
                        gather(s)
                    s.write(("tile", q, q), 0)
                    if sch == 0:
                        s.wait("rows_done", q - 2 if q >= 3 else 0)
                    s.read(("tile", q, q), 0)
                    s.read(("rv", q), q)
                factor(s)
                if sch == 1 and (f & FUSED):
                    s.wait("off1_ready", q + 1)
                    trsm(s, q + 1)
                    s.publish("next_done", q + 1)
                    counts_two.add(t)
                self.last[t] = s.cur
                row_members[q].append(t)
                prev_end = s.cur
                continue
            s = self.Seq(self, t, f"OFF({q},{j})")
            if preload:
                gather(s)
            self.update(s, q, j, pa, pb)
            if not preload:
                gather(s)
            s.write(("tile", q, j), 0)
            if sch == 1 and (f & NOSOLVE):
                s.publish("off1_ready", q + 1)                     # update only: the fused diagonal task solves it
                self.last[t] = s.cur
                prev_end = s.cur
                continue
            s.wait("potrf_done", q + 1)
            trsm(s, j)
            self.last[t] = s.cur
            row_members[q].append(t)
            s_end = s.cur
            if sch == 0:
                s.publish(("rvrow", j), q + 1)                     # (behind the row's count: see k_chol_dag)
            prev_end = s.cur
            self.last[t] = s_end
        g = self.g
        for q in range(P):
            members = [self.last[t] for t in row_members[q]]
            assert len(members) + sum(1 for t in row_members[q] if t in counts_two) == P - q, (q, len(members))
            n = g.node(f"rows_done={q + 1}")
            for mbr in members:
                g.edge(mbr, n)
            if self.inorder and q > 0:
                g.edge(self.pubs["rows_done"][-1][1], n)
            self.pubs["rows_done"].append((q + 1, n))
        rep = g.node("report")
        for ev in self.last.values():
            g.edge(ev, rep)
        for q in range(P):
            self.reads.append((("acc", q), 0, rep, "report"))
        self.wait_nodes = []
        for flag, target, ev, counter in self.waits:
            cands = self.adds[flag] if counter else [e for v, e in self.pubs[flag] if v >= target]
            assert len(cands) >= (target if counter else 1), f"nobody ever brings {flag} to {target}"
            n = g.node(f"{flag}>={target}")
            g.edge(n, ev)
            self.wait_nodes.append([n, ev, set(cands), target if counter else 0])
        self.refine()
