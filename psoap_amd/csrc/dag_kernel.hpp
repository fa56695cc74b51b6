// dag_kernel.hpp -- the whole batched factorisation as ONE persistent launch.
//
// The staged path (chol_kernels.hpp) issues three kernels per 128-row panel; every panel
// boundary quantises the work into rounds of resident workgroups and puts the diagonal-block
// factorisation on the critical path of the whole batch.  Here the same tile operations are
// tasks of a dependency graph executed by persistent 256-thread workgroups (2 per CU):
//
//   DIAG(b,q)    tile (q,q) of matrix b: left-looking MFMA update over the q finished block
//                rows, then an in-block Cholesky of the 128 x 128 tile (potrf_blocked.hpp) that also yields
//                U11^-T (operand of the strip solve), z_q = U11^-T r_q and the logdet/quad sums.
//   OFF(b,q,j)   tile (q,j), j > q: the same MFMA update, then X = U11^-T (tile) as a K=128 MFMA
//                product, then r[j-block] -= X^T z_q.
//
// Tasks are drawn from atomic ticket counters in the order (q, DIAGs first, then b, j), so a task
// only ever waits for tasks with smaller tickets of the same queue, which are held by running
// workgroups: no co-residency assumption, no deadlock, and while one matrix waits for its diagonal
// block the workgroups work on the other matrices of the batch.  There is one queue per XCD: matrix
// b belongs to queue b mod 8 and a workgroup serves the queue of the XCD it runs on (HW_REG_XCC_ID)
// before stealing from the others, so the tiles of one block row -- which all stream the same
// A operand -- run side by side under one L2 (measured before: 21 % L2 hit rate with one global
// queue, the shared operand being fetched once per XCD).  Hand-offs follow the agent-scope
// release/acquire recipe of cdna_hip_programming.md Guideline 16: plain stores, every wave drains
// (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane release fence + drain, relaxed agent-scope
// atomic on the counter; consumers poll relaxed, ONE acquire fence, drain, barrier, plain vector
// loads.  Every spin is bounded and reports through DagCtl::error instead of hanging the GPU.
//
// Per matrix three words: rows_done (block rows fully finished), potrf_done (diagonal blocks
// factored), cnt (finished tasks of the current block row).  The update of block row q reads
// rows < q; it is split so that rows < q-1 are consumed before waiting for row q-1 (look-ahead).
#pragma once
#include <stdlib.h>

// The following scheme (scheme 2: strip solves that follow the factorisation step by step, dag_pss / dag_special) is part
// of the build unless -DPSOAP_NO_FOLLOW is given (the build's fallback rung and the variant matrix of DESIGN.md 3.4 use
// the older structure, in which the fused diagonal task is the only out-of-line routine).
#if !defined(PSOAP_NO_FOLLOW) && !defined(PSOAP_FOLLOW)
#define PSOAP_FOLLOW 1
#endif

#include <algorithm>
#include <utility>
#include <vector>

#include "chol_kernels.hpp"
#include "fill_kernels.hpp"
#include "potrf_blocked.hpp"
#include "potrf_spine.hpp"
#include "orbit_kernels.hpp"

namespace psoap {

struct alignas(64) MatFlags {
    int rows_done;    // block rows completely finished
    int potrf_done;   // diagonal blocks factored
    int cnt[2];       // finished tasks of block row q in cnt[q & 1]: the latency scheme starts DIAG(q+1) while
                      // row q is still being solved, so two rows count at the same time (never three: DIAG(q+2)
                      // needs all of row q)
    int next_done;    // q + 1 once U(q, q+1), the tile right of the diagonal, is final (fused into DIAG(q))
    int off1_ready;   // q + 1 once tile (q, q+1) holds its fully updated value, ready for the strip solve
    int step_w[4];    // per wave of the fused diagonal task: 8 q + b + 1 once its share of step b of block q (the row
                      // of U11 and W_bb^T the strip solves need) is in the matrix's mailbox (dag_pss)
    int cnt3;         // scheme 0 (round 4): the counter of block rows q = 2 (mod 3) -- with tile-level dependencies DIAG(q)
                      // runs as soon as tile (q-1, q) is final, up to three block rows are being counted at a time
    int pad[5];
    int xcol[256][2]; // per column tile j, per publishing wave of the following strip solve of tile (q, j): 8 q + b + 1
                      // once its half of row block b of the solved tile is in memory -- the tasks of block row q+1 that
                      // read the tile (the diagonal task of block q+1, the strip solves of tiles (q+1, .)) follow it in turn
    int rvrow[256];   // per column tile j: q + 1 once the strip solve of tile (q, j) has applied its contribution to the
                      // right-hand side block j and finished (the next row's strip solve of that column waits for it before
                      // its own: a plain read-modify-write)
};
static_assert(sizeof(MatFlags) == 64 + 2048 + 1024, "one cache line of row state + the per-column progress words");

// Mailbox of a matrix, right behind its two Wt tiles: the fused diagonal task publishes, step by step, what a strip
// solve needs of block row b of the diagonal block it is factoring -- the blocks U_bJ (J > b) of U11's row b and
// V_b = W_bb^T -- so that the strip solves of the row can FOLLOW the factorisation instead of starting behind it.
//   slot (q & 1, b, J), J = 0..7: U_bJ (written for J > b);  J = 8: V_b;  J = 9: the right-hand side block z_b (first
//   column).   Blocks in the accumulator-linear form.
using ps::MB_BLOCKS;
constexpr size_t MB_DOUBLES = (size_t)2 * 8 * MB_BLOCKS * 256;
constexpr size_t WT_THIRD = (size_t)2 * NB * NB + MB_DOUBLES;      // scheme 0: a third Wt tile, behind the mailbox (block q in tile q mod 3)
constexpr size_t WT_STRIDE = WT_THIRD + (size_t)NB * NB;           // doubles per matrix: two Wt tiles + the mailbox + the third tile
__host__ __device__ inline size_t mb_slot(int q, int b, int J) { return ((size_t)((q & 1) * 8 + b) * MB_BLOCKS + J) * 256; }

constexpr int DAG_QUEUES = 8;   // one ticket queue per XCD (MI355X: 8 XCDs, each with its own 4 MiB L2)

struct alignas(64) DagCtl {
    unsigned int reserved;
    unsigned int error;
    unsigned int pad[14];                 // pad[0..2]: diagnostics of the first timed-out wait; pad[3]: tasks whose workgroup
                                          // MOVED to another compute unit while they ran (dag_where: the launch is tainted
                                          // and the host evaluates again); pad[4]: those that changed the XCD as well
    struct alignas(64) {
        unsigned int next;                // next ticket of this queue
        unsigned int fill[15];
    } queue[DAG_QUEUES];
};

// queue g holds tasks[first[g] .. first[g+1]) in ticket order (kernel argument, by value)
struct DagQueues {
    unsigned int first[DAG_QUEUES + 1];
    unsigned int follow_first;      // scheme 2: the first block row whose strip solves follow (0, or 2: PSOAP_FOLLOW_ROW0=0)
};
// Ready-only hand-out of the PART tasks (round 5, the review's item 4, asked for since round 3) -- BUILT, MEASURED, NOT
// SHIPPED: compiled in with -DPSOAP_POOL only (tools/build_variant.py pool -DPSOAP_POOL).  The premise was round 3's reading of
// tools/wg_occupancy.py: "80-130 of the 256 workgroups of a single N = 6000 evaluation hold PARTs that wait".  That column
// counts a part from its start to the stamp behind its LAST panel's wait -- the look-ahead K-loop over its older panels
// included.  The stamps that add up the waits themselves (tools/part_wait_share.py, profiles/r5_pool_*.txt) say: in list
// order the parts spend 6.9 % of the time they hold a workgroup waiting for block rows and 5.6 % for their predecessor's
// tile at N = 6000 (2.7 / 2.1 % at N = 8192, 2.9 / 2.4 % for eight matrices) -- at most 8 % of the launch's capacity, on
// a launch whose length is the row-to-row chain's.  Handed out ready-only (three iterations: compare-exchange per final,
// chains overlapping again, fetch-add with held tickets; windows of 128 ... 4096 parts; just-in-time leads 0 ... 16) the
// waits for rows drop to 1.2 % and the finals pay for it: they are drawn later, hold their workgroups for 186 ms in all
// instead of 121 (eight matrices: 4.92 s instead of 3.37) and the row-to-row period grows from 55 to 70 us -- a single
// N = 6000 evaluation takes 3.25 ms against 2.57, eight take 15.0 against 10.9, N = 8192 5.85 against 4.82, predict 11.7
// against 10.5.  The list order with its just-in-time parts IS the better scheduler here; what bounds the single evaluation
// is the chain.
// How it works, for the record.  With ONE in-order ticket list a workgroup that draws a PART whose panels or predecessor are
// not there yet holds it and waits, while ready PARTs further down the list wait for a workgroup.  The list is handed out
// in two parts per queue:
//   main  the finals (DIAG / OFF / SCHUR), in the list's order, from a ticket counter as before -- but a final with a chain
//         is only handed out once the chain's LAST part has been taken (so whoever holds a final waits for running work only);
//   pool  the PARTs, in the list's order, each with a `taken` bit: a workgroup that finds no final to take scans a window
//         of the pool from its first untaken entry and takes a part that is READY -- its panels' block rows complete
//         (rows_done >= pb) and its predecessor in the chain TAKEN (it adds the predecessor's running sum at the end of
//         its own update, so the parts of a chain overlap as they do in list order).
// Every wait still targets a task somebody is running: finals wait for finals with smaller main tickets (all handed out)
// and for their chain (all taken); parts wait for nothing.  And something can always be taken: when nothing runs, either
// the head final's chain is taken (it can be handed out) or the pool's first untaken part is ready (its predecessors are
// done, the rows it reads belong to finals ahead of the head) -- tests/test_dag_plan.py plays it through.
// order[first[g] .. first[g+1]) of queue g: n_main[g] task ids of finals, then the ids of its PARTs; dep[] (main entries):
// position in order[] of the last part of the final's chain, DAG_POOL_NONE without one.
constexpr unsigned int DAG_POOL_NONE = 0xffffffffu;
struct DagPool {
    const unsigned int* order;      // nullptr: the launch hands its tasks out in list order (schemes 0, streams)
    const unsigned int* dep;
    unsigned int* taken;            // one BIT per entry of order[] (bit p & 31 of word p >> 5), zeroed per launch
    unsigned int n_main[DAG_QUEUES];
};

// Scheme 0 (the kernels without the latency paths, LAT = false; round 4): updates wait for the TILES they read -- the
// per-column progress words MatFlags::rvrow -- instead of whole block rows (dag_update).  Compile-time: a run-time switch
// around a one-lane poll is the code shape on which hipcc parks values under the poll's exec mask (the
// build's assembly scan caught exactly that in the first version).  -DPSOAP_NO_TILE_DEPS: whole rows as in rounds 1-3 (A/B).
#ifdef PSOAP_NO_TILE_DEPS
constexpr bool DAG_TILE_DEPS = false;
#else
constexpr bool DAG_TILE_DEPS = true;
#endif

// One entry of the host-built task list (dag_build_tasks); the ticket is the index.
//   PART : partial left-looking update of tile (q, j) over finished block rows [pa, pb); the
//          128 x 128 partial sum goes to workspace slot `slot`, then arrive[ctr] += 1.
//   DIAG / OFF : the final part [pa, pb) of the update, plus the S-1 partials of slots
//          slot .. slot+S-2 (added in slot order once arrive[ctr] == S-1), then the tile's
//          factorisation (DIAG) or strip solve (OFF).
// Splitting along K serves two purposes: the diagonal tile of block row q+1 is pre-accumulated
// over rows < q while block row q is still in flight (its final part is one panel long, so the
// critical chain per block row is 128-row update -> in-block Cholesky), and the last block rows,
// which have too few tiles to occupy the persistent grid, are cut into up to 8 parts per tile.
// Latency scheme only -- the row-to-row critical path potrf(q) -> strip solve of (q, q+1) -> update of
// (q+1, q+1) -> potrf(q+1) is kept inside the DIAG tasks, one cross-workgroup hand-off per block row:
//   DAG_FUSED    (DIAG)  after the in-block Cholesky the same workgroup solves tile (q, q+1) and publishes
//                        next_done = q + 1;
//   DAG_WAITNEXT (DIAG)  the final part [q-1, q) waits for next_done >= q instead of the whole block row;
//   DAG_NOSOLVE  (OFF)   tile (q, q+1): update only, publishes off1_ready = q + 1 (the DIAG task solves it).
//   DAG_SCHUR    (predict) final of a tile of the Schur complement  A - W^T W  = Sigma: rows AND columns lie in the
//                        appended range (q, j >= P), update over all P block rows, then the tile -- with the prior
//                        covariance A evaluated on the fly like K -- is stored into DagAug::S and mirrored; no solve.
enum : unsigned char { DAG_PART = 0, DAG_DIAG = 1, DAG_OFF = 2, DAG_SCHUR = 3, DAG_TYPE_MASK = 0x0F, DAG_CHAIN = 0x10,
                       DAG_NOSOLVE = 0x20, DAG_WAITNEXT = 0x40, DAG_FUSED = 0x80 };
struct DagTask {
    unsigned char type, q, j, S;
    unsigned short b;
    unsigned char pa, pb;
    unsigned int slot;
    unsigned int ctr;
};
static_assert(sizeof(DagTask) == 16, "DagTask is 16 bytes");

constexpr long long DAG_MAX_SPINS = 2000000;  // x (s_sleep + atomic round trip) ~ seconds

#define PSOAP_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Poll of a flag word (poll_word, common.hpp).  Rounds 1-5 wrote this as an atomic add of zero in the belief that an agent-scope
// LOAD could be served stale from the polling XCD's L2 "indefinitely"; hipcc compiled every such add to exactly that load
// (`flat_load_dword ... sc1`) all along, and the litmus tests of round 6 (tools/litmus.py, profiles/r6_litmus.txt) show sc1 loads
// coherent between XCDs -- line planted in the reader's L2 or not, idle or under L2 pressure, 0 stale in 20,000 rounds per
// combination.  -DPSOAP_RMW_POLL makes the polls the returning atomics they were meant to be (1 % slower; not needed).
__device__ __forceinline__ int dag_peek(int* flag) { return poll_word(flag); }

// A tile element other workgroups will read (the two store routines every tile goes through: dag_store_updated, dag_trsm)
// is WRITTEN THROUGH (sc0 sc1) instead of left dirty in the writing XCD's L2 for the release fence to write back: what the
// consumer reads is in memory when the storing wave's vmcnt reaches zero, wherever the workgroup is when its release
// fence runs (round 5; measured performance-neutral in round 4: 848.1 against 847.3 evals/s).  -DPSOAP_NO_WT_STORES: plain.
// WT = false: a plain store -- the tiles of a matrix that ONE workgroup factors by itself (solo_kernel.hpp: nobody else reads
// them inside the launch, and what stays in the L2 is what that workgroup reads next).
template <bool WT = true>
__device__ __forceinline__ void dag_st(double* p, double v)
{
#ifdef PSOAP_NO_WT_STORES
    *p = v;
#else
    if constexpr (WT)
        __hip_atomic_store((__attribute__((address_space(1))) double*)(uintptr_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else
        *p = v;
#endif
}

// Where this workgroup runs: XCC_ID[3:0] and HW_ID[15:8] (compute unit, shader array, shader engine).  A workgroup normally
// stays where it was started.  When several processes share the device its scheduler suspends running workgroups (compute
// wave save / restore) and may resume them on ANOTHER compute unit -- whose vector L1 was not invalidated by the acquire
// the workgroup ran before it was suspended, so a tile that was rewritten since that compute unit last read it (the
// updated tile -> the solved tile, the ping-pong slots of a chain, the W tiles) can come back stale: the silent wrong
// lnprob of DESIGN.md 5 (1e-4 per evaluation with 12-16 worker processes).  Every task samples its place when it starts
// and again before it retires; a task that MOVED marks the launch (DagCtl::pad[3]) -- and its lane in a stream -- as
// tainted, and the host evaluates again (psoap_gp.hip: share_*).  One s_getreg pair per task.
__device__ __forceinline__ unsigned int dag_where()
{
    const unsigned int hw = __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11));      // HW_REG_HW_ID[15:8]: CU_ID, SH_ID, SE_ID
    const unsigned int xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));               // HW_REG_XCC_ID[3:0]
    return hw | (xcc << 8);
}

// thread 0, behind a task's last store and before it retires: did the workgroup move since the task started?
// (lane_taint: the word of the task's lane in a stream, nullptr otherwise)
__device__ __forceinline__ void dag_moved_check(unsigned int w0, DagCtl* ctl, unsigned int* lane_taint)
{
#ifdef PSOAP_NO_MOVE_CHECK            // (A/B builds only: what the check costs)
    (void)w0; (void)ctl; (void)lane_taint;
    return;
#endif
    const unsigned int w1 = dag_where();
    if (w1 != w0) {
        __hip_atomic_fetch_add(&ctl->pad[3], 1u, PSOAP_RLX_AGENT);
        if ((w1 ^ w0) >> 8) __hip_atomic_fetch_add(&ctl->pad[4], 1u, PSOAP_RLX_AGENT);      // ... to another XCD (diagnostics)
        if (lane_taint) __hip_atomic_store(lane_taint, 1u, PSOAP_RLX_AGENT);
    }
}

// consumer side: one lane polls, one acquire, drain, barrier
__device__ __forceinline__ void dag_wait_ge(int* flag, int target, DagCtl* ctl, unsigned int code = 0)
{
    if (threadIdx.x == 0) {
        long long spins = 0;
        while (dag_peek(flag) < target) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > DAG_MAX_SPINS) {
                if (__hip_atomic_fetch_or(&ctl->error, 1u, PSOAP_RLX_AGENT) == 0u) {
                    // first failure: what was waited for (diagnostics only)
                    __hip_atomic_store(&ctl->pad[0], code, PSOAP_RLX_AGENT);
                    __hip_atomic_store(&ctl->pad[1], (unsigned int)target, PSOAP_RLX_AGENT);
                    __hip_atomic_store(&ctl->pad[2], (unsigned int)dag_peek(flag), PSOAP_RLX_AGENT);
                }
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// the same for two words (the two column tiles an update reads): both >= target
__device__ __forceinline__ void dag_wait_ge2(int* fa, int* fb, int target, DagCtl* ctl, unsigned int code = 0)
{
    if (threadIdx.x == 0) {
        long long spins = 0;
        for (;;) {
            int v = dag_peek(fa);
            if (v >= target && fb != fa) v = dag_peek(fb);
            if (v >= target) break;
            __builtin_amdgcn_s_sleep(32);
            if (++spins > DAG_MAX_SPINS) {
                if (__hip_atomic_fetch_or(&ctl->error, 1u, PSOAP_RLX_AGENT) == 0u) {
                    __hip_atomic_store(&ctl->pad[0], code, PSOAP_RLX_AGENT);
                    __hip_atomic_store(&ctl->pad[1], (unsigned int)target, PSOAP_RLX_AGENT);
                    __hip_atomic_store(&ctl->pad[2], (unsigned int)v, PSOAP_RLX_AGENT);
                }
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
}

// -DPSOAP_CHAOS (a test build, tools/build_variant.py chaos -DPSOAP_CHAOS): one task in sixteen -- chosen by a hash of its
// ticket, its matrix's submission number and the place -- sleeps ~100 us at its start, or between its last output and its
// publications.  A protocol whose every dependency is an explicit hand-off returns the same bits whatever a task's timing;
// one that leans on "that task is always through by then" does not: round 6 found two such places in the following scheme
// (the accumulator chain, the in-order progress words) through the rare wrong values they produced, and this is the
// instrument that shows there is no third (tools/soak_stream.py under the chaos build; -DPSOAP_NO_INORDER brings the
// second one back, to show that the instrument sees it).
__device__ __forceinline__ void dag_chaos(unsigned int key, unsigned int place)
{
#ifdef PSOAP_CHAOS
    unsigned int h = (key ^ (place * 0x9E3779B9u)) * 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    if ((h & 15u) == 0u)
        for (int i = 0; i < 30; ++i) __builtin_amdgcn_s_sleep(127);
#else
    (void)key; (void)place;
#endif
}

// thread 0 only, no fence, no barrier: until a progress word has reached `target` -- the IN-ORDER publication of the words
// whose readers take "value >= q + 1" to mean "everything up to q": potrf_done and rows_done.  Their producers -- the
// diagonal tasks, the last finishers of the block rows -- normally end in order; in the following scheme (second level)
// only TIME said so (diagonal task q + 1 still has a block to factor when task q writes its outputs), and a producer that
// is late by a factorisation's length -- a preempted workgroup on a shared device -- let its successor's larger value
// satisfy waits for its own unfinished outputs (round 6; the accumulator records of common.hpp close the silent half of
// that hole, this closes the rest).  The predecessor holds a smaller ticket: it runs or is through.
__device__ __forceinline__ void dag_spin_ge(int* flag, int target, DagCtl* ctl, unsigned int code)
{
#ifdef PSOAP_NO_INORDER            // (A/B of the chaos build only: rounds 3-5's publication, ordered by time alone)
    (void)flag; (void)target; (void)ctl; (void)code;
    return;
#endif
    long long spins = 0;
    while (dag_peek(flag) < target) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > DAG_MAX_SPINS) {
            if (__hip_atomic_fetch_or(&ctl->error, 1u, PSOAP_RLX_AGENT) == 0u) {
                __hip_atomic_store(&ctl->pad[0], code, PSOAP_RLX_AGENT);
                __hip_atomic_store(&ctl->pad[1], (unsigned int)target, PSOAP_RLX_AGENT);
                __hip_atomic_store(&ctl->pad[2], (unsigned int)dag_peek(flag), PSOAP_RLX_AGENT);
            }
            break;
        }
    }
}

// producer side, part 1 (all threads): drain own stores, meet at the barrier
__device__ __forceinline__ void dag_drain()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// producer side, part 2 (thread 0 only): release, then signal
__device__ __forceinline__ void dag_release_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// a task of block row q of this matrix is complete (thread 0, after dag_release_fence)
// (M3: three counters in turn -- scheme 0, whose diagonal tasks run up to two block rows ahead of the strip solves)
template <bool M3 = false>
__device__ __forceinline__ void dag_task_done(MatFlags* f, int q, int ntasks_row, int n = 1, DagCtl* ctl = nullptr)
{
    int* cnt = M3 ? (q % 3 == 2 ? &f->cnt3 : &f->cnt[q % 3]) : &f->cnt[q & 1];
    const int old = __hip_atomic_fetch_add(cnt, n, PSOAP_RLX_AGENT);
    if (old + n == ntasks_row) {
        // last finisher of the row: order after every other task's release, then publish the row
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the reset must be visible before any task of the next row -- released by rows_done -- adds to
        // cnt: drain the reset, then publish the row with a release store (two relaxed stores are unordered)
        __hip_atomic_store(cnt, 0, PSOAP_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (rows are published in order: "rows_done >= q + 1" has to mean rows 0 .. q -- dag_spin_ge)
        if (ctl && q > 0) dag_spin_ge(&f->rows_done, q, ctl, 11u);
        __hip_atomic_store(&f->rows_done, q + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// left-looking update over finished block rows [pa, pb) with look-ahead: all but the last panel
// need rows_done >= pb-1, the last one rows_done >= pb
// (wa, wb -- tile-level dependencies, scheme 0 since round 4: the progress words of the two column tiles the update reads
// (MatFlags::rvrow of block columns k0 / 128 and j0 / 128: r + 1 once tile (r, column) is final) instead of the count of
// completed block rows.  A task then never waits for the stragglers of the row above, only for its own two operands --
// with whole rows, every task of a row taken soon after the row above stalls until that row's last task is through: 7.5 %
// of all worker time in a streamed run, where few matrices share a queue and the rows follow each other closely.)
template <bool SW = false, class SM = SmemKernel, bool ROWMAP = false, bool TD = false>
__device__ __forceinline__ void dag_update(Tile& t, double* Km, int ld, int k0, int j0, int pa, int pb, MatFlags* f,
                                           DagCtl* ctl, bool wait_next, unsigned long long* tl, int wave_s = -1, SM sm = SM(),
                                           int* wa = nullptr, int* wb = nullptr)
{
    if (pb <= pa) return;
    const bool diag = (k0 == j0);
    // (task log: the time spent in the two waits below is added up in bits 40.. of word 7 -- tools/stream_timeline.py)
    unsigned long long w0 = 0;
    if (tl && threadIdx.x == 0) w0 = __builtin_amdgcn_s_memrealtime();
    if (pb - pa > 1) {
        if constexpr (TD) dag_wait_ge2(wa, wb, pb - 1, ctl, 1u);
        else dag_wait_ge(&f->rows_done, pb - 1, ctl, 1u);
        if (tl && threadIdx.x == 0) {
            const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
            tl[7] += (w1 - w0) << 40;
        }
        const size_t r0 = (size_t)pa * NB;
        tile_gemm_tn<SW, SM, ROWMAP>(t, Km + r0 * ld + k0, (size_t)ld, Km + r0 * ld + j0, (size_t)ld, (pb - 1 - pa) * NB, diag, 0x7fffffff, wave_s, sm);
    }
    // the last panel: the whole block row above, or (diagonal tile of the latency scheme) only its tile
    // right of the diagonal -- U(pb-1, pb), all this tile reads of that row
    if (tl && threadIdx.x == 0) w0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (TD) dag_wait_ge2(wa, wb, pb, ctl, 2u);
    else dag_wait_ge(wait_next ? &f->next_done : &f->rows_done, pb, ctl, 2u);
    if (tl && threadIdx.x == 0) {
        const unsigned long long w1 = __builtin_amdgcn_s_memrealtime();
        tl[4] = w1;
        tl[7] += (w1 - w0) << 40;
    }
    const size_t r1 = (size_t)(pb - 1) * NB;
    tile_gemm_tn<SW, SM, ROWMAP>(t, Km + r1 * ld + k0, (size_t)ld, Km + r1 * ld + j0, (size_t)ld, NB, diag, 0x7fffffff, wave_s, sm);
}

// The last panel of a following strip solve's update, K = 128: its operands -- tiles (q-1, q) and (q-1, j) -- are being
// solved by tasks that follow the factorisation of block q-1 and deliver their row blocks one by one (dag_pss, xpub;
// MatFlags::xcol): stage ch is row block ch of both.  A stage is requested ahead of the products of the one before it
// when it is known to be there, behind them otherwise (every wave stages its own rows and polls for itself).
template <class SM>
__device__ __forceinline__ void dag_update_following(Tile& t, const double* __restrict__ A, const double* __restrict__ B,
                                                     size_t ld, int* xa, int* xb, int base, unsigned int* err, int wave, SM sm,
                                                     unsigned long long* tl = nullptr)
{
    const int wr = wave >> 1, wc = wave & 1;
    const int lane = hw_lane();
    int avail = ps::x_wait(xa, xb, base, err, 1, lane);
    if (tl && threadIdx.x == 0) tl[4] = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    stage_glds_w(A, ld, B, ld, 0, 0, wave, sm);
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < NB / KB; ++ch) {
        const int cur = ch & 1;
        bool early = false;
        if (ch + 1 < NB / KB) {
            if (avail < ch + 2) avail = ps::x_steps(xa, xb, base, lane);
            if (avail >= ch + 2) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                stage_glds_w(A, ld, B, ld, (ch + 1) * KB, cur ^ 1, wave, sm);
                early = true;
            }
        }
        tile_mma_chunk<false, SM>(t, cur, wr, wc, 0, 0, sm);
        if (ch + 1 < NB / KB && !early) {
            avail = ps::x_wait(xa, xb, base, err, ch + 2, lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            stage_glds_w(A, ld, B, ld, (ch + 1) * KB, cur ^ 1, wave, sm);
        }
        __syncthreads();
    }
}

// The ONE consumer of the accumulators.  Every task ends its update here:
//     dest(i, j) <- scale * K(i, j) - acc(i, j)
//   final tasks : dest = the tile (k0, j0) of the matrix, scale = 1 (0 in a chain, whose first PART
//                 carries K instead);
//   PART tasks  : dest = a workspace slot (row-major 128 x 128), scale = 0.
// The partial tiles left by a tile's PART tasks (all S-1 of them, or the running sum of a chain) have
// been folded into the accumulators before (dag_sub_partials).
// K is evaluated on the fly in the MFMA accumulator layout (same arithmetic as k_fill_sym:
// squared-exponential sum, diagonal rule, sigma^2 on the diagonal, identity padding), so the
// covariance matrix is never materialised in HBM: each tile is written once, already updated.
// (A separate store routine for PART tasks makes hipcc spill the 128 accumulator registers inside the
// MFMA loops -- a second consumer -- so PART tasks share this one and skip the exp() evaluation through
// a wave-uniform branch per element: +0.7 % end to end, the fp64 pipe time goes back to the MFMAs.)
// Augmented columns (predict: [B | Cx^T]): column tiles j >= P carry cross-covariances between the data
// grid (rows) and a prediction grid (columns).  colx holds the prediction abscissae per component,
// (C, Rpad); a component that must not contribute to a column is given the abscissa 1e30 there (its
// exponent underflows to an exact +0).  Columns >= R are padding (zeros).
struct DagAug {
    int Pt;               // column tiles in total (P + extra); == P when there are none
    int R, Rpad;          // valid / padded extra columns
    const double* colx;   // (C, Rpad)
    // Schur-complement tiles (DAG_SCHUR; predict's Sigma = A - W^T W computed inside the same launch):
    const double* rowx;   // (C, Rpad) the same abscissae for tile ROWS, with -1e30 where colx has +1e30 (a component that
                          // contributes to neither the row's nor the column's block must see r = 2e30, not r = 0)
    const double* diag;   // (Rpad) prior variances: the diagonal of A (amp^2 sums, + the 1e-8 nugget of predict_f_g_sum)
    double* S;            // (Rpad x lds) Sigma
    size_t lds;
};

// One matrix of the batch.  The batch may be heterogeneous (matrices of several chunks with their own
// size, data-noise vector and storage); a task reads its matrix's record with scalar loads.
struct DagMat {
    double* K;            // (Npad x ld) row-major upper storage, overwritten by the factor
    double* R;            // (Npad) r -> z
    double* Wt;           // 2 x (128 x 128): U11^-T of diagonal block q, k-major, in buffer q & 1 (DIAG(q+1) may
                          // factor while the strip solves of row q still read theirs)
    const double* lw;     // (C, N) ln-wavelengths per component
    const double* gp;     // (2C) amp, l per component
    const double* sigma;  // (N)
    MatAcc* acc;
    int N, Npad, P, ld;
};

// INPLACE: the result replaces the accumulators instead of going to memory (the strip solve that follows works on the
// tile in registers: dag_pss)
#ifdef PSOAP_FAST_STREAM
#define DAG_FAST_STORE(stream) true
#else
#define DAG_FAST_STORE(stream) (!(stream))
#endif
// FAST: a task that adds no covariance (every PART but a chain's first, and the final of a chain) stores its tile through a
// loop of its own (see there).
template <int C, bool AUG, bool INPLACE = false, bool ROWMAP = false, bool FAST = false, bool WT = true>
__device__ __forceinline__ void dag_store_updated(Tile& t, double* __restrict__ dest, size_t ldd, int k0, int j0,
                                                  const double* __restrict__ lw, const GpDev& g, double dsum,
                                                  const double* __restrict__ sigma, int N, double scale,
                                                  int Npad, const DagAug& aug, double* __restrict__ mirror = nullptr,
                                                  bool plain = false)
{
    static_assert(INPLACE || !ROWMAP, "tiles that go to memory use the plain accumulator map");
    if constexpr (FAST && !INPLACE) {
        if (plain && scale == 0.0) {        // (plain: a workspace slot or a tile of the matrix -- no mirror image)
            // 0 * K - acc, and K >= 0 is finite, so 0.0 - acc is the same value bit for bit: 64 stores back to back, four
            // instructions each.  (In the general routine below such a task jumps over the covariance blocks and still
            // executes ~40 vector instructions per element -- index tests, selects, 64-bit address arithmetic -- each of
            // which, beside a neighbour that streams MFMAs, waits for a gap in that stream: 33 us for its 64 stores with no
            // memory wait between them, 4.5 us here -- tools/predict_timeline.py.  Not instruction fetch: SQC_ICACHE_MISSES
            // is 0.02 % of the requests.)
            int tid_ = threadIdx.x;
            asm volatile("" : "+v"(tid_));
            const int lane = tid_ & 63, wave = tid_ >> 6;
            const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dag_st<WT>(&dest[(size_t)tile_row(wr, m, lane, r) * ldd + (size_t)tile_col(wc, n, lane)], 0.0 - t.acc[m][n][r]);
            return;
        }
    }
    // the thread id passes through an opaque statement: everything below (coordinate loads, addresses)
    // depends on it and so cannot be hoisted above the K-loops of the update, where it would sit in
    // registers the MFMA loop then has to spill around
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63, wave = tid_ >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    double xj[4][C];
    int jj[4];
    const bool cross = AUG && j0 >= Npad;      // wave-uniform: a tile of the appended columns
    const bool rowcross = AUG && k0 >= Npad;   // wave-uniform: a tile of the Schur complement (rows appended too)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        jj[n] = j0 + tile_col(wc, n, lane);
        if (cross) {
            // map to a column index that passes / fails the `j < N` test below and never equals a data row index
            const int e = jj[n] - Npad;
#pragma unroll
            for (int c = 0; c < C; ++c) xj[n][c] = (e < aug.R) ? aug.colx[(size_t)c * aug.Rpad + e] : 0.0;
            jj[n] = (e < aug.R) ? -1 - e : 0x7fffffff;
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) xj[n][c] = (jj[n] < N) ? lw[(size_t)c * N + jj[n]] : 0.0;
        }
    }
    if constexpr (INPLACE) {
        // (nothing is stored here: the load of a diagonal element's sigma_i inside the element loop delays nobody, and this
        // shape -- rounds 2-5 -- keeps the chain phases of dag_special in registers)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            double xi[4][C];
            int ii[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ii[r] = k0 + (ROWMAP ? 16 * tile_rowblock(wr, m) + (lane >> 4) + 4 * r : tile_row(wr, m, lane, r));
                if (rowcross) {
                    const int e = ii[r] - Npad;        // the same index map as the columns: equal indices = the diagonal of A
#pragma unroll
                    for (int c = 0; c < C; ++c) xi[r][c] = (e < aug.R) ? aug.rowx[(size_t)c * aug.Rpad + e] : 0.0;
                    ii[r] = (e < aug.R) ? -1 - e : 0x7ffffffe;
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) xi[r][c] = (ii[r] < N) ? lw[(size_t)c * N + ii[r]] : 0.0;
                }
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                double kv[4] = {0.0, 0.0, 0.0, 0.0};
                if (scale != 0.0) kern_elem4_skip<C>(xi, xj[n], g, kv);   // wave-uniform: only one task per tile adds K
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = ii[r], j = jj[n];
                    double v;
                    if (i < N && j < N) {
                        if (i == j) {
#pragma clang fp contract(off)
                            if (rowcross) {
                                v = aug.diag[-1 - i];
                            } else {
                                const double sg = sigma[i];
                                v = dsum + sg * sg;
                            }
                        } else {
                            v = kv[r];
                        }
                    } else {
                        v = (i == j) ? 1.0 : 0.0;
                    }
                    t.acc[m][n][r] = scale * v - t.acc[m][n][r];
                }
            }
        }
        return;
    }
    // The element loop contains no load: the rows' coordinates are fetched per 16-row block m ahead of its 16 element
    // groups, and the DIAGONAL elements -- the only ones that need another load (sigma_i, or the prior variance of a Schur
    // row) -- are written a second time, with their values, by the pass below.  (Rounds 2-5 fetched sigma[i] / diag[e]
    // inside an `i == j` branch of this loop; hipcc joins such a branch with `s_waitcnt vmcnt(0)`, and on gfx9 that counter
    // counts STORES too: every one of a tile's 64 stores waited for the one before it to reach memory.)  An element of the
    // diagonal can only sit in the 16 x 16 block n == m of a wave with wr == wc (row block 4 wr + m against column block
    // 4 wc + n) of a tile with k0 == j0.
    const auto row_index = [&](int m, int r) {
        const int i = k0 + tile_row(wr, m, lane, r);
        if (!rowcross) return i;
        const int e = i - Npad;            // the same index map as the columns: equal indices = the diagonal of A
        return (e < aug.R) ? -1 - e : 0x7ffffffe;
    };
    const auto put = [&](int m, int n, int r, double out) {
        const int jc = tile_col(wc, n, lane), ic = tile_row(wr, m, lane, r);      // column and row inside the tile
        if (AUG && mirror) {
            // a tile of Sigma, written twice: as it is and transposed.  A DIAGONAL tile (mirror == dest) comes out of
            // the update with its lower-left quadrant missing (the symmetric update skips it): only the elements
            // on and above the diagonal are stored, each also at its mirror position.
            if (mirror != dest || ic <= jc) {
                dest[(size_t)ic * ldd + (size_t)jc] = out;
                mirror[(size_t)jc * ldd + (size_t)ic] = out;
            }
            return;
        }
        dag_st<WT>(&dest[(size_t)ic * ldd + (size_t)jc], out);
    };
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        double xi[4][C];
        int ii[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ii[r] = row_index(m, r);
            if (rowcross) {
                const int e = -1 - ii[r];
#pragma unroll
                for (int c = 0; c < C; ++c) xi[r][c] = (ii[r] < 0) ? aug.rowx[(size_t)c * aug.Rpad + e] : 0.0;
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) xi[r][c] = (ii[r] < N) ? lw[(size_t)c * N + ii[r]] : 0.0;
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            double kv[4] = {0.0, 0.0, 0.0, 0.0};
            if (scale != 0.0) kern_elem4_skip<C>(xi, xj[n], g, kv);   // wave-uniform: only one task per tile adds K
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ii[r], j = jj[n];
                const double v = (i < N && j < N) ? kv[r] : 0.0;
                // (an element of the diagonal is written here like its neighbours and again, with its value, by the pass below:
                // two stores of one lane to one address arrive in program order, and no exec-masked store sits in this loop)
                put(m, n, r, scale * v - t.acc[m][n][r]);
            }
        }
    }
    // the diagonal of the matrix: sigma_i^2 + sum of amp^2 (the prior variance for a row of the Schur complement), 1 in
    // the identity padding -- the 16 loads first, then the stores
    if (k0 == j0 && wr == wc) {
#ifdef PSOAP_DIAG_PASS_WAIT
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the loop's stores are in memory before the pass stores
#endif
        double dg[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = row_index(m, r);
                if (rowcross) {
                    dg[m][r] = (i < 0) ? aug.diag[-1 - i] : 1.0;      // (rows and columns beyond R never compare equal)
                } else {
                    const double sg = (i < N) ? sigma[i] : 0.0;
                    {
#pragma clang fp contract(off)
                        dg[m][r] = (i < N) ? dsum + sg * sg : 1.0;
                    }
                }
            }
        // (the 16 values before the first conditional store: one wait for the loads, outside the branches -- hipcc otherwise
        // waits inside every one of them, and each such wait covers the store of the branch before)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) dg[m][r] = scale * dg[m][r] - t.acc[m][m][r];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (row_index(m, r) == jj[m]) put(m, m, r, dg[m][r]);
    }
    (void)plain;
}

// the accumulators as a tile in memory (after dag_store_updated<.., INPLACE>): the same element map and mirror rule
template <bool AUG>
__device__ __forceinline__ void dag_store_tile(const Tile& t, double* __restrict__ dest, size_t ldd, double* __restrict__ mirror)
{
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63, wave = tid_ >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ic = tile_row(wr, m, lane, r), jc = tile_col(wc, n, lane);
                const double out = t.acc[m][n][r];
                if (AUG && mirror) {
                    if (mirror != dest || ic <= jc) {
                        dest[(size_t)ic * ldd + (size_t)jc] = out;
                        mirror[(size_t)jc * ldd + (size_t)ic] = out;
                    }
                } else {
                    dest[(size_t)ic * ldd + (size_t)jc] = out;
                }
            }
}

// accumulators -= partial tiles prev[0 .. n_prev), in slot order (row-major 128 x 128 workspace slots).
// All 64 loads of a slot are in flight together; folding the partial sums into the accumulators keeps
// the store routine free of them (a load per element between its stores serialised on the memory
// latency: 64 round trips, 44 us per tile).  A chain's final calls this BEFORE it waits for the block row
// above -- its PARTs ran ahead -- so the running sum is read off the row-to-row path.
// BATCH: through tile_load16 (gemm_core.hpp) -- four round trips to memory per tile where hipcc schedules this loop as 64
// (behind an update, and in dag_special: 25-29 us per partial tile, tools/predict_timeline.py).
template <bool ROWMAP = false, bool BATCH = false>
__device__ __forceinline__ void dag_sub_partials(Tile& t, const double* __restrict__ prev, int n_prev)
{
    // (opaque copy of the thread id, as in dag_store_updated: otherwise the 64 per-lane element offsets below are
    // computed once at the top of the kernel, kept for its whole life and spilled)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63, wave = tid_ >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    for (int sidx = 0; sidx < n_prev; ++sidx) {
        const double* __restrict__ ps = prev + (size_t)sidx * NB * NB;
#ifndef PSOAP_NO_BATCH_FOLD
        if constexpr (BATCH && !ROWMAP) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                double v[4][4];          // [n][r]
                tile_load16(ps + (size_t)tile_row(wr, m, lane, 0) * NB + (size_t)tile_col(wc, 0, lane), (size_t)4 * NB, v);
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) t.acc[m][n][r] -= v[n][r];
            }
            continue;
        }
#endif
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    t.acc[m][n][r] -= ps[(size_t)(ROWMAP ? 16 * tile_rowblock(wr, m) + (lane >> 4) + 4 * r : tile_row(wr, m, lane, r)) * NB +
                                         (size_t)tile_col(wc, n, lane)];
    }
}

// strip solve + right-hand-side update for tile (k0, j0); the updated tile is already in memory
// BAL: the row blocks are dealt to the waves by work (tile_gemm_tn_lower_balanced and its accumulator map)
// VP: how the two 128-entry LDS vectors are addressed (double* inside a kernel, lds_double* inside dag_diag_fast);
// SM: how the tile engine's LDS array is reached (gemm_core.hpp)
template <bool BAL = false, class VP = double*, class SM = SmemKernel, bool WT = true>
__device__ __forceinline__ void dag_trsm(Tile& t, double* Km, int ld, int k0, int j0, const double* Wm, double* Rv,
                                         int Npad, VP zk, VP colsum, SM sm = SM())
{
    const int tid = threadIdx.x;
    if (tid < NB) zk[tid] = Rv[k0 + tid];
    t.zero();
    if (BAL) tile_gemm_tn_lower_balanced(t, Wm, (size_t)NB, Km + (size_t)k0 * ld + j0, (size_t)ld, sm);
    else tile_gemm_tn_lower(t, Wm, (size_t)NB, Km + (size_t)k0 * ld + j0, (size_t)ld);
    int tid_ = threadIdx.x;                      // opaque: the store offsets below are not kernel-lifetime values
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63, wave = tid_ >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = BAL ? 16 * lower_rowblock(wr, m) + (lane >> 4) + 4 * r : tile_row(wr, m, lane, r);
            const double z = zk[row];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const double x = t.acc[m][n][r];
                dag_st<WT>(&Km[(size_t)(k0 + row) * ld + j0 + tile_col(wc, n, lane)], x);
                part[n] = fma(x, z, part[n]);
            }
        }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        part[n] += __shfl_xor(part[n], 16, 64);
        part[n] += __shfl_xor(part[n], 32, 64);
    }
    if (wr == 1 && lane < 16) {
#pragma unroll
        for (int n = 0; n < 4; ++n) colsum[tile_col(wc, n, lane)] = part[n];
    }
    __syncthreads();
    if (wr == 0 && lane < 16 && j0 < Npad) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = tile_col(wc, n, lane);
            Rv[j0 + col] -= part[n] + colsum[col];
        }
    }
}

// Diagonal tile of the latency scheme, the row-to-row critical path: the running sum of its PART chain,
// the last K = 128 symmetric update and the in-block Cholesky all happen in registers (potrf_spine_fused)
// -- no tile engine, no store / drain / reload of the tile in between -- and, when the task is
// DAG_FUSED, the strip solve of the tile right of the diagonal follows in the same workgroup.
// A function of its own (not inlined): inside the kernel body a second instance of the factorisation makes
// hipcc spill in the MFMA loops of every other task (32-walker batch: 39.5 -> 44.2 ms).
// Being a non-kernel function it names NO __shared__ variable: the kernel hands it the base of the dynamic LDS
// array (`smem`) and its two LDS vectors as address_space(3) pointers, and the spine's progress flag lives in
// that array (pb::OFF_FLAG) -- see gemm_core.hpp (SmemArg) and DESIGN.md 3.4 for why.
//   -DPSOAP_DIAG_INLINE     compiles the routine into the LAT kernels instead (2-5 % slower in the latency regime; the
//                           build's fallback rung, psoap_amd/build.py).
// (The code-shape variants of the round-3 compiler-defect study -- poll in front of the call, table-lookup callee, round-2
// staging forms, s_nop padding, ... -- are no longer in this file: tools/lat_variants.py applies them as a patch,
// tools/lat_variants.patch, to a scratch copy of the sources.)
// The LDS address of a __shared__ object as a value the optimiser cannot see through: with the address visible at
// the (only) call sites, interprocedural constant propagation puts the object's name right back into the callee.
__device__ __forceinline__ lds_double* dag_opaque_lds(double* shared_obj)
{
    unsigned int a = (unsigned int)(uintptr_t)(lds_double*)shared_obj;
    asm volatile("" : "+s"(a));
    return (lds_double*)(uintptr_t)a;
}
#if defined(PSOAP_DIAG_INLINE) || defined(PSOAP_FOLLOW)
#define PSOAP_DIAG_FN __device__ __forceinline__
#else
#define PSOAP_DIAG_FN __device__ __attribute__((noinline))
#endif
#if defined(PSOAP_DIAG_INLINE) && !defined(PSOAP_FOLLOW)
#define PSOAP_DIAG_SMEM(smem) SmemKernel()
#else
#define PSOAP_DIAG_SMEM(smem) SmemArg{smem}
#endif
PSOAP_DIAG_FN void dag_diag_fast(double* Km, int ld, int k0, double* Wm, double* Rv, MatAcc* acc,
                                                        const double* prev, int Npad, MatFlags* f, DagCtl* ctl, int q,
                                                        int ntasks_row, bool fused, int* chain_ctr, int chain_len,
                                                        lds_double* smem, lds_double* zk, lds_double* colsum,
                                                        unsigned long long* tl, bool xfollow = false, bool two_panels = false,
                                                        unsigned int chaos = 0u)
{
    const auto sm = PSOAP_DIAG_SMEM(smem);
    // the wait for the tile's PART chain (it ran ahead: normally no wait at all) happens in here, not in front of the
    // call: a one-lane poll right before a call is where hipcc's register allocator parked the values that live
    // across the call UNDER THE POLL'S EXEC MASK (tools/check_exec_restore.py)
    dag_wait_ge(chain_ctr, chain_len, ctl, 4u);
    __builtin_amdgcn_s_setprio(3);
    // (xfollow: no wait for the finished tile above -- its row blocks are awaited one by one inside the update)
    // (two_panels: the final also applies tile (q-2, q), final with the whole of block row q-2)
    auto wait_dep = [f, ctl, q, tl, xfollow, two_panels]() {
        if (!xfollow) dag_wait_ge(&f->next_done, q, ctl, 2u);
        else if (two_panels) dag_wait_ge(&f->rows_done, q - 1, ctl, 1u);
        if (tl && threadIdx.x == 0) tl[4] = __builtin_amdgcn_s_memrealtime();
    };
    // (the mailbox sits behind the matrix's two Wt tiles: Wm is tile q & 1 of them)
    double* wt0 = Wm - (size_t)(q & 1) * NB * NB;
#ifndef PSOAP_FOLLOW
    potrf_spine_fused(Km, ld, k0, Wm, Rv, acc, prev, Km + (size_t)(k0 - NB) * ld + k0, wait_dep, tl, sm);
    (void)wt0;
#else
    // (block 0 has no tile above it: strip == nullptr, no update)
    potrf_spine_fused(Km, ld, k0, Wm, Rv, acc, prev, q > 0 ? Km + (size_t)(k0 - NB) * ld + k0 : nullptr, wait_dep, tl, sm,
                      ps::SpinePub{wt0 + 2 * NB * NB + mb_slot(q, 0, 0), f->step_w, 8 * q},
                      xfollow ? ps::SpineFollow{wt0 + 2 * NB * NB + mb_slot(q - 1, 0, 0), f->xcol[q], 8 * (q - 1), &ctl->error,
                                                two_panels ? Km + (size_t)(k0 - 2 * NB) * ld + k0 : nullptr}
                              : ps::SpineFollow{nullptr, nullptr, 0, nullptr, nullptr});
#endif
    dag_chaos(chaos, 1u);        // (the tail of a diagonal task: its outputs are issued, nothing is published yet)
    dag_drain();
    if (tl && threadIdx.x == 0) tl[2] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        dag_release_fence();
        if (q > 0) dag_spin_ge(&f->potrf_done, q, ctl, 12u);        // (in order: dag_spin_ge)
        __hip_atomic_store(&f->potrf_done, q + 1, PSOAP_RLX_AGENT);
    }
    if (fused) {
        Tile t;
        dag_wait_ge(&f->off1_ready, q + 1, ctl, 5u);
        dag_trsm<true>(t, Km, ld, k0, k0 + NB, Wm, Rv, Npad, zk, colsum, sm);
        dag_drain();
        if (threadIdx.x == 0) {
            dag_release_fence();
            __hip_atomic_store(&f->next_done, q + 1, PSOAP_RLX_AGENT);
            dag_task_done(f, q, ntasks_row, 2, ctl);
        }
    } else if (threadIdx.x == 0) {
        dag_task_done(f, q, ntasks_row, 1, ctl);
    }
    __builtin_amdgcn_s_setprio(0);
    if (tl && threadIdx.x == 0) tl[3] = __builtin_amdgcn_s_memrealtime();
}

// ---- Progressive strip solve ("following", scheme 2; round 3).  Measured against scheme 1 (profiles/r3_follow_table.txt):
// 6-10 % faster for single evaluations and batches of up to four matrices at N <= 4096, a tie at N = 6000, 3-5 % slower
// beyond (a late task runs through the published steps at about the cost of the product with the explicit inverse,
// and its row blocks are not balanced over the waves) -- dag_auto_scheme picks it accordingly.  The whole task lives in
// the out-of-line routine dag_special: with any part of it in the kernel body, hipcc's code for every other task of the
// latency-scheme kernels got 12-24 % slower.
// Tile (q, j), j > q, stays in the accumulators after its update and is solved
// block row by block row BEHIND the fused diagonal task that is factoring block q, instead of after it:
//     step b:  X_b  = W_bb T_b                 (V_b = W_bb^T from the mailbox; the waves that own row block b)
//              T_I -= U_bI^T X_b,  I > b        (U_bI from the mailbox, X_b through LDS; every wave, its row blocks)
// -- the recurrence the in-block factorisation applies to its own right-hand-side column, carried on for the 8 column
// blocks of this tile.  Same MFMA count as the product with the explicit inverse (1152 per tile), but no W operand to
// stage, no store / drain / reload of the tile between update and solve, and the last row block is final one step
// after the factorisation's last step.  A task that arrives late simply runs through the published steps.
// All 256 threads; LDS: two row buffers of 8 blocks (32 KB) at the start of the dynamic array + one int behind them.
template <class BoxPtr>
__device__ __forceinline__ int dag_wait_steps(MatFlags* f, int target, DagCtl* ctl, BoxPtr box)
{
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {            // wave 0: lanes 0..3 poll one wave's flag each
        int v = 0;
        long long spins = 0;
        for (;;) {
            int x = 0x7fffffff;
            if (lane < 4) x = dag_peek(&f->step_w[lane]);
            x = min(x, __shfl_xor(x, 1, 64));
            x = min(x, __shfl_xor(x, 2, 64));
            v = __builtin_amdgcn_readfirstlane(x);
            if (v >= target) break;
            __builtin_amdgcn_s_sleep(4);
            if (++spins > DAG_MAX_SPINS) {
                if (lane == 0 && __hip_atomic_fetch_or(&ctl->error, 1u, PSOAP_RLX_AGENT) == 0u) {
                    __hip_atomic_store(&ctl->pad[0], 6u, PSOAP_RLX_AGENT);
                    __hip_atomic_store(&ctl->pad[1], (unsigned int)target, PSOAP_RLX_AGENT);
                    __hip_atomic_store(&ctl->pad[2], (unsigned int)v, PSOAP_RLX_AGENT);
                }
                v = 0x7fffffff;        // give up waiting: the results are invalid and reported as such
                break;
            }
        }
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *box = v;
        }
    }
    __syncthreads();
    return *box;
}

__device__ __forceinline__ d4 dag_mb_load(const double* __restrict__ mbq, int b, int J, int lane)
{
    const double* src = mbq + ((size_t)b * MB_BLOCKS + J) * 256;
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = src[r * 64 + lane];
    return v;
}

// A function of its own, like dag_diag_fast and for the same reason (inlined, the kernel around it spills in every other
// task); the tile travels through the stack (128 registers out, 128 in: ~1 us against the ~10 us of the store / drain /
// reload it replaces), LDS is reached through the pointers the kernel hands over (gemm_core.hpp, SmemArg).
template <int C, bool AUG>
__device__ __forceinline__ void dag_pss(Tile& t, double* Km, int ld, int k0, int j0,
                                                  const double* __restrict__ mbq, MatFlags* f, int q, DagCtl* ctl,
                                                  double* Rv, int Npad, lds_double* smem, lds_double* zk, lds_double* colsum,
                                                  int xpub, int skip_rv, int rv_wait)
{
    // t: the updated tile (the caller evaluated the covariance into the accumulators: dag_special)
    const SmemArg sm{smem};
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int lane = tid_ & 63, wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    lds_int* box = (lds_int*)sm.ptr(2 * 8 * 256);
    int avail = 0;                              // steps of the factorisation known to be in the mailbox
    // (rolled over the two halves of the block rows: the register slots stay statically indexed -- row block b is slot
    // b & 3 of the wave row b >> 2 -- at half the code.  Measured and not kept: row blocks dealt to the wave rows by work
    // -- {0, 1, 7, 6} / {2, 3, 5, 4}, 288 MFMAs per wave instead of 416 / 160, with the update's operand rows permuted to
    // match -- together with the next step's operands requested a step early: 5-10 % SLOWER on single evaluations.)
#pragma unroll 1
    for (int hb = 0; hb < 2; ++hb)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        const int b = 4 * hb + mb;
        if (b >= avail) avail = dag_wait_steps(f, 8 * q + b + 1, ctl, box) - 8 * q;
        const int buf = (b & 1) * 8 * 256;
        // the operands of this step that come from the factorisation (issued together: one memory latency per step)
        d4 V = {0.0, 0.0, 0.0, 0.0}, U[4];
        if (wr == hb) V = dag_mb_load(mbq, b, 8, lane);
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (4 * wr + m > b) U[m] = dag_mb_load(mbq, b, 4 * wr + m, lane);
        // X_b: the waves whose row blocks include b (wave-uniform)
        if (wr == hb) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const d4 res = pb::mma16(V, t.acc[mb][n], d4{0.0, 0.0, 0.0, 0.0});
                t.acc[mb][n] = res;
                pb::store_blk(buf + (4 * wc + n) * 256, lane, res, sm);
            }
        }
        __syncthreads();
        // xpub (the tasks of block row q+1 that read this tile follow THIS task): row block b goes to its place in the
        // matrix right away, with write-through stores, from the upper wave row -- it has the fewer products below (none
        // from step 4 on) -- and the flag rises behind them
        if (xpub && wr == 0) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const d4 xb = pb::load_blk(buf + (4 * wc + n) * 256, lane, sm);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __hip_atomic_store(&Km[(size_t)(k0 + tile_row(hb, mb, lane, r)) * ld + j0 + tile_col(wc, n, lane)], xb[r],
                                       PSOAP_RLX_AGENT);
            }
        }
        // T_I -= U_bI^T X_b on this wave's row blocks below b
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (4 * wr + m > b) {
                const d4 xs = d4{-U[m][0], -U[m][1], -U[m][2], -U[m][3]};
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    t.acc[m][n] = pb::mma16(xs, pb::load_blk(buf + (4 * wc + n) * 256, lane, sm), t.acc[m][n]);
            }
        if (xpub && wr == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(&f->xcol[j0 / NB][wc], 8 * q + b + 1, PSOAP_RLX_AGENT);
        }
    }
    // (skip_rv -- tile (q, q+1), delivered to a diagonal task that follows it: that task takes the tile's contribution
    // to the right-hand side itself -- potrf_spine.hpp, SpineFollow)
    if (skip_rv) return;
    // the solved tile goes out (xpub: it is in memory already); the right-hand side update needs z of this block row,
    // final once the factorisation has written it (potrf_done)
    auto row_of = [&](int m, int r) { return tile_row(wr, m, lane, r); };
    if (!xpub) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    dag_st(&Km[(size_t)(k0 + row_of(m, r)) * ld + j0 + tile_col(wc, n, lane)], t.acc[m][n][r]);
    }
    dag_wait_ge(&f->potrf_done, q + 1, ctl, 7u);
    // (rv_wait -- the row above is a following one: its strip solve of this column has applied ITS contribution to the
    // right-hand side block and finished; until round 3's last day only the timing said so -- it runs a whole
    // factorisation ahead)
    if (rv_wait) dag_wait_ge(&f->rvrow[j0 / NB], q, ctl, 8u);
    if (tid_ < NB) zk[tid_] = Rv[k0 + tid_];
    __syncthreads();
    double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double z = zk[row_of(m, r)];
#pragma unroll
            for (int n = 0; n < 4; ++n) part[n] = fma(t.acc[m][n][r], z, part[n]);
        }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        part[n] += __shfl_xor(part[n], 16, 64);
        part[n] += __shfl_xor(part[n], 32, 64);
    }
    if (wr == 1 && lane < 16) {
#pragma unroll
        for (int n = 0; n < 4; ++n) colsum[tile_col(wc, n, lane)] = part[n];
    }
    __syncthreads();
    if (wr == 0 && lane < 16 && j0 < Npad) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = tile_col(wc, n, lane);
            Rv[j0 + col] -= part[n] + colsum[col];
        }
    }
}

__device__ __forceinline__ void dag_negate(Tile& t)
{
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) t.acc[m][n][r] = -t.acc[m][n][r];
}

#ifdef PSOAP_FOLLOW
// ONE out-of-line routine for the two kinds of task that do not fit the kernel body -- the fused diagonal task and the
// following strip solve -- behind ONE call site that takes ONE pointer: with a second call (and its two dozen arguments)
// in the persistent loop, hipcc's allocation for the in-kernel strip solve broke down (spills in every stage of it).
struct DagSpecialArgs {
    int mode;                      // 0: fused diagonal task, 1: following strip solve
    double* Km;
    int ld, k0, j0;
    double* Wm;
    double* Rv;
    MatAcc* acc;
    const double* prev;
    int Npad;
    MatFlags* f;
    DagCtl* ctl;
    int q, ntasks_row, fused;
    int* chain_ctr;
    int chain_len;
    lds_double *smem, *zk, *colsum;
    unsigned long long* tl;
    const double* mbq;
    const double *lw, *gp, *sigma;
    int N, pubnext;
    double scale;
    DagAug aug;
    int pa, pb, n_wait, preload, n_prev;       // the following task does its own update (the task record's fields)
    int* arrive_ctr;
    int xlink;                     // diagonal task: the strip above arrives row block by row block (SpineFollow);
                                   // following strip solve: it delivers its tile that way (dag_pss, xpub)
    int xfirst;                    // the first block row whose strip solves follow
    unsigned int chaos;            // -DPSOAP_CHAOS: the task's key (dag_chaos)
};

// The record travels through LDS (round 4; rounds 2-3: through the caller's stack frame, i.e. 288 B of scratch memory per
// lane written before every call and read back through the vector memory path at the head of the chain's tasks): every
// lane of the caller writes the same values, the callee reads them with broadcast ds_reads.
typedef __attribute__((address_space(3))) const DagSpecialArgs lds_special_args;
// not_tail_called: with no pointer into the caller's frame among the arguments LLVM marks the call `tail`, and a function
// with a `tail` call site is not compiled without callee-saved registers (isSafeForNoCSROpt) -- the routine then saves and
// restores ~270 registers through scratch memory around every task (1120 B per lane).
template <int C, bool AUG, int WPE = 2>
__device__ __attribute__((noinline, not_tail_called)) void dag_special(lds_special_args* a)
{
    if (a->mode == 0) {
        dag_diag_fast(a->Km, a->ld, a->k0, a->Wm, a->Rv, a->acc, a->prev, a->Npad, a->f, a->ctl, a->q, a->ntasks_row,
                      a->fused != 0, a->chain_ctr, a->chain_len, a->smem, a->zk, a->colsum, a->tl, a->xlink != 0,
                      a->pb - a->pa == 2, a->chaos);
        return;
    }
    // the whole task: the tile so far (covariance - the running sum of its chain), the left-looking update over the
    // panels that are left -- the last one FOLLOWING the strip solves of the row above where they deliver row block by
    // row block -- and then the tile, still in the accumulators, is solved behind the factorisation of block q.
    // (The argument record is read ONCE, into scalars: behind a pointer every use is a load of its own -- the stores in
    // between may alias -- and each costs a memory round trip on the chain.)
    const auto si = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    const auto sp = [](auto* p) {
        const unsigned long long u = (unsigned long long)(uintptr_t)p;
        const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)u), hi = __builtin_amdgcn_readfirstlane((unsigned int)(u >> 32));
        return (decltype(p))(uintptr_t)(((unsigned long long)hi << 32) | lo);
    };
    // (... and only what the update needs before it: whatever else is live across the K-loops is what hipcc spills and
    // reloads inside their stages)
    double* const Km = sp(a->Km);
    const double* const prev = sp(a->prev);
    MatFlags* const f = sp(a->f);
    DagCtl* const ctl = sp(a->ctl);
    int* const arrive_ctr = sp(a->arrive_ctr);
    unsigned long long* const tl = sp(a->tl);
    const int ld = si(a->ld), k0 = si(a->k0), j0 = si(a->j0), q = si(a->q);
    const int pa = si(a->pa), pb = si(a->pb), n_wait = si(a->n_wait), preload = si(a->preload), n_prev = si(a->n_prev);
    const int xlink = si(a->xlink);
    const double scale = a->scale;
    lds_double* const smem = (lds_double*)(uintptr_t)(unsigned int)si((int)(unsigned int)(uintptr_t)a->smem);   // wave-uniform: a scalar
    const SmemArg sm{smem};
    const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    Tile t;
    t.zero();
    if (n_wait > 0) {
        dag_wait_ge(arrive_ctr, n_wait, ctl, 4u);
        if (tl && threadIdx.x == 0) tl[1] = __builtin_amdgcn_s_memrealtime();
        dag_sub_partials<false, true>(t, prev, preload ? 1 : n_prev);
    }
    // the covariance goes in BEFORE the wait for the row above (the accumulators then hold -(tile) during the update);
    // the final of a chain has nothing to add -- the chain's first part carried K
    if (scale != 0.0) {
        GpDev g;
        load_gp(a->gp, C, g);
        double dsum = g.a2[0];
        {
#pragma clang fp contract(off)
            for (int c = 1; c < C; ++c) dsum = dsum + g.a2[c];
        }
        // (member by member: the record lives in LDS, DagAug's copy constructor takes a generic reference)
        const DagAug aug_l{a->aug.Pt, a->aug.R, a->aug.Rpad, a->aug.colx, a->aug.rowx, a->aug.diag, a->aug.S, a->aug.lds};
        dag_store_updated<C, AUG, true>(t, nullptr, 0, k0, j0, a->lw, g, dsum, a->sigma, a->N, scale, a->Npad, aug_l, nullptr);
        dag_negate(t);
    }
    const int xupd = xlink && q >= si(a->xfirst) + 1;      // the row above is a following one that delivers its tiles progressively
    if (xupd) {
        if (pb - pa > 1) {
            dag_wait_ge(&f->rows_done, pb - 1, ctl, 1u);
            const size_t r0 = (size_t)pa * NB;
            tile_gemm_tn<true, SmemArg>(t, Km + r0 * ld + k0, (size_t)ld, Km + r0 * ld + j0, (size_t)ld, (pb - 1 - pa) * NB, false,
                                        0x7fffffff, wave_s, sm);
        }
        const size_t r1 = (size_t)(pb - 1) * NB;
        if (tl && threadIdx.x == 0) tl[2] = __builtin_amdgcn_s_memrealtime();
        dag_update_following(t, Km + r1 * ld + k0, Km + r1 * ld + j0, (size_t)ld, f->xcol[q], f->xcol[j0 / NB], 8 * (q - 1),
                             &ctl->error, wave_s, sm, tl);
    } else {
        dag_update<true, SmemArg>(t, Km, ld, k0, j0, pa, pb, f, ctl, false, tl, wave_s, sm);
    }
    dag_negate(t);       // the accumulators hold the sums with the opposite sign: tile = -acc
    if (tl && threadIdx.x == 0) tl[5] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();     // the K-loop's LDS is free: every wave has left it
    double* const Rv = sp(a->Rv);
    const double* const mbq = sp(a->mbq);
    const int Npad = si(a->Npad), ntasks_row = si(a->ntasks_row), pubnext = si(a->pubnext);
    dag_pss<C, AUG>(t, Km, ld, k0, j0, mbq, f, q, ctl, Rv, Npad, smem, a->zk, a->colsum, xlink, xlink && pubnext,
                    q > si(a->xfirst));
    if (tl && threadIdx.x == 0) tl[6] = __builtin_amdgcn_s_memrealtime();
    dag_chaos(a->chaos, 2u);     // (the tail of a following strip solve)
    dag_drain();
    if (threadIdx.x == 0) {
        dag_release_fence();
        if (pubnext) __hip_atomic_store(&f->next_done, q + 1, PSOAP_RLX_AGENT);   // tile (q, q+1) is final
        __hip_atomic_store(&f->rvrow[j0 / NB], q + 1, PSOAP_RLX_AGENT);           // ... and the right-hand side block j has its share
        dag_task_done(f, q, ntasks_row, 1, ctl);
    }
}
#endif

// ---------------------------------------------------------------------------------------------
// Streamed evaluation (round 4): consecutive ensemble steps through ONE resident launch.
//
// A persistent launch ramps up (first block rows: short K-loops, both workgroups of a compute unit in their epilogues at
// the same time) and drains (the last tasks are all handed out, their holders wait on the row-to-row chain); a second
// launch cannot fill either (its workgroups get compute units only when those of the first exit).  In
// streamed mode the launch stays resident and the MATRICES come and go instead:
//   * the device keeps `n_lanes` matrix workspaces ("lanes": storage, flags, arrival counters, split-K slots).  Every lane
//     runs the SAME single-matrix task list (dag_build_tasks({P}, workers / lanes)) with a ticket counter of its own, so a
//     task only ever waits for smaller tickets of its own lane, which running workgroups hold: no deadlock, and a
//     matrix's order of summation does not depend on what else is in flight -- a proposal's lnprob is bit-identical for
//     every batch size, submission order and world size;
//   * the host writes a proposal into pinned memory and publishes a ring entry; workgroup 0 of the launch, the DISPATCHER,
//     takes no tasks: it polls the ring, pulls the proposal over PCIe into the lane's device arrays (no copy engine, no
//     blit kernel -- neither can run beside a resident grid that owns every register file), forms r = fl - mu, clears the
//     lane's flags and counters and opens the lane (ticket counter = 0);
//   * workers serve the lanes of their own XCD round-robin (lane l belongs to XCD l mod 8: the tiles of one block row
//     share an L2), steal from the others when those have nothing to hand out, and sleep on one word when nothing does;
//   * the workgroup that finishes a matrix's last diagonal block writes lnprob and the submission number straight into
//     pinned host memory: psoap_stream_fetch returns while the other lanes keep the device busy.
// The launch ends when the host closes the stream or nothing was in flight for `idle_ticks`; what was published but not
// opened survives in the ring and the next launch carries on (psoap_gp.hip relaunches on demand).  Every wait is bounded
// and reports through DagCtl::error, mirrored into StreamHost::error.
struct alignas(64) StreamLane {
    unsigned int next;              // next ticket of the matrix in this lane; >= n_tasks: nothing to hand out
    unsigned int pad0;
    unsigned long long seq;         // submission number of that matrix
    unsigned long long stamp;       // when the lane's last burst was handed out (s_memrealtime; 0: none yet)
    unsigned int retired;           // tasks of the matrix that have RETIRED (their last store is out): the one that makes it
                                    // n_tasks reports the result -- see stream_retire
    unsigned int too_fast;          // an orbit submission with |v| >= c somewhere: the result is -inf (sample_parallel.py:186)
    unsigned int tainted;           // a task of that matrix ran on a workgroup that moved (dag_moved_check): the host resubmits
    unsigned int pad[7];
};
constexpr unsigned short STREAM_BURST_END = 0x8000;   // DagTask::b of a lane's task list (the matrix index is the lane):
                                                      // the last ticket of a burst -- the next one starts a block row
struct alignas(64) StreamCursor {
    unsigned int lane;              // the lane the workgroups of this XCD draw from now
    unsigned int pad[15];
};
struct alignas(64) StreamDev {
    unsigned int stop;              // dispatcher -> workers: leave when nothing is left to hand out
    unsigned int opens;             // bumped at every lane opening (and at stop): idle workers sleep on it
    unsigned int pad0[14];
    StreamCursor cur[DAG_QUEUES];   // per XCD.  The lanes of an XCD are served a BURST at a time -- one block row of one
                                    // matrix, its diagonal task first -- in turn: handed out ticket by ticket in turn, a
                                    // row's tasks would start spread over a whole round of the lanes, the row would end a
                                    // task's length before the next one's tickets come up, and every task of that next row
                                    // would wait for it (measured: 6 % slower than one launch per step)
    unsigned long long opened;      // submissions opened so far (the dispatcher's cursor; survives relaunches)
    unsigned long long pad1[7];
    unsigned long long completed;   // matrices finished (added to by the finishing workgroups)
    unsigned long long pad2[7];
};
constexpr int STREAM_RING = 256;    // ring of submissions; at most `n_lanes` (<= 64) are ever outstanding
constexpr int STREAM_MAX_LANES = 64;
// what the host left in the lane's pinned buffer: the ln-wavelengths themselves (psoap_stream_submit), the radial
// velocities per component and epoch -- the dispatcher shifts the chunk's grid (psoap_stream_submit_velocities;
// replicate_wls + lredshift, psoap/data.py:37,61) --, or orbital parameters -- the dispatcher solves Kepler's equation
// per epoch first (psoap_stream_submit_orbits; orbit.models[model](...).get_velocities(), sample_parallel.py:183-187)
enum : int { STREAM_IN_LWL = 0, STREAM_IN_VELOCITIES = 1, STREAM_IN_ORBITS = 2 };
struct StreamEntry {
    int lane;
    int kind;                       // STREAM_IN_* in the low byte, the orbit model (ORB_*) in the next
    double mu;
};
struct StreamResult {
    double lnp;
    unsigned long long seq1;        // submission number + 1 once lnp is valid
    unsigned long long tainted;     // != 0: a workgroup moved under one of the matrix's tasks -- lnp is not to be trusted
};
struct StreamHost {                 // pinned, host-coherent memory
    unsigned long long head;        // host -> device: entries [0, head) are published
    unsigned int close;             // host -> device: leave as soon as everything published is done
    unsigned int pad0[13];
    unsigned int error;             // device -> host: a bounded wait gave up (results invalid) ...
    unsigned int err_code, err_target, err_seen;   // ... and which
    unsigned int exits;             // device -> host: launches that have ended
    unsigned int pad1[11];
    StreamEntry entry[STREAM_RING];
    StreamResult result[STREAM_RING];
};
struct StreamArgs {                 // kernel argument, by value
    StreamLane* lanes;
    StreamDev* dev;
    StreamHost* host;
    const double* h_lw;             // pinned proposals, lane-major: C x N ln-wavelengths ...
    const double* h_gp;             // ... and 2 C hyper-parameters per lane
    const double* fl;               // the chunk's flux vector (device)
    const double* grid;             // observed-frame ln-wavelengths, epoch of every pixel, observation dates (device;
    const int32_t* epoch;           // psoap_chunk_set_grid / _set_dates) -- nullptr: only ln-wavelength submissions
    const double* dates;
    int n_epochs;
    int h_stride;                   // doubles per lane in h_lw: max(C N, 16) -- ln-wavelengths, velocities or parameters
    unsigned int n_lanes, n_tasks, ctrs_per_lane, slots_per_lane;
    int C, N;
    unsigned int gate;              // scheme 0: 100 MHz ticks that have to lie between two bursts (block rows) of one lane
                                    // (stream_next_lane); 0: the lanes of an XCD strictly in turn
    unsigned int pad1;
    unsigned long long idle_ticks;  // 100 MHz ticks without anything in flight after which the launch ends
    unsigned int tlog_cap;          // submissions the task log holds (ring)
    unsigned int pad;
};

#define PSOAP_RLX_SYSTEM __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM

// lnp = -0.5 (z^T z + 2 sum log U_ii)   (covariance.py:329-331); -inf when not positive definite -- the expression of
// k_finalize (chol_kernels.hpp), kept in one place so that both paths round alike
__device__ __forceinline__ double stream_lnp(double logdet_half, double quad, bool bad)
{
    return bad ? -INFINITY : -0.5 * (quad + 2.0 * logdet_half);
}

// Worker side: wave 0 finds the next task -- a ticket of the lane its XCD's cursor points at; when that lane has none, of
// the next lane of the XCD that has (the cursor moves there), else of another XCD's current lane (stealing); sleeps on
// StreamDev::opens when no lane has any.  Returns through LDS.
__device__ __forceinline__ void stream_take(const StreamArgs& st, int home, DagCtl* ctl, unsigned int* s_ticket, int* s_lane)
{
    if (threadIdx.x < 64) {
        const int l = hw_lane();
        int got = -1;
        unsigned int ticket = 0;
        int xcd = home;                 // whose cursor to follow
        bool try_cur = true;
        long long spins = 0;
        for (;;) {
            if (try_cur) {
                unsigned int t = 0xffffffffu, L = 0;
                if (l == 0) {
                    L = poll_word(&st.dev->cur[xcd].lane);
                    t = __hip_atomic_fetch_add(&st.lanes[L].next, 1u, PSOAP_RLX_AGENT);
                }
                t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
                L = (unsigned int)__builtin_amdgcn_readfirstlane((int)L);
                if (t < st.n_tasks) {
                    got = (int)L;
                    ticket = t;
                    break;
                }
            }
            // what the wake-up word holds BEFORE the scan: an opening between scan and sleep is then not missed
            unsigned int seen = 0;
            if (l == 0) seen = poll_word(&st.dev->opens);
            seen = (unsigned int)__builtin_amdgcn_readfirstlane((int)seen);
            unsigned int v = 0xffffffffu;
            if (l < (int)st.n_lanes) v = poll_word(&st.lanes[l].next);
            const unsigned long long m = __ballot(v < st.n_tasks);
            if (m != 0ull) {
                const unsigned long long mine = m & (0x0101010101010101ull << home);
                if (mine) {
                    // the cursor's lane ran dry (its matrix is handed out): on to the next lane of this XCD that has tickets
                    unsigned int L = 0;
                    if (l == 0) L = poll_word(&st.dev->cur[home].lane);
                    L = (unsigned int)__builtin_amdgcn_readfirstlane((int)L);
                    const int start = ((int)L + 8) & 63;
                    const unsigned long long rot = start ? ((mine >> start) | (mine << (64 - start))) : mine;
                    const int pick = (__builtin_ctzll(rot) + start) & 63;
                    if (l == 0) __hip_atomic_store(&st.dev->cur[home].lane, (unsigned int)pick, PSOAP_RLX_AGENT);
                    xcd = home;
                } else {
                    // steal: another XCD's current lane (its burst order is kept), the XCDs tried in turn from a start
                    // that differs from worker to worker
                    const int start = (int)((blockIdx.x * 5u + (unsigned int)spins) & 63u);
                    const unsigned long long rot = start ? ((m >> start) | (m << (64 - start))) : m;
                    const int pick = (__builtin_ctzll(rot) + start) & 63;
                    xcd = pick & 7;
                    unsigned int L = 0;
                    if (l == 0) L = poll_word(&st.dev->cur[xcd].lane);
                    L = (unsigned int)__builtin_amdgcn_readfirstlane((int)L);
                    if (!((m >> (L & 63u)) & 1ull) && l == 0)
                        __hip_atomic_store(&st.dev->cur[xcd].lane, (unsigned int)pick, PSOAP_RLX_AGENT);
                    ++spins;
                }
                try_cur = true;
                continue;
            }
            // nothing to hand out: leave when told to, else sleep until a lane opens
            int leave = 0;
            for (;;) {
                unsigned int o = 0, stop = 0, err = 0;
                if (l == 0) {
                    stop = poll_word(&st.dev->stop);
                    o = poll_word(&st.dev->opens);
                    err = poll_word(&ctl->error);
                }
                stop = (unsigned int)__builtin_amdgcn_readfirstlane((int)stop);
                o = (unsigned int)__builtin_amdgcn_readfirstlane((int)o);
                err = (unsigned int)__builtin_amdgcn_readfirstlane((int)err);
                if (stop != 0u || err != 0u) {
                    leave = 1;
                    break;
                }
                if (o != seen) break;
                __builtin_amdgcn_s_sleep(127);
                __builtin_amdgcn_s_sleep(127);
                if (++spins > 64 * DAG_MAX_SPINS) {      // (the dispatcher ends the launch long before: belt and braces)
                    leave = 1;
                    break;
                }
            }
            if (leave) break;
            xcd = home;
            try_cur = true;
        }
        if (l == 0) {
            *s_ticket = ticket;
            *s_lane = got;
            if (got >= 0) {
                // the lane's arrays were (re)written by the dispatcher before it opened the lane
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    __syncthreads();
}

// The ticket just taken from `lane` ended a burst (thread 0): the XCD's workgroups draw from its next lane now -- the next
// one in turn whose previous burst lies at least `gate` ticks back.  Tickets are handed out in order and cannot be given
// back.  A strip solve needs the row above only for the LAST panel of its update and potrf of its own row at the end, so
// a block row may be handed out long before the row above is through -- but not right behind it: then its tasks reach
// their last panel before the tiles above them are final and some forty workgroups wait (measured with the lanes strictly
// in turn: 5.5 % of all worker time in such waits whenever half the lanes were between two matrices).  About 250 us of
// spacing is what the tail of a task (last panel, covariance, potrf, strip solve) takes.  When no lane of the XCD is
// that far, the turn decides.  (A lane without tickets is skipped by whoever finds it so: stream_take.)
__device__ __forceinline__ void stream_next_lane(const StreamArgs& st, int lane)
{
    const int nl = (int)st.n_lanes;
    const int per = (nl + 7 - (lane & 7)) / 8;            // lanes of this XCD
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    __hip_atomic_store(&st.lanes[lane].stamp, now, PSOAP_RLX_AGENT);
    int fallback = -1, pick = -1;
    int cand = lane;
    for (int k = 0; k < per; ++k) {
        cand = cand + 8 < nl ? cand + 8 : (cand & 7);
        if (!st.gate) {
            pick = cand;
            break;
        }
        if (poll_word(&st.lanes[cand].next) >= st.n_tasks) continue;   // nothing to hand out
        if (fallback < 0) fallback = cand;
        const unsigned long long last = cand == lane ? now : __hip_atomic_load(&st.lanes[cand].stamp, PSOAP_RLX_AGENT);
        if (now - last >= (unsigned long long)st.gate) {
            pick = cand;
            break;
        }
    }
    if (pick < 0) pick = fallback >= 0 ? fallback : (lane + 8 < nl ? lane + 8 : (lane & 7));
    __hip_atomic_store(&st.dev->cur[lane & 7].lane, (unsigned int)pick, PSOAP_RLX_AGENT);
}

// The matrix of `lane` is complete: lnprob and the submission number go straight to pinned host memory.  Called by ALL
// 64 lanes of wave 0 of the workgroup whose task retired last (dag_task_end).
// The sums over the matrix's P block records (common.hpp, MatAcc) in block order -- acc_total's order (chol_kernels.hpp),
// so a stream returns the bits a launch per step returns for the same task list: lane l fetches record l (l + 64, ...)
// with returning atomics (rmw_read: the value the MEMORY holds -- a load could be served from a line this XCD's L2 keeps
// from an earlier matrix of the lane), then the records are added one after the other, each broadcast with v_readlane.
// Six registers per lane: the version of this that ONE thread ran with eight records in flight cost the stream kernels
// 13-20 more spilled registers.
__device__ __forceinline__ double stream_bcast(unsigned long long w, int k)
{
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)w, k);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(w >> 32), k);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ void stream_complete(const StreamArgs& st, int lane, MatAcc* rec, int P)
{
    const int l = hw_lane();
    double lh = 0.0, qd = 0.0;
    int bad = 0;
    for (int q0 = 0; q0 < P; q0 += 64) {
        unsigned long long wl = 0ull, wq = 0ull, wi = 0ull;
        if (q0 + l < P) {
            MatAcc* r = rec + q0 + l;
            wl = rmw_read(reinterpret_cast<unsigned long long*>(&r->logdet_half));
            wq = rmw_read(reinterpret_cast<unsigned long long*>(&r->quad));
            wi = rmw_read(reinterpret_cast<unsigned long long*>(&r->info));
        }
        const int n = P - q0 < 64 ? P - q0 : 64;
        for (int k = 0; k < n; ++k) {            // (wave-uniform: every lane ends with the same sums)
            lh += stream_bcast(wl, k);
            qd += stream_bcast(wq, k);
            bad |= stream_bcast(wi, k) != 0.0 ? 1 : 0;
        }
    }
    if (l == 0) {
        const unsigned long long seq = __hip_atomic_load(&st.lanes[lane].seq, PSOAP_RLX_AGENT);
        const unsigned int fast = __hip_atomic_load(&st.lanes[lane].too_fast, PSOAP_RLX_AGENT);
        const unsigned int taint = __hip_atomic_load(&st.lanes[lane].tainted, PSOAP_RLX_AGENT);
        StreamResult* res = &st.host->result[seq % STREAM_RING];
        __hip_atomic_store(&res->lnp, stream_lnp(lh, qd, bad != 0 || fast != 0u), PSOAP_RLX_SYSTEM);
        __hip_atomic_store(&res->tainted, (unsigned long long)taint, PSOAP_RLX_SYSTEM);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&res->seq1, seq + 1ull, PSOAP_RLX_SYSTEM);
        __hip_atomic_fetch_add(&st.dev->completed, 1ull, PSOAP_RLX_AGENT);
    }
}

// A task of the matrix in `lane` has retired (thread 0, behind the task's last store): true for the task that retires
// LAST.  The result is reported by that one, not by the last diagonal task: in the latency schemes the tasks that FOLLOW a
// factorisation hand their tiles over row block by row block, the diagonal task they feed can be through before they have
// written their own completion words -- and the host, told too early, resubmits to the lane, whose flags the dispatcher
// then clears under a straggler's late store (seen as a rare wrong lnprob with eight processes sharing one GPU).
__device__ __forceinline__ bool stream_retire(const StreamArgs& st, int lane)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's completion words are out
    const unsigned int old = __hip_atomic_fetch_add(&st.lanes[lane].retired, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1u != st.n_tasks) return false;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return true;
}

// The end of every task: thread 0 checks whether the workgroup moved (dag_moved_check) and, in a stream, counts the task as
// retired; wave 0 of the workgroup whose task was the matrix's last then reports the result (stream_complete).
// wave_s: the wave index as a scalar (the branch on it is a scalar branch, not an exec-masked region).
template <bool STREAM>
__device__ __forceinline__ void dag_task_end(const StreamArgs& st, int b, MatAcc* acc, int P, unsigned int where0, DagCtl* ctl,
                                             int wave_s)
{
    int last = 0;
    if (threadIdx.x == 0) {
        dag_moved_check(where0, ctl, STREAM ? &st.lanes[b].tainted : nullptr);
        if constexpr (STREAM) last = stream_retire(st, b) ? 1 : 0;
    }
    if constexpr (STREAM) {
        if (wave_s == 0) {
            if (__builtin_amdgcn_readfirstlane(last) != 0) stream_complete(st, b, acc, P);
        }
    }
}

// Workgroup 0 of a streamed launch (all 256 threads): open lanes as the host publishes submissions, end the launch.
__device__ __forceinline__ void stream_dispatch(const StreamArgs& st, const DagMat* __restrict__ mats, MatFlags* flags,
                                                int* arrive, DagCtl* ctl, unsigned long long* box /* LDS, 4 words */,
                                                double* vbuf /* LDS: 3 x n_epochs velocities + 16 parameters */)
{
    const int tid = threadIdx.x;
    unsigned long long opened = __hip_atomic_load(&st.dev->opened, PSOAP_RLX_AGENT);
    unsigned long long last_busy = __builtin_amdgcn_s_memrealtime();
    unsigned int sweep = 0;
    for (;;) {
        // A lane nobody submits to rests at next = 0x40000000 and every failed take of a worker whose cursor points at it
        // adds one: long before the word could wrap (2e7 matrices at N = 6000) it is put back.  Only this workgroup opens
        // lanes, so the exchange cannot undo an opening; a worker's add in between makes it fail, and the next sweep retries.
        if ((++sweep & 4095u) == 0u && tid < (int)st.n_lanes) {
            unsigned int v = poll_word(&st.lanes[tid].next);
            if (v >= 0x60000000u)
                (void)__hip_atomic_compare_exchange_strong(&st.lanes[tid].next, &v, 0x40000000u, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) {
            box[0] = __hip_atomic_load(&st.host->head, PSOAP_RLX_SYSTEM);
            box[1] = (unsigned long long)__hip_atomic_load(&st.host->close, PSOAP_RLX_SYSTEM);
            box[2] = poll_word(&st.dev->completed);
            box[3] = (unsigned long long)poll_word(&ctl->error);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const unsigned long long head = box[0], closing = box[1], completed = box[2], err = box[3];
        __syncthreads();
        if (err != 0ull) {
            if (tid == 0) {
                __hip_atomic_store(&st.host->err_code, __hip_atomic_load(&ctl->pad[0], PSOAP_RLX_AGENT), PSOAP_RLX_SYSTEM);
                __hip_atomic_store(&st.host->err_target, __hip_atomic_load(&ctl->pad[1], PSOAP_RLX_AGENT), PSOAP_RLX_SYSTEM);
                __hip_atomic_store(&st.host->err_seen, __hip_atomic_load(&ctl->pad[2], PSOAP_RLX_AGENT), PSOAP_RLX_SYSTEM);
                __hip_atomic_store(&st.host->error, (unsigned int)err, PSOAP_RLX_SYSTEM);
            }
            break;
        }
        if (opened < head) {
            int* const fastv = reinterpret_cast<int*>(vbuf + 3 * st.n_epochs + 16);      // per entry of this pass: |v| >= c
            // everything published so far in one go: the proposals of all its lanes first (one pass of PCIe pulls), ONE
            // drain and release (the write-back of this XCD's L2 is what a release costs), then the lanes open together
            const unsigned int nb = (unsigned int)(head - opened < (unsigned long long)st.n_lanes ? head - opened : st.n_lanes);
            bool bad = false;
            for (unsigned int k = 0; k < nb; ++k) {
                const StreamEntry* e = &st.host->entry[(opened + k) % STREAM_RING];
                // (every thread reads the same entry: said to be wave-uniform, so that the branches below are scalar branches
                // and not exec-masked regions -- the shape the build's assembly scan objects to)
                const int lane = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&e->lane, PSOAP_RLX_SYSTEM));
                const int kind_word = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&e->kind, PSOAP_RLX_SYSTEM));
                const int kind = kind_word & 0xff, model = (kind_word >> 8) & 0xff;
                const double mu = __hip_atomic_load(&e->mu, PSOAP_RLX_SYSTEM);
                if (lane < 0 || lane >= (int)st.n_lanes || kind > STREAM_IN_ORBITS ||
                    (kind != STREAM_IN_LWL && (!st.grid || !st.epoch || (kind == STREAM_IN_ORBITS && !st.dates)))) {
                    bad = true;                                    // a corrupt entry: refuse, loudly
                    break;
                }
                const DagMat mat = mats[lane];
                if (kind == STREAM_IN_LWL && tid == 0) fastv[k] = 0;
                if (kind != STREAM_IN_LWL) {
                    // velocities (C x n_epochs) into LDS -- from the host as they are, or from the orbital parameters
                    // there with the arithmetic of k_orbit_velocities (one thread per epoch) -- then the Doppler
                    // shift of the chunk's grid with the arithmetic of k_doppler_shift
                    const double* src = st.h_lw + (size_t)lane * st.h_stride;
                    const int nv = st.C * st.n_epochs;
                    unsigned int fast = 0u;
                    __syncthreads();                               // (vbuf of the previous entry has been read)
                    if (tid == 0) fastv[k] = 0;
                    if (kind == STREAM_IN_VELOCITIES) {
                        for (int i = tid; i < nv; i += GEMM_THREADS) vbuf[i] = __hip_atomic_load(&src[i], PSOAP_RLX_SYSTEM);
                    } else {
                        const int np = orbit_n_params(model);
                        double* par = vbuf + 3 * st.n_epochs;
                        if (tid < np) par[tid] = __hip_atomic_load(&src[tid], PSOAP_RLX_SYSTEM);
                        __syncthreads();
                        for (int ep = tid; ep < st.n_epochs; ep += GEMM_THREADS) {
                            double v[3];
                            orbit_velocities_at(model, par, st.dates[ep], v);
                            for (int k = 0; k < st.C; ++k) {
                                vbuf[k * st.n_epochs + ep] = v[k];
                                if (fabs(v[k]) >= C_KMS) fast = 1u;     // sample_parallel.py:186-187
                            }
                        }
                    }
                    if (fast) fastv[k] = 1;                        // (after the barrier above; every writer writes 1)
                    __syncthreads();
                    double* dst = const_cast<double*>(mat.lw);
                    for (int k = 0; k < st.C; ++k)
                        for (int i = tid; i < st.N; i += GEMM_THREADS) {
                            const double v = vbuf[k * st.n_epochs + st.epoch[i]];
                            dst[(size_t)k * st.N + i] = st.grid[i] + (-v) / C_KMS;
                        }
                    if (tid < 2 * st.C)
                        const_cast<double*>(mat.gp)[tid] = __hip_atomic_load(&st.h_gp[(size_t)lane * 2 * st.C + tid], PSOAP_RLX_SYSTEM);
                } else
                // the proposal: pinned host memory -> the lane's device arrays (uncached system-scope loads, sixteen in
                // flight per thread: one after the other, a lane's 96 KB took 48 PCIe round trips)
                {
                    const size_t n = (size_t)st.C * st.N;
                    const double* src = st.h_lw + (size_t)lane * st.h_stride;
                    double* dst = const_cast<double*>(mat.lw);
                    size_t i = tid;
                    for (; i + 15 * GEMM_THREADS < n; i += 16 * GEMM_THREADS) {
                        double v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[u] = __hip_atomic_load(&src[i + (size_t)u * GEMM_THREADS], PSOAP_RLX_SYSTEM);
#pragma unroll
                        for (int u = 0; u < 16; ++u) dst[i + (size_t)u * GEMM_THREADS] = v[u];
                    }
                    for (; i < n; i += GEMM_THREADS) dst[i] = __hip_atomic_load(&src[i], PSOAP_RLX_SYSTEM);
                    if (tid < 2 * st.C)
                        const_cast<double*>(mat.gp)[tid] = __hip_atomic_load(&st.h_gp[(size_t)lane * 2 * st.C + tid], PSOAP_RLX_SYSTEM);
                }
                // r = fl - mu (covariance.py:331), padded with zeros; accumulators, flags and arrival counters cleared
                for (int i = tid; i < mat.Npad; i += GEMM_THREADS) mat.R[i] = (i < mat.N) ? (st.fl[i] - mu) : 0.0;
                // (nothing to clear in mat.acc: every block's record is written by its factorisation -- common.hpp, MatAcc)
                int* fz = reinterpret_cast<int*>(flags + lane);
                for (int i = tid; i < (int)(sizeof(MatFlags) / sizeof(int)); i += GEMM_THREADS) fz[i] = 0;
                int* az = arrive + (size_t)lane * st.ctrs_per_lane;
                for (int i = tid; i < (int)st.ctrs_per_lane; i += GEMM_THREADS) az[i] = 0;
            }
            if (bad) {
                if (tid == 0) {
                    __hip_atomic_fetch_or(&ctl->error, 2u, PSOAP_RLX_AGENT);
                    __hip_atomic_store(&ctl->pad[0], 100u, PSOAP_RLX_AGENT);
                }
                __syncthreads();
                continue;
            }
            dag_drain();
            if (tid == 0) {
                dag_release_fence();
                for (unsigned int k = 0; k < nb; ++k) {
                    const int lane = __hip_atomic_load(&st.host->entry[(opened + k) % STREAM_RING].lane, PSOAP_RLX_SYSTEM);
                    __hip_atomic_store(&st.lanes[lane].seq, opened + k, PSOAP_RLX_AGENT);
                    __hip_atomic_store(&st.lanes[lane].stamp, 0ull, PSOAP_RLX_AGENT);
                    __hip_atomic_store(&st.lanes[lane].retired, 0u, PSOAP_RLX_AGENT);
                    __hip_atomic_store(&st.lanes[lane].too_fast, (unsigned int)fastv[k], PSOAP_RLX_AGENT);
                    __hip_atomic_store(&st.lanes[lane].tainted, 0u, PSOAP_RLX_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (unsigned int k = 0; k < nb; ++k) {
                    const int lane = __hip_atomic_load(&st.host->entry[(opened + k) % STREAM_RING].lane, PSOAP_RLX_SYSTEM);
                    __hip_atomic_store(&st.lanes[lane].next, 0u, PSOAP_RLX_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&st.dev->opened, opened + nb, PSOAP_RLX_AGENT);
                __hip_atomic_fetch_add(&st.dev->opens, 1u, PSOAP_RLX_AGENT);
            }
            opened += nb;
            last_busy = __builtin_amdgcn_s_memrealtime();
            continue;
        }
        if (completed < opened) {
            last_busy = __builtin_amdgcn_s_memrealtime();
        } else if (closing != 0ull || __builtin_amdgcn_s_memrealtime() - last_busy > st.idle_ticks) {
            break;      // everything published is done, and the host has closed the stream or gone quiet
        }
        __builtin_amdgcn_s_sleep(32);
    }
    if (tid == 0) {
        __hip_atomic_store(&st.dev->stop, 1u, PSOAP_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&st.dev->opens, 1u, PSOAP_RLX_AGENT);
        __hip_atomic_fetch_add(&st.host->exits, 1u, PSOAP_RLX_SYSTEM);
    }
}

// Ready-only hand-out (DagPool): wave 0 finds this workgroup's next task -- the head final of a queue when it may be handed
// out, else the first READY part in a window of that queue's pool, its own queue first, then the others from `steal0` on.
// *s_ticket: the task's index in tasks[], or 0xffffffff when every queue is through (or the launch has failed).
// Everything read here changes under the reader: the flag words are read with atomics (executed at the memory side, like
// every poll of this kernel), claims are atomic compare-exchange / or.  Per scan of a queue: the ticket word, the pool's
// first-untaken cursor, two words of the taken bitmap (a window of 64 parts), rows_done once per matrix in the window and
// one wave instruction of arrival-counter reads.
constexpr unsigned int DAG_NO_TASK = 0xffffffffu;
#ifndef PSOAP_POOL_WINDOWS
#define PSOAP_POOL_WINDOWS 4
#endif
constexpr unsigned int DAG_POOL_WINDOWS = PSOAP_POOL_WINDOWS;
__device__ __forceinline__ void pool_take(const DagTask* __restrict__ tasks, const DagQueues& queues, const DagPool& pool,
                                          DagCtl* ctl, MatFlags* flags, int home, int steal0,
                                          unsigned int* s_ticket, unsigned int* s_pending)
{
    // *s_pending (in / out, LDS): the position in order[] of a final this workgroup has drawn a ticket for but may not
    // run yet -- its chain's last part has not been taken.  Finals are drawn with ONE fetch-add each, like the tickets of
    // the in-order list (a compare-exchange per final serialised the workgroups: one winner per round trip, 30-60 us per
    // block row); whoever overshoots the finals that may be handed out keeps its ticket and takes ready parts meanwhile.
    if (threadIdx.x < 64) {
        const int l = hw_lane();
        unsigned int got = DAG_NO_TASK;
        unsigned int pending = (unsigned int)__builtin_amdgcn_readfirstlane((int)*s_pending);
        long long spins = 0;
        auto taken_bit = [&](unsigned int d) -> unsigned int {
            unsigned int w = 0u;
            if (l == 0) w = poll_word(&pool.taken[d >> 5]);
            return ((unsigned int)__builtin_amdgcn_readfirstlane((int)w) >> (d & 31u)) & 1u;
        };
        for (;;) {
            bool left = pending != DAG_NO_TASK;     // something is still to be handed out somewhere
            if (pending != DAG_NO_TASK) {
                const unsigned int d = pool.dep[pending];
                if (d == DAG_POOL_NONE || taken_bit(d)) {
                    got = pool.order[pending];
                    pending = DAG_NO_TASK;
                    break;
                }
            }
            for (int probe = 0; probe <= DAG_QUEUES && got == DAG_NO_TASK; ++probe) {
                const int g = probe == 0 ? home : (steal0 + probe - 1) % DAG_QUEUES;
                if (probe > 0 && g == home) continue;
                const unsigned int lo = queues.first[g], n_all = queues.first[g + 1] - lo, n_main = pool.n_main[g];
                if (n_all == 0u) continue;
                // 1. the head final: handed out in list order, and only once the last part of its chain has been taken
                unsigned int m = 0u;
                if (l == 0) m = poll_word(&ctl->queue[g].next);
                m = (unsigned int)__builtin_amdgcn_readfirstlane((int)m);
                if (m < n_main) {
                    left = true;
                    const unsigned int d = pool.dep[lo + m];
                    if (pending == DAG_NO_TASK && (d == DAG_POOL_NONE || taken_bit(d))) {
                        unsigned int tk = 0u;
                        if (l == 0) tk = __hip_atomic_fetch_add(&ctl->queue[g].next, 1u, PSOAP_RLX_AGENT);
                        tk = (unsigned int)__builtin_amdgcn_readfirstlane((int)tk);
                        if (tk < n_main) {
                            const unsigned int d2 = pool.dep[lo + tk];
                            if (d2 == DAG_POOL_NONE || taken_bit(d2)) {
                                got = pool.order[lo + tk];
                                break;
                            }
                            pending = lo + tk;      // drawn ahead of what may run: held, parts meanwhile
                        }
                    }
                }
                // 2. the pool: the 64 parts of the two bitmap words around the first untaken one
                const unsigned int p0 = lo + n_main, p1 = lo + n_all;       // the pool's positions in order[]
                if (p0 == p1) continue;
                unsigned int cur = 0u;
                if (l == 0) cur = poll_word(&ctl->queue[g].fill[0]);
                cur = p0 + (unsigned int)__builtin_amdgcn_readfirstlane((int)cur);
                if (cur >= p1) continue;
                left = true;
                // (a second window when the first holds nothing ready: the parts at the cursor may all wait for a predecessor
                // that is running, while the next rows' are ready)
#pragma unroll 1
                for (unsigned int win = 0u; win < DAG_POOL_WINDOWS && got == DAG_NO_TASK; ++win) {
                const unsigned int base = (cur & ~31u) + 64u * win;
                if (base >= p1) break;
                unsigned int word = 0xffffffffu;
                if (l < 2 && base + 32u * (unsigned int)l < p1)
                    word = poll_word(&pool.taken[(base >> 5) + (unsigned int)l]);
                const unsigned int w0 = (unsigned int)__builtin_amdgcn_readlane((int)word, 0);
                const unsigned int w1 = (unsigned int)__builtin_amdgcn_readlane((int)word, 1);
                const unsigned int pos = base + (unsigned int)l;
                const bool mine = pos >= cur && pos < p1 && !(((l < 32 ? w0 : w1) >> (l & 31)) & 1u);     // untaken, in range
                // the cursor moves up to the first untaken part (monotone; a stale view only makes it lag)
                if (win == 0u) {
                    const unsigned long long un = __ballot(mine);
                    const unsigned int first_un = un ? base + (unsigned int)__builtin_ctzll(un) : (base + 64u < p1 ? base + 64u : p1);
                    if (l == 0 && first_un > cur) __hip_atomic_fetch_max(&ctl->queue[g].fill[0], first_un - p0, PSOAP_RLX_AGENT);
                }
                // ready: the block rows its panels read are complete and its predecessor's running sum is there
                DagTask t{};
                if (mine) t = tasks[pool.order[pos]];
                bool ready = false;
                {
                    unsigned long long todo = __ballot(mine);
                    while (todo) {                  // rows_done once per matrix of the window
                        const int src = __builtin_ctzll(todo);
                        const int b0 = __builtin_amdgcn_readlane((int)t.b, src);
                        int rd = 0;
                        if (l == 0) rd = dag_peek(&flags[b0].rows_done);
                        rd = __builtin_amdgcn_readfirstlane(rd);
                        const bool same = mine && (int)t.b == b0;
                        if (same) ready = rd >= (int)t.pb;
                        todo &= ~__ballot(same);
                    }
                }
                if (ready && (t.type & DAG_CHAIN) && t.S > 0) {
                    // the predecessor has been taken (it runs, or is through): the part may start, it waits for the
                    // predecessor's tile only when its own update is done
                    const unsigned int pp = pool.dep[pos];
                    ready = (poll_word(&pool.taken[pp >> 5]) >> (pp & 31u)) & 1u;
                }
                unsigned long long rdy = __ballot(ready);
                while (rdy) {
                    // (the workgroups scan the same window at the same time: each starts at another ready part)
                    const int n_rdy = __builtin_popcountll(rdy);
                    int skip = (int)(blockIdx.x % (unsigned int)n_rdy);
                    unsigned long long r2 = rdy;
                    while (skip-- > 0) r2 &= r2 - 1ull;
                    const int pick = __builtin_ctzll(r2);
                    unsigned int old = 0xffffffffu;
                    if (l == pick) old = __hip_atomic_fetch_or(&pool.taken[pos >> 5], 1u << (pos & 31u), PSOAP_RLX_AGENT);
                    old = (unsigned int)__builtin_amdgcn_readlane((int)old, pick);
                    const unsigned int ppos = base + (unsigned int)pick;
                    if (!((old >> (ppos & 31u)) & 1u)) {
                        got = pool.order[ppos];
                        break;
                    }
                    rdy &= ~(1ull << pick);
                }
                }
            }
            if (got != DAG_NO_TASK || !left) break;
            unsigned int err = 0u;
            if (l == 0) err = poll_word(&ctl->error);
            if (__builtin_amdgcn_readfirstlane((int)err) != 0) break;
            __builtin_amdgcn_s_sleep(64);
            if (++spins > DAG_MAX_SPINS) {          // (nothing became ready for seconds: a failed launch, reported)
                if (l == 0 && __hip_atomic_fetch_or(&ctl->error, 1u, PSOAP_RLX_AGENT) == 0u)
                    __hip_atomic_store(&ctl->pad[0], 10u, PSOAP_RLX_AGENT);
                break;
            }
        }
        if (l == 0) {
            *s_ticket = got;
            *s_pending = pending;
        }
    }
    __syncthreads();
}

// LAT: the instantiation launched for task lists of the latency scheme; only it contains the fused diagonal
// fast path (dag_diag_fast).  With that path compiled into the one kernel, hipcc keeps a spilled value in the
// MFMA loops of every task (a scratch load per 64-MFMA stage: 32-walker batch 39.5 -> 43.7 ms); the
// throughput scheme never runs it, so it gets a kernel without it.
// STREAM: the resident form (above): tickets come from the lanes' own counters, matrix index = lane, workgroup 0
// dispatches.  The task bodies are the same code.
// WPE: waves per SIMD the kernel is compiled for.  2 = two workgroups per compute unit, 256 registers per lane.  1 (LAT
// kernels launched with at most one workgroup per compute unit -- single evaluations, predict: dag_pick_workers): the whole
// unified file, 512 registers per lane -- the accumulator tile can live in the AccVGPR half and what the chain phases of the
// out-of-line routine spilled to scratch memory (29-62 VGPRs, 632-904 B per lane with 256) stays in registers.
template <int C, bool AUG = false, bool LAT = false, bool STREAM = false, int WPE = 2>
__global__ __launch_bounds__(GEMM_THREADS, WPE) void k_chol_dag(const DagMat* __restrict__ mats,
                                                             const DagTask* __restrict__ tasks, DagQueues queues,
                                                             MatFlags* flags, int* arrive, double* wspace,
                                                             DagCtl* ctl, unsigned long long* tlog, DagAug aug,
                                                             StreamArgs st, DagPool pool)
{
    __shared__ double vec1[NB];   // z_k (OFF)
    __shared__ double vec2[NB];   // column sums (OFF)
    __shared__ unsigned int s_ticket;
    __shared__ unsigned int s_pending;      // ready-only hand-out: a final drawn ahead of its turn (pool_take)
    __shared__ int s_lane;
    if constexpr (STREAM) {
        if (blockIdx.x == 0) {
            stream_dispatch(st, mats, flags, arrive, ctl, reinterpret_cast<unsigned long long*>(vec1), psoap_smem);
            return;
        }
    }
    constexpr size_t SLOT = (size_t)NB * NB;   // doubles per workspace slot
    // the XCD this workgroup runs on: its queue first (L2 locality), the others when it runs dry.
    // Stealing starts at a queue picked uniformly among the NON-EMPTY ones (by workgroup index): with
    // fewer than 8 matrices the idle XCDs' workgroups would otherwise all pile onto queue 0 and the
    // matrices would finish one after the other (B = 4, N = 6000: 15.7 -> 10 ms).
    const int home = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20 /* HW_REG_XCC_ID[3:0] */) & 7u);
    int steal0 = 0;
    {
        unsigned int mask = 0;
        for (int g = 0; g < DAG_QUEUES; ++g) mask |= (queues.first[g + 1] > queues.first[g] ? 1u : 0u) << g;
        const int nq = __builtin_popcount(mask);
        int sel = nq > 0 ? (int)(blockIdx.x % (unsigned int)nq) : 0;
        for (int g = 0; g < DAG_QUEUES; ++g)
            if ((mask >> g) & 1u) {
                if (sel == 0) {
                    steal0 = g;
                    break;
                }
                --sel;
            }
    }
    // the wave index as a scalar for the staging of every K-loop (the lane id comes from the exec mask: no thread-id
    // register lives across a K-loop), and the strip solve with its row blocks dealt to the waves by work -- together
    // 38.9 -> 38.7 ms per 32-walker step; the balanced solve alone brings a thread-id reload into the K-loop stages.
    // Both kinds of kernel use these forms (round 2 kept the LAT kernels on the plain ones only because any change to
    // them could bring the fault of DESIGN.md 3.4 back; measured in round 3: within 1 % either way, profiles/r3_ab_forms.txt)
    const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    unsigned int dry = 0;                       // bit g: queue g is exhausted (wave-uniform)
    int probe = 0;
    // Scheme 0 (round 4): the strip solve of tile (q, q+1) CONTINUES into the diagonal task of block q+1 (DAG_FUSED on that
    // OFF final; the DIAG record is the next entry of the list, flagged DAG_NOSOLVE = "owned": whoever draws its ticket
    // skips it).  The tile above the diagonal is all DIAG(q+1) waits for, so it starts the moment it can, never holds a
    // workgroup waiting (257 us per diagonal task before, 1.9 % of all worker time of a streamed run) and potrf(q+1) is
    // out long before the strip solves of row q+1 ask for it.
    constexpr bool CONT = !LAT && DAG_TILE_DEPS;
    // ready-only hand-out of the PART tasks (DagPool): an experiment of round 5, compiled in with -DPSOAP_POOL only -- it is
    // SLOWER than the list order (see the comment at DagPool)
#ifdef PSOAP_POOL
    constexpr bool POOL = LAT && !STREAM;
#else
    constexpr bool POOL = false;
#endif
    (void)pool;
    if (threadIdx.x == 0) s_pending = DAG_NO_TASK;
    __syncthreads();
    bool cont = false;
    unsigned int ticket = 0;
    int b = 0;
    for (;;) {
        Tile t;
        const bool owned_run = CONT && cont;
        cont = false;
        if (owned_run) {
            ticket += 1u;           // the record behind the strip solve just finished: its diagonal task (same matrix)
        } else if constexpr (STREAM) {
            stream_take(st, home, ctl, &s_ticket, &s_lane);
            ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)s_ticket);
            b = __builtin_amdgcn_readfirstlane(s_lane);
            __syncthreads();  // both are rewritten by the next stream_take
            if (b < 0) return;
            // the lane's proposal arrays change from matrix to matrix: nothing of the previous one may survive in the
            // scalar cache (uniform loads -- the hyper-parameters -- go through it; the vector caches were invalidated by
            // the acquire in stream_take)
            asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        } else if constexpr (POOL) {
            // the latency schemes' plain launches: finals in list order, parts ready-only (pool_take)
            pool_take(tasks, queues, pool, ctl, flags, home, steal0, &s_ticket, &s_pending);
            ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)s_ticket);
            __syncthreads();  // s_ticket is rewritten by the next pool_take
            if (ticket == DAG_NO_TASK) return;
        } else {
            if (dry == (1u << DAG_QUEUES) - 1u) return;
            const int g = (probe == 0) ? home : (steal0 + probe - 1) % DAG_QUEUES;
            if (dry & (1u << g)) {
                ++probe;
                continue;
            }
            if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&ctl->queue[g].next, 1u, PSOAP_RLX_AGENT);
            __syncthreads();
            // wave-uniform by construction: keep it (and everything decoded from it) in scalar registers
            const unsigned int local = __builtin_amdgcn_readfirstlane(s_ticket);
            __syncthreads();  // s_ticket is rewritten at the top of the next iteration
            if (local >= queues.first[g + 1] - queues.first[g]) {
                dry |= 1u << g;
                ++probe;
                continue;
            }
            ticket = queues.first[g] + local;
        }
        if (poll_word(&ctl->error) != 0u) return;
        // the compute unit this task starts on (compared before it retires: dag_moved_check)
        const unsigned int where0 = dag_where();
        // (ticket and matrix index are loop-carried since round 4 -- a strip solve continues into the next record, a stream
        // keeps its lane: said to be wave-uniform HERE, so that the records below are scalar loads and the pointers in them
        // scalars -- as loop-carried values the compiler kept copies of them in vector registers and spilled those)
        ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
        const DagTask task = tasks[ticket];
        if constexpr (!STREAM) b = task.b;
        b = __builtin_amdgcn_readfirstlane(b);
        if constexpr (STREAM) {
            // (the burst ends with the TICKET, whoever runs the task)
            if (!owned_run && (task.b & STREAM_BURST_END) && threadIdx.x == 0) stream_next_lane(st, b);
        }
        if constexpr (CONT) {
            if (!owned_run && (task.type & DAG_TYPE_MASK) == DAG_DIAG && (task.type & DAG_NOSOLVE)) continue;   // owned
        }
        const int q = task.q, j = task.j;
        const DagMat mat = mats[b];
        // STREAM: arrival counters, split-K slots and task-log rows of this lane
        int* const arrive_l = STREAM ? arrive + (size_t)b * st.ctrs_per_lane : arrive;
        double* const wspace_l = STREAM ? wspace + (size_t)b * st.slots_per_lane * ((size_t)NB * NB) : wspace;
        unsigned long long* tlog_l = tlog;
        if constexpr (STREAM) {
            if (tlog) {
                const unsigned long long sq = __hip_atomic_load(&st.lanes[b].seq, PSOAP_RLX_AGENT);
                tlog_l = tlog + (size_t)(sq % st.tlog_cap) * st.n_tasks * 8;
            }
        }
        double* Km = mat.K;
        double* Rv = mat.R;
        // (scheme 0: three Wt tiles in turn, the third one behind the mailbox)
        double* Wm = CONT ? (q % 3 == 2 ? mat.Wt + WT_THIRD : mat.Wt + (size_t)(q % 3) * NB * NB)
                          : mat.Wt + (size_t)(q & 1) * NB * NB;
        const int ld = mat.ld, N = mat.N, Npad = mat.Npad;
        MatFlags* f = flags + b;
        const int k0 = q * NB, j0 = j * NB;
        const int ntasks_row = (AUG ? aug.Pt : mat.P) - q;

        if (tlog_l && threadIdx.x == 0) {
            tlog_l[ticket * 8 + 0] = __builtin_amdgcn_s_memrealtime();
            // where it ran: the XCD and HW_REG_HW_ID (wave slot, SIMD, CU, SH, SE) -- tools/dag_cu_overlap.py
            tlog_l[ticket * 8 + 7] = ((unsigned long long)(unsigned int)home << 32) |
                                   (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11));
        }
        unsigned int chaos_key = 0u;
#ifdef PSOAP_CHAOS
        chaos_key = ticket * 2654435761u;
        if constexpr (STREAM) chaos_key ^= (unsigned int)__hip_atomic_load(&st.lanes[b].seq, PSOAP_RLX_AGENT) * 40503u;
        else chaos_key ^= (unsigned int)__builtin_amdgcn_readfirstlane((int)__builtin_amdgcn_s_memrealtime()) & 0xffffff00u;
        dag_chaos(chaos_key, 0u);        // (a task that starts late)
#endif
        const int ttype = task.type & DAG_TYPE_MASK;
        const bool chain = (task.type & DAG_CHAIN) != 0;
        const bool is_part = (ttype == DAG_PART);
        // partial sums of a split tile, two schemes (dag_emit): gathered -- the final waits for all S-1
        // PARTs and adds their S-1 tiles; chained -- PART s waits for PART s-1 and adds its tile to its
        // own, the final starts from the last one
        const int n_wait = is_part ? (chain ? (int)task.S : 0) : (int)task.S - 1;
        int n_prev = is_part ? (n_wait > 0 ? 1 : 0) : (chain ? (n_wait > 0 ? 1 : 0) : n_wait);
        // chained PART: the predecessor's tile sits in the other slot of the even/odd pair
        const double* prev = wspace_l + (size_t)(is_part ? task.slot ^ 1u : task.slot) * SLOT;
        const bool preload = !is_part && chain && n_wait > 0;
#ifdef PSOAP_FOLLOW
        // (-DPSOAP_FOLLOW: both out-of-line task kinds go through dag_special, further down, at one call site)
        const bool fast_diag = LAT && preload && ttype == DAG_DIAG && (task.type & DAG_WAITNEXT) &&
                               (task.pb - task.pa == 1 || task.pb == 0 || ((task.type & DAG_NOSOLVE) && task.pb - task.pa == 2));
        const bool follow = LAT && ttype == DAG_OFF && (task.type & DAG_WAITNEXT);
        if (fast_diag || follow) {
            // scheme 2's strip solves follow the factorisation of block q step by step, on the tile in registers; the
            // whole task -- update, covariance evaluation, solve -- runs in the out-of-line routine
            __shared__ DagSpecialArgs args;      // (every lane writes the same record; the barriers of the ticket hand-out
                                                 // separate its readers from the next task's writes)
            args.mode = fast_diag ? 0 : 1;
            args.Km = Km; args.ld = ld; args.k0 = k0; args.j0 = j0; args.Wm = Wm; args.Rv = Rv; args.acc = mat.acc;
            args.prev = prev; args.Npad = Npad; args.f = f; args.ctl = ctl; args.q = q; args.ntasks_row = ntasks_row;
            args.fused = (task.type & DAG_FUSED) != 0; args.chain_ctr = &arrive_l[task.ctr]; args.chain_len = n_wait;
            args.smem = dag_opaque_lds(psoap_smem); args.zk = dag_opaque_lds(vec1); args.colsum = dag_opaque_lds(vec2);
            args.tl = tlog_l ? tlog_l + ticket * 8 : nullptr;
            args.mbq = mat.Wt + 2 * NB * NB + mb_slot(q, 0, 0);
            args.lw = mat.lw; args.gp = mat.gp; args.sigma = mat.sigma; args.N = N;
            args.pubnext = (task.type & DAG_NOSOLVE) != 0;
            args.scale = (chain ? task.S <= 1 : true) ? 1.0 : 0.0;
            args.aug = aug;
            args.pa = task.pa; args.pb = task.pb; args.n_wait = n_wait; args.preload = preload ? 1 : 0; args.n_prev = n_prev;
            args.arrive_ctr = &arrive_l[task.ctr];
            // second level of following: DAG_NOSOLVE on the diagonal task (it follows the strip solve of the tile above),
            // DAG_FUSED on a strip solve (it delivers its tile row block by row block, and follows the row above likewise)
            args.xlink = fast_diag ? ((task.type & DAG_NOSOLVE) != 0) : ((task.type & DAG_FUSED) != 0);
            args.xfirst = (int)queues.follow_first;
            args.chaos = chaos_key;
            __syncthreads();
            {
                unsigned int aa = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) DagSpecialArgs*)&args;
                asm volatile("" : "+s"(aa));     // (opaque, like dag_opaque_lds: the callee must not be specialised on it)
                dag_special<C, AUG, WPE>((lds_special_args*)(uintptr_t)aa);
            }
            if (tlog_l && threadIdx.x == 0) tlog_l[ticket * 8 + 3] = __builtin_amdgcn_s_memrealtime();
            dag_task_end<STREAM>(st, b, mat.acc, mat.P, where0, ctl, wave_s);
            continue;
        }
#else
        if (LAT && preload && ttype == DAG_DIAG && (task.type & DAG_WAITNEXT) && task.pb - task.pa == 1) {
            dag_diag_fast(Km, ld, k0, Wm, Rv, mat.acc, prev, Npad, f, ctl, q, ntasks_row, (task.type & DAG_FUSED) != 0,
                          &arrive_l[task.ctr], n_wait, dag_opaque_lds(psoap_smem), dag_opaque_lds(vec1), dag_opaque_lds(vec2),
                          tlog_l ? tlog_l + ticket * 8 : nullptr);
            dag_task_end<STREAM>(st, b, mat.acc, mat.P, where0, ctl, wave_s);
            continue;
        }
#endif
        t.zero();
        if (preload) {
            // the chain ran ahead (its PARTs need older block rows): normally no wait at all
            dag_wait_ge(&arrive_l[task.ctr], n_wait, ctl, 4u);
            dag_sub_partials<false, true>(t, prev, 1);
            n_prev = 0;
        }
        // (DAG_WAITNEXT on a DIAG task: its last panel needs only the tile right of the diagonal above; on an OFF task
        // the bit means "follow the factorisation" and the last panel needs the whole block row above, as always)
        // (scheme 0: tile-level dependencies; never in the LAT kernels, whose task lists carry their own protocols)
        dag_update<true, SmemKernel, false, !LAT && DAG_TILE_DEPS>(
            t, Km, ld, k0, j0, task.pa, task.pb, f, ctl, ttype == DAG_DIAG && (task.type & DAG_WAITNEXT) != 0,
            tlog_l ? tlog_l + ticket * 8 : nullptr, wave_s, SmemKernel(), &f->rvrow[q], &f->rvrow[j]);
        if (tlog_l && threadIdx.x == 0) tlog_l[ticket * 8 + 5] = __builtin_amdgcn_s_memrealtime();
        if (!preload && n_wait > 0) dag_wait_ge(&arrive_l[task.ctr], n_wait, ctl, 4u);
        if (tlog_l && threadIdx.x == 0) tlog_l[ticket * 8 + 6] = __builtin_amdgcn_s_memrealtime();
        dag_sub_partials<false, true>(t, prev, n_prev);
        {
            GpDev g;
            load_gp(mat.gp, C, g);
            double dsum = g.a2[0];
            {
#pragma clang fp contract(off)
                for (int c = 1; c < C; ++c) dsum = dsum + g.a2[c];
            }
            double* dest = is_part ? (wspace_l + (size_t)task.slot * SLOT) : (Km + (size_t)k0 * ld + j0);
            size_t ldd = is_part ? (size_t)NB : (size_t)ld;
            double* mirror = nullptr;
            if (AUG && ttype == DAG_SCHUR) {
                // a tile of Sigma: rows k0 - Npad .., columns j0 - Npad .. of DagAug::S, and its transpose
                dest = aug.S + (size_t)(k0 - Npad) * aug.lds + (size_t)(j0 - Npad);
                ldd = aug.lds;
                mirror = aug.S + (size_t)(j0 - Npad) * aug.lds + (size_t)(k0 - Npad);     // == dest on the diagonal
            }
            // who adds K(i, j): the final task -- except in a chain, where the FIRST PART carries it, so
            // that the exp() evaluations are off the critical row-to-row path (the final of a chain runs
            // right after the block row above completes; its PARTs ran ahead)
            const bool carries_k = chain ? (is_part ? task.S == 0 : task.S <= 1) : !is_part;
            if (tlog_l && is_part && threadIdx.x == 0) tlog_l[ticket * 8 + 1] = __builtin_amdgcn_s_memrealtime();   // folded
            dag_store_updated<C, AUG, false, false, DAG_FAST_STORE(STREAM)>(t, dest, ldd, k0, j0, mat.lw, g, dsum, mat.sigma, N, carries_k ? 1.0 : 0.0, Npad, aug,
                                      mirror, !(AUG && ttype == DAG_SCHUR));
        }
        if (tlog_l && !is_part && threadIdx.x == 0) tlog_l[ticket * 8 + 4] = __builtin_amdgcn_s_memrealtime();     // a final's stores issued
        if (is_part) {
            if (tlog_l && threadIdx.x == 0) tlog_l[ticket * 8 + 2] = __builtin_amdgcn_s_memrealtime();              // stores issued
            dag_chaos(chaos_key, 3u);
            dag_drain();
            if (tlog_l && threadIdx.x == 0) tlog_l[ticket * 8 + 4] = __builtin_amdgcn_s_memrealtime();              // drained
            if (threadIdx.x == 0) {
                dag_release_fence();
                __hip_atomic_fetch_add(&arrive_l[task.ctr], 1, PSOAP_RLX_AGENT);
            }
            if (tlog_l && threadIdx.x == 0) tlog_l[ticket * 8 + 3] = __builtin_amdgcn_s_memrealtime();
            dag_task_end<STREAM>(st, b, mat.acc, mat.P, where0, ctl, wave_s);
            continue;
        }
        if (AUG && ttype == DAG_SCHUR) {             // nobody inside the launch reads Sigma: no drain, no counter
            if (threadIdx.x == 0) dag_moved_check(where0, ctl, nullptr);
            continue;
        }
        dag_drain();  // the tile is re-read below in another layout by other waves of this block
        if (tlog_l && threadIdx.x == 0) tlog_l[ticket * 8 + 1] = __builtin_amdgcn_s_memrealtime();
        if (ttype == DAG_DIAG) {
            // the diagonal task is the row-to-row critical path: let its waves win the issue arbitration
            // against the co-resident workgroup (N = 2000, B = 32: -4 % with the deferred W output)
            __builtin_amdgcn_s_setprio(3);
            // (tile-level dependencies: block q - 3's strip solves read the W tile this factorisation writes, and the row
            // counters take turns likewise -- all of row q - 3 has to be through; it has been for two rows' time)
            // (no branch around the poll: rows_done >= 0 always holds)
            if constexpr (CONT) dag_wait_ge(&f->rows_done, q >= 3 ? q - 2 : 0, ctl, 9u);
            potrf_blocked(Km, ld, k0, Wm, Rv, mat.acc);
            dag_chaos(chaos_key, 4u);
            dag_drain();
            if (tlog_l && threadIdx.x == 0) tlog_l[ticket * 8 + 2] = __builtin_amdgcn_s_memrealtime();
            if (threadIdx.x == 0) {
                dag_release_fence();
                if (q > 0) dag_spin_ge(&f->potrf_done, q, ctl, 12u);        // (in order: dag_spin_ge)
                __hip_atomic_store(&f->potrf_done, q + 1, PSOAP_RLX_AGENT);
            }
            if (task.type & DAG_FUSED) {
                // the critical path stays in this workgroup: solve the tile right of the diagonal now (its
                // updated value was prepared by a DAG_NOSOLVE task while the block was being factored) and
                // hand U(q, q+1) -- all that the next diagonal tile needs from this block row -- to DIAG(q+1)
                dag_wait_ge(&f->off1_ready, q + 1, ctl, 5u);
                dag_trsm(t, Km, ld, k0, k0 + NB, Wm, Rv, Npad, vec1, vec2);
                dag_drain();
                if (threadIdx.x == 0) {
                    dag_release_fence();
                    __hip_atomic_store(&f->next_done, q + 1, PSOAP_RLX_AGENT);
                    dag_task_done(f, q, ntasks_row, 2, ctl);
                }
            } else if (threadIdx.x == 0) {
                dag_task_done<CONT>(f, q, ntasks_row, 1, ctl);
            }
            __builtin_amdgcn_s_setprio(0);
        } else if (task.type & DAG_NOSOLVE) {
            if (threadIdx.x == 0) {
                dag_release_fence();
                __hip_atomic_store(&f->off1_ready, q + 1, PSOAP_RLX_AGENT);
            }
        } else {
            dag_wait_ge(&f->potrf_done, q + 1, ctl, 3u + 16u * (unsigned int)q + 4096u * (unsigned int)b);
            if (tlog_l && threadIdx.x == 0) tlog_l[ticket * 8 + 2] = __builtin_amdgcn_s_memrealtime();
            dag_trsm<true>(t, Km, ld, k0, j0, Wm, Rv, Npad, vec1, vec2);
            dag_chaos(chaos_key, 5u);
            dag_drain();
            if (threadIdx.x == 0) {
                dag_release_fence();
                // (counted BEFORE the tile is announced: whatever reads the tile belongs to a later row, so a row is
                // complete -- rows_done -- in row order)
                dag_task_done<CONT>(f, q, ntasks_row, 1, ctl);
                // tile (q, j) is final and its share of the right-hand side block j applied
                if constexpr (CONT) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_store(&f->rvrow[j], q + 1, PSOAP_RLX_AGENT);
                }
            }
            if constexpr (CONT) cont = (task.type & DAG_FUSED) != 0;
        }
        if (tlog_l && threadIdx.x == 0) tlog_l[ticket * 8 + 3] = __builtin_amdgcn_s_memrealtime();
        dag_task_end<STREAM>(st, b, mat.acc, mat.P, where0, ctl, wave_s);
    }
}

// ---------------------------------------------------------------------------------------------
// Host: build the task lists (one queue per XCD, matrix b in queue b mod 8) for a batch of B
// matrices of P block rows on `workers` persistent workgroups.  Ticket order inside a queue
// (every wait targets a smaller ticket of the same queue):
//   for each block row q:  DIAG finals of row q (the queue's matrices)
//                          PARTs that pre-accumulate the diagonal tile of row q+1 over rows < q
//                          PARTs + OFF finals of row q (b-major, then j)
// ---------------------------------------------------------------------------------------------
struct DagPlan {
    std::vector<DagTask> tasks;
    // ready-only hand-out (DagPool): filled for the latency schemes by dag_build_tasks
    std::vector<unsigned int> order, dep;
    unsigned int n_main[DAG_QUEUES] = {};
    DagQueues queues{};
    unsigned int n_slots = 0;
    unsigned int n_ctrs = 0;
    int scheme = 0;            // 0 throughput, 1 latency: selects the kernel instantiation (k_chol_dag<.., LAT>)
};

inline int dag_split_factor(int tasks_in_row, int q, int workers, int scheme, int n_mats = 0, bool augmented = false)
{
    // cut tiles of sparse block rows until the row offers about `workers` tasks (at most 8 parts).
    // Throughput scheme: full occupancy, parts at least two panels long.  Latency scheme: half the workers,
    // parts at least four panels long -- every part costs a round trip of its 128 KB partial tile through the
    // workspace and a dependency hand-off, and the workers that are not on a matrix's critical path have
    // slack (measured, tools/split_sweep.sh: N = 2000, B = 32: 3.0 -> 2.5 ms; N = 6000, B = 4: 7.7 -> 7.1 ms;
    // single evaluations unchanged; a quarter of the workers is too few from N = 6000, B = 8 on).
    // PSOAP_DAG_SPLIT_PCT / PSOAP_DAG_SPLIT_MIN override both numbers (experiments).
    static const int env_pct = getenv("PSOAP_DAG_SPLIT_PCT") ? atoi(getenv("PSOAP_DAG_SPLIT_PCT")) : 0;
    static const int env_min = getenv("PSOAP_DAG_SPLIT_MIN") ? atoi(getenv("PSOAP_DAG_SPLIT_MIN")) : 0;
    // (scheme 2: 35 % -- with the PARTs handed out just in time the chains run ahead of the finals anyway, and every part
    // less is a partial-tile hand-over less; measured over N = 4096 .. 8192, B = 1 .. 8: 25 / 35 / 50 / 70 %)
    // (predict -- appended columns: the launch is bound by throughput, not by its chain: 494 of 512 workgroups busy, 88 % of
    // their time in PART tasks (tools/predict_timeline.py), and every part less is a 128 KB partial tile that does not
    // travel: 25 % measured 9.85-10.0 ms against 10.25-10.3 at 35 %, profiles/r5_experiments.txt)
    const int pct = env_pct > 0 ? env_pct : (scheme == 2 ? (augmented ? 25 : 35) : (scheme >= 1 ? 50 : 100));
    const int minp = env_min > 0 ? env_min : (scheme >= 1 ? 4 : 2);
    int S = 1;
    while (S < 8 && tasks_in_row * S * 100 < workers * pct && minp * S <= q) S *= 2;
    return S;
}

// Tasks of one tile whose update over panels [pa_first, pb_last) is cut into pieces.
//   scheme 0 (throughput): nsplit equal ranges; the first nsplit-1 are PARTs, the final takes the last
//     range and GATHERS the nsplit-1 partial tiles.  Least work per tile; right when other matrices of
//     the queue hide the wait for the block row above.
//   scheme 1 (latency): nsplit PARTs over equal ranges of [pa_first, pb_last - 1) plus a final that
//     covers the LAST panel only -- the one piece that has to wait for the block row above, kept as
//     short as the dependency allows (the PARTs wait for older rows and run ahead).  The partial sums
//     are CHAINED: PART s adds the tile PART s-1 left in the previous slot to its own, so every task
//     reads one partial tile.  Right when a queue holds one matrix and the row-to-row chain is the
//     critical path.
// Encoding: PART.S = index in the chain (0 when gathered), PART.slot = its output (gathered:
// consecutive slots; chained: an even/odd pair used alternately); final.S = number of pieces,
// final.slot = first slot to read (gather: the first PART's, chain: the last PART's).  nsplit == 1: one final over the whole range.
inline int dag_final_panels()
{
    const char* e = getenv("PSOAP_FINAL_PANELS");      // experiments
    return e ? atoi(e) : 2;
}
inline int dag_jit_rows()
{
    const char* e = getenv("PSOAP_DAG_JIT");      // experiments; 0: readiness order
    return e ? atoi(e) : 6;
}
inline int dag_follow_first_row()
{
    const char* e = getenv("PSOAP_FOLLOW_ROW0");      // experiments; 0: rows 0 and 1 keep the forms of scheme 1
    return (e && e[0] == '0') ? 2 : 0;
}
inline bool dag_xfollow_enabled()
{
    const char* e = getenv("PSOAP_XFOLLOW");
    return !(e && e[0] == '0');
}
// final_panels: how many of the last panels a chain's final takes itself (1: only the one that depends on the block row
// above; 2 -- following strip solves: the chain's last PART then needs the row before that only and is folded in a whole
// row period before the final gets its last operands -- the hand-over of a partial tile costs 30-40 us)
inline void dag_emit(DagPlan& plan, int type, int b, int q, int j, int pa_first, int pb_last, int nsplit, int scheme,
                     unsigned char final_flags = 0, int final_panels = 1)
{
    if (final_panels > 1) {
        const int left = pb_last - final_panels - pa_first;         // panels for the PARTs
        if (left <= 0) nsplit = 1;
        else if (nsplit > left) nsplit = left;
        if (scheme < 1 || nsplit <= 1) final_panels = 1;            // (only a chain's final has a fixed range)
    }
    const bool chain = (scheme >= 1) && nsplit > 1;
    const int nparts = chain ? nsplit : nsplit - 1;                 // PART tasks
    const unsigned int ctr = (nparts > 0) ? plan.n_ctrs++ : 0u;
    if (chain) plan.n_slots += plan.n_slots & 1u;                   // a chain ping-pongs between an even/odd slot pair
    const unsigned int slot0 = plan.n_slots;
    const int pb_parts = chain ? pb_last - final_panels : pb_last;
    const int span = pb_parts - pa_first;
    const unsigned char flag = chain ? DAG_CHAIN : 0;
    for (int sidx = 0; sidx < nparts; ++sidx) {
        DagTask t{};
        t.type = DAG_PART | flag;
        t.b = (unsigned short)b;
        t.q = (unsigned char)q;
        t.j = (unsigned char)j;
        t.S = (unsigned char)(chain ? sidx : 0);
        t.pa = (unsigned char)(pa_first + (long long)span * sidx / nsplit);
        t.pb = (unsigned char)(pa_first + (long long)span * (sidx + 1) / nsplit);
        // gathered: one slot per PART; chained: PART s reads slot0 + ((s - 1) & 1) and writes slot0 + (s & 1)
        // (its predecessor's reader -- itself -- is the only one, so two slots per tile are enough)
        t.slot = chain ? slot0 + (unsigned int)(sidx & 1) : plan.n_slots++;
        t.ctr = ctr;
        plan.tasks.push_back(t);
    }
    if (chain) plan.n_slots = slot0 + 2;
    DagTask t{};
    t.type = (unsigned char)type | flag | final_flags;
    t.b = (unsigned short)b;
    t.q = (unsigned char)q;
    t.j = (unsigned char)j;
    t.S = (unsigned char)(nparts + 1);
    t.pa = (unsigned char)(chain ? pb_last - final_panels : pa_first + (long long)span * (nsplit - 1) / nsplit);
    t.pb = (unsigned char)pb_last;
    t.slot = (nparts > 0) ? (chain ? slot0 + (unsigned int)((nparts - 1) & 1) : slot0) : 0u;
    t.ctr = ctr;
    plan.tasks.push_back(t);
}

// task list of ONE queue: the matrices in `mats`, served by about `workers` workgroups
// `Bq_nominal` (the largest queue's matrix count) decides the split factors, so every matrix of the
// batch gets the same task structure and identical proposals give identical bits in any batch slot
// Tasks of the Schur complement of the appended columns (predict: Sigma = A - W^T W), Ms x Ms tiles, upper triangle:
// tile (P + i, P + j) is a left-looking update over ALL P block rows with nothing to solve afterwards -- work that
// needs no critical path, cut into chained parts of `len` panels whose boundaries are staggered from tile to tile, so
// that about the same number of parts becomes ready with every finished block row and the workgroups that wait on the
// factorisation's row-to-row chain always find one.  The final covers the last panel and stores into DagAug::S.
inline void dag_emit_schur(DagPlan& plan, int b, int P, int Ms, int len = 8)
{
    static const int env_len = getenv("PSOAP_SCHUR_LEN") ? atoi(getenv("PSOAP_SCHUR_LEN")) : 0;   // experiments
    if (env_len > 0) len = env_len;
    int tile = 0;
    for (int i = 0; i < Ms; ++i)
        for (int j = i; j < Ms; ++j, ++tile) {
            std::vector<int> cuts;                           // part boundaries in [0, P - 1]
            cuts.push_back(0);
            for (int c = 1 + tile % len; c < P - 1; c += len) cuts.push_back(c);
            if (P - 1 > cuts.back()) cuts.push_back(P - 1);
            const int nparts = (int)cuts.size() - 1;         // PARTs cover [0, P - 1); may be 0 when P == 1
            const unsigned int ctr = plan.n_ctrs++;
            plan.n_slots += plan.n_slots & 1u;
            const unsigned int slot0 = plan.n_slots;
            for (int sidx = 0; sidx < nparts; ++sidx) {
                DagTask t{};
                t.type = DAG_PART | DAG_CHAIN;
                t.b = (unsigned short)b;
                t.q = (unsigned char)(P + i);
                t.j = (unsigned char)(P + j);
                t.S = (unsigned char)sidx;
                t.pa = (unsigned char)cuts[sidx];
                t.pb = (unsigned char)cuts[sidx + 1];
                t.slot = slot0 + (unsigned int)(sidx & 1);
                t.ctr = ctr;
                plan.tasks.push_back(t);
            }
            plan.n_slots = slot0 + (nparts > 1 ? 2 : (nparts > 0 ? 1 : 0));
            DagTask fin{};
            fin.type = DAG_SCHUR | DAG_CHAIN;
            fin.b = (unsigned short)b;
            fin.q = (unsigned char)(P + i);
            fin.j = (unsigned char)(P + j);
            fin.S = (unsigned char)(nparts + 1);
            fin.pa = (unsigned char)(P - 1);
            fin.pb = (unsigned char)P;
            fin.slot = nparts > 0 ? slot0 + (unsigned int)((nparts - 1) & 1) : 0u;
            fin.ctr = ctr;
            plan.tasks.push_back(fin);
        }
}

// PSOAP_FIXED_PLAN=1 (round 4): every matrix gets the task structure of a stream lane (dag_build_lane_plan: scheme 0, the
// split factors of ONE matrix on the nominal share of the workgroups) whatever the batch -- so the order of summation
// inside a matrix, and with it every bit of its lnprob, is the same for every batch size, for every number of chunks in
// a launch, for every number of GPUs, and equal to what a stream returns.  What it costs: small batches lose the
// latency schemes (a single N = 6000 evaluation: 11 ms instead of 2.6).  For runs that have to be reproducible across
// world sizes (an MH chain decided in the last bits: the reference's np.sum over chunks is deterministic,
// psoap/sample_parallel.py:387).
inline bool dag_fixed_plan()
{
    const char* e = getenv("PSOAP_FIXED_PLAN");
    return e && e[0] == '1';
}
constexpr int STREAM_NOMINAL_LANES = 32;
inline int dag_nominal_share(int workers_total) { const int s = workers_total / STREAM_NOMINAL_LANES / 2; return s > 0 ? s : 1; }

inline void dag_build_queue(DagPlan& plan, const std::vector<int>& mats, const std::vector<int>& Ps, int workers,
                            int Bq_nominal, int scheme, int Mt = 0, int Ms = 0, int fixed_share = 0)
{
    // Ps[b]: block rows of matrix b.  A heterogeneous batch (matrices of several chunks) walks the block
    // rows of all its matrices together; a matrix simply drops out once its rows are used up.
    if (mats.empty()) return;
    int P = 0;
    bool uniform = true;
    for (int b : mats) {
        P = Ps[b] > P ? Ps[b] : P;
        uniform = uniform && Ps[b] == Ps[mats[0]];
    }
    std::vector<std::vector<DagTask>> early_final(P);   // DIAG finals whose PARTs were emitted a row early
    // scheme 0 (round 4): the final of DIAG(q+1) sits right BEHIND the final of tile (q, q+1) and is run by the workgroup
    // that ran that one (DAG_FUSED on the OFF final: "continue with the next record"; DAG_NOSOLVE on the DIAG final:
    // "owned", skipped by whoever draws its ticket) -- see k_chol_dag
    const bool cont0 = (scheme == 0) && DAG_TILE_DEPS;
    std::vector<DagTask> owned_final(Ps.size());          // per matrix: the DIAG(q+1) final to emit behind tile (q, q+1)
    std::vector<char> has_owned(Ps.size(), 0);
    for (int q = 0; q < P; ++q) {
        // tiles of this block row in the queue; uniform batches use the nominal matrix count so that the
        // split factors do not depend on the slot a matrix sits in
        long long row_tiles = 0;
        int live = 0;
        for (int b : mats)
            if (q < Ps[b]) {
                row_tiles += Ps[b] + Mt - q;
                ++live;
            }
        if (uniform) {
            row_tiles = (long long)Bq_nominal * (P + Mt - q);
            live = Bq_nominal;
        }
        const int Bq = live;
        // (fixed_share > 0: per matrix, from its own size only)
        auto s_off = [&](int b) {
            return fixed_share > 0 ? dag_split_factor(Ps[b] + Mt - q, q, fixed_share, scheme, 1)
                                   : dag_split_factor((int)row_tiles, q, workers, scheme, (int)Ps.size(), Mt > 0);
        };
        // latency scheme: DIAG(q) also solves the tile right of the diagonal (DAG_FUSED) whenever a next
        // diagonal tile exists, and DIAG(q >= 1) waits only for that tile of the row above (DAG_WAITNEXT)
        // scheme 2 ("following"): from block row 2 on -- where the diagonal task is the fused fast one, which publishes its
        // block rows step by step -- the strip solves FOLLOW the factorisation (dag_pss: DAG_WAITNEXT on an OFF task),
        // the diagonal task solves nothing itself, and the strip solve of tile (q, q+1) publishes next_done (DAG_NOSOLVE
        // on a following OFF task).  Rows 0 and 1 keep the forms of scheme 1.
        // (dag_follow_first_row(): 0 -- the first block rows follow as well: their diagonal tasks then need a running sum to
        // start from, which a PART with an empty range provides, it "carries K"; 2: rows 0 and 1 in the forms of scheme 1)
        const int q_f = dag_follow_first_row();
        const bool following = (scheme == 2) && q >= q_f;
        // (block row r: its solved tiles are delivered row block by row block (DAG_FUSED on a following OFF task) to the
        // tasks of block row r+1 that read them -- the diagonal task of block r+1 (DAG_NOSOLVE on a DIAG task) and the
        // last panel of the strip solves' updates; PSOAP_XFOLLOW=0 keeps the first level only -- A/B measurements)
        auto xlink = [&](int r) { return scheme == 2 && r >= q_f && dag_xfollow_enabled(); };
        auto fused = [&](int b) { return scheme >= 1 && !following && q + 1 < Ps[b]; };
        // 1. DIAG finals of this row
        if (q <= 1 && following) {
            // the fused fast diagonal task (the one that publishes its steps) for the first block rows too: a chain of
            // one PART over no panels -- the covariance tile -- and the final over [0, q)
            for (int b : mats) {
                if (q >= Ps[b]) continue;
                const unsigned int ctr = plan.n_ctrs++;
                plan.n_slots += plan.n_slots & 1u;
                const unsigned int slot0 = plan.n_slots;
                plan.n_slots = slot0 + 1;
                DagTask t{};
                t.type = DAG_PART | DAG_CHAIN;
                t.b = (unsigned short)b;
                t.q = t.j = (unsigned char)q;
                t.S = 0;
                t.pa = t.pb = 0;
                t.slot = slot0;
                t.ctr = ctr;
                plan.tasks.push_back(t);
                DagTask fin{};
                fin.type = DAG_DIAG | DAG_CHAIN | DAG_WAITNEXT | ((q == 1 && xlink(0)) ? DAG_NOSOLVE : 0);
                fin.b = (unsigned short)b;
                fin.q = fin.j = (unsigned char)q;
                fin.S = 2;
                fin.pa = 0;
                fin.pb = (unsigned char)q;
                fin.slot = slot0;
                fin.ctr = ctr;
                plan.tasks.push_back(fin);
            }
        } else if (q <= 1) {
            for (int b : mats)
                if (q < Ps[b]) {
                    if (cont0 && q == 1) {
                        // DIAG(1): one final over [0, 1), emitted behind tile (0, 1) -- row 0 has come by already: here
                        // only for a matrix whose row 0 had no such tile (never: q < Ps[b] means P >= 2)
                        continue;
                    }
                    dag_emit(plan, DAG_DIAG, b, q, q, 0, q, 1, scheme,
                             (unsigned char)((fused(b) ? DAG_FUSED : 0) | (scheme >= 1 && q == 1 ? DAG_WAITNEXT : 0)));
                }
        } else {
            for (const DagTask& t : early_final[q]) plan.tasks.push_back(t);
        }
        // 2. pre-accumulate the diagonal tile of row q+1 over rows [0, q): PARTs now, final (panel q) later.
        // Latency scheme: the PART that needs the block row just above (panel q-1, available only when ALL
        // of row q-1 is finished) is one panel long; the long ones cover [0, q-1) and run a row earlier.
        if (q + 1 < P && q >= 1) {
            const int S_pre = fixed_share > 0
                                  ? dag_split_factor(1, q, fixed_share / 4 > 0 ? fixed_share / 4 : 1, scheme, 1)
                                  : dag_split_factor(Bq, q, workers / 4 > 0 ? workers / 4 : 1, scheme, (int)Ps.size());
            for (int b : mats) {
                if (q + 1 >= Ps[b]) continue;
                const unsigned int ctr = plan.n_ctrs++;
                const bool chain = (scheme >= 1);
                if (chain) plan.n_slots += plan.n_slots & 1u;
                const unsigned int slot0 = plan.n_slots;
                const unsigned char flag = chain ? DAG_CHAIN : 0;
                // ranges of the PARTs
                std::vector<std::pair<int, int>> ranges;
                // (second level of following, two-panel finals: the diagonal task itself applies panel q-1 -- with a PART
                // for it, the hand-over of the partial tile sat on the row-to-row path)
                const bool two = xlink(q) && dag_final_panels() == 2;
                if (chain && q >= 2) {
                    int S_long = S_pre;
                    while (S_long > 1 && (q - 1) / S_long < 1) S_long /= 2;
                    for (int sidx = 0; sidx < S_long; ++sidx)
                        ranges.emplace_back((int)((long long)(q - 1) * sidx / S_long),
                                            (int)((long long)(q - 1) * (sidx + 1) / S_long));
                    if (!two) ranges.emplace_back(q - 1, q);
                } else {
                    for (int sidx = 0; sidx < S_pre; ++sidx)
                        ranges.emplace_back((int)((long long)q * sidx / S_pre), (int)((long long)q * (sidx + 1) / S_pre));
                }
                const int n_parts = (int)ranges.size();
                for (int sidx = 0; sidx < n_parts; ++sidx) {
                    DagTask t{};
                    t.type = DAG_PART | flag;
                    t.b = (unsigned short)b;
                    t.q = (unsigned char)(q + 1);
                    t.j = (unsigned char)(q + 1);
                    t.S = (unsigned char)(chain ? sidx : 0);
                    t.pa = (unsigned char)ranges[sidx].first;
                    t.pb = (unsigned char)ranges[sidx].second;
                    t.slot = chain ? slot0 + (unsigned int)(sidx & 1) : plan.n_slots++;
                    t.ctr = ctr;
                    plan.tasks.push_back(t);
                }
                if (chain) plan.n_slots = slot0 + (n_parts > 1 ? 2 : 1);
                DagTask fin{};
                fin.type = DAG_DIAG | flag;
                if (chain) fin.type |= DAG_WAITNEXT;
                if (chain && q + 2 < Ps[b] && !(scheme == 2 && q + 1 >= q_f)) fin.type |= DAG_FUSED;
                // second level of following: the strip solve of tile (q, q+1) follows the factorisation of block q
                // (q >= 2) and this task follows IT -- DAG_NOSOLVE here, DAG_FUSED on that strip solve (step 3 below)
                if (xlink(q)) fin.type |= DAG_NOSOLVE;
                fin.b = (unsigned short)b;
                fin.q = fin.j = (unsigned char)(q + 1);
                fin.S = (unsigned char)(n_parts + 1);
                fin.pa = (unsigned char)(two && chain && q >= 2 ? q - 1 : q);
                fin.pb = (unsigned char)(q + 1);
                fin.slot = chain ? slot0 + (unsigned int)((n_parts - 1) & 1) : slot0;
                fin.ctr = ctr;
                if (cont0) {
                    fin.type |= DAG_NOSOLVE;          // owned by the strip solve of tile (q, q+1): step 3
                    owned_final[b] = fin;
                    has_owned[b] = 1;
                } else {
                    early_final[q + 1].push_back(fin);
                }
            }
        }
        // 3. off-diagonal tiles of this row
        for (int b : mats)
            for (int j = q + 1; j < Ps[b] + Mt && q < Ps[b]; ++j) {
                const bool owner = cont0 && j == q + 1 && q + 1 < Ps[b];     // its workgroup goes on with DIAG(q+1)
                dag_emit(plan, DAG_OFF, b, q, j, 0, q, s_off(b), scheme,
                         following ? (unsigned char)(DAG_WAITNEXT | (xlink(q) ? DAG_FUSED : 0) |
                                                     ((j == q + 1 && q + 1 < Ps[b]) ? DAG_NOSOLVE : 0))
                                   : (unsigned char)(((j == q + 1 && fused(b)) ? DAG_NOSOLVE : 0) | (owner ? DAG_FUSED : 0)),
                         following ? dag_final_panels() : 1);
                if (owner) {
                    if (q == 0) {
                        dag_emit(plan, DAG_DIAG, b, 1, 1, 0, 1, 1, scheme, DAG_NOSOLVE);   // DIAG(1): one final over [0, 1)
                    } else {
                        plan.tasks.push_back(owned_final[b]);
                        has_owned[b] = 0;
                    }
                }
            }
    }
    if (Ms > 0)
        for (int b : mats) dag_emit_schur(plan, b, Ps[b], Ms);
}


// The two hand-out orders of a latency-scheme list (DagPool): per queue the finals in list order, then the PARTs in list
// order; for a final with a chain the position of the chain's last part.
inline void dag_build_pool(DagPlan& plan)
{
    const size_t n = plan.tasks.size();
    plan.order.assign(n, 0u);
    plan.dep.assign(n, DAG_POOL_NONE);
    for (int g = 0; g < DAG_QUEUES; ++g) {
        const unsigned int lo = plan.queues.first[g], hi = plan.queues.first[g + 1];
        unsigned int pos = lo;
        for (unsigned int t = lo; t < hi; ++t)
            if ((plan.tasks[t].type & DAG_TYPE_MASK) != DAG_PART) plan.order[pos++] = t;
        plan.n_main[g] = pos - lo;
        std::vector<unsigned int> last_part(plan.n_ctrs + 1, DAG_POOL_NONE);      // per arrival counter: its last part's position
        for (unsigned int t = lo; t < hi; ++t)
            if ((plan.tasks[t].type & DAG_TYPE_MASK) == DAG_PART) {
                // a chained part adds its predecessor's running sum at the END of its own update: it may start while the
                // predecessor still runs -- but only once the predecessor has been TAKEN (dep[] of a pool entry)
                if ((plan.tasks[t].type & DAG_CHAIN) && plan.tasks[t].S > 0) plan.dep[pos] = last_part[plan.tasks[t].ctr];
                last_part[plan.tasks[t].ctr] = pos;        // (a chain's parts are in list order: the last one wins)
                plan.order[pos++] = t;
            }
        for (unsigned int m = lo; m < lo + plan.n_main[g]; ++m) {
            const DagTask& f = plan.tasks[plan.order[m]];
            if (f.S > 1) plan.dep[m] = last_part[f.ctr];   // S - 1 parts
        }
    }
}

// scheme: 0 throughput, 1 latency (dag_emit), -1 automatic: latency while the row-to-row dependency chain,
// not the MFMA work, bounds the run time -- i.e. while a queue has too few block rows in flight to hide the
// wait for the row above.  Measured on MI355X (tools/scheme_table.py; N = 2000 .. 8192, B = 1 .. 32, round 2,
// with the critical path fused into the diagonal tasks): the latency scheme wins or ties while the block
// rows of the matrices of the fullest queue add up to at most ~150 (N = 6000: B = 1: 12.7 -> 4.8 ms,
// B = 8: 17.6 -> 12.9 ms, B = 24: 31.1 -> 30.8 ms; N = 2000, B = 32: 3.9 -> 3.1 ms), or when no queue holds
// more than one matrix; throughput beyond (N = 6000, B = 32: 39.2 vs 39.9 ms; N = 8192, B = 32: 95.0 vs 96.1).
// (Readiness ordering was also tried for the throughput scheme: 800 -> 776 evals/s, not adopted.)
// (Round 6, after the round-5 kernel work: tools/latency_quick.py under PSOAP_DAG_SCHEME=0 / 1, ms per batch, scheme 1 / 0 --
//   N = 6000: 10 matrices 13.03 / 13.50, 12: 15.31 / 15.48, 16: 19.94 / 19.47, 24: 29.14 / 28.37; N = 8192: 12: 36.21 / 35.81, 16:
//   47.86 / 46.81; N = 4096: 20: 8.49 / 8.67, 24: 9.91 / 9.94, 28: 11.41 / 11.35, 32: 12.98 / 12.61; N = 2000: 32: 2.44 / 2.78
// -- the throughput scheme is ahead from about 740 block rows in the launch on (was 1200): 92 per queue.)
constexpr int DAG_LATENCY_QUEUE_ROWS = 92;
// How many of the 8 ticket queues a batch uses (matrix b goes to queue b mod that number; the workgroups of an XCD whose own
// queue is empty spread evenly over the queues in use -- k_chol_dag's steal0).  A queue's matrices share its workgroups, so a
// batch is through when its FULLEST queue is: 12 matrices on 8 queues are 2 + 1 per queue and cost what 16 do, 9 cost
// what 16 do.  Round 4 (tools/queue_sweep.py, profiles/r4_queue_sweep.txt; N = 6000, ms per batch with 8 / 4 / 2 / 1 queues):
//    9 matrices 17.9 / 14.2 / 13.1 / 12.6     12: 20.1 / 16.0 / 16.0 / --      13: 20.3 / 19.4 / 17.7 / 17.4
//   17 matrices 26.8 / 24.0 / 22.2 / 21.8     25: 33.4 / 33.4 / 32.2 / 31.3    16, 24, 32: the same within 0.5 % (8 ahead)
// i.e. time ~ ceil(B / n) x n, with all XCDs drawing from ONE in-order list costing about 1 % (every matrix in front of
// all eight L2s).  The rule: the n in {8, 4, 2, 1} with the smallest ceil(B / n) x n, the larger n on a tie.  Up to 8
// matrices keep a queue each (the XCDs without one steal; one shared queue measures the same).
inline int dag_queue_count(int B)
{
    if (const char* e = getenv("PSOAP_DAG_QUEUES"))      // experiments
        if (atoi(e) > 0) return atoi(e) < DAG_QUEUES ? atoi(e) : DAG_QUEUES;
    if (B <= DAG_QUEUES) return DAG_QUEUES;
    int best = DAG_QUEUES, best_cost = (B + DAG_QUEUES - 1) / DAG_QUEUES * DAG_QUEUES;
    for (int n = DAG_QUEUES / 2; n >= 1; n /= 2) {
        const int cost = (B + n - 1) / n * n;
        if (cost < best_cost) {
            best = n;
            best_cost = cost;
        }
    }
    return best;
}
constexpr int DAG_FOLLOW_MAX_MATS = 8;
constexpr int DAG_FOLLOW_SMALL_ROWS = 20;
inline int dag_auto_scheme(const std::vector<int>& Ps)
{
    long long rows[DAG_QUEUES] = {};
    int count[DAG_QUEUES] = {};
    const int nq = dag_queue_count((int)Ps.size());
    for (size_t b = 0; b < Ps.size(); ++b) {
        rows[b % nq] += Ps[b];
        ++count[b % nq];
    }
    long long max_rows = 0;
    int max_count = 0;
    for (int g = 0; g < DAG_QUEUES; ++g) {
        max_rows = rows[g] > max_rows ? rows[g] : max_rows;
        max_count = count[g] > max_count ? count[g] : max_count;
    }
    // (block rows per XCD: a queue shared by 8 / nq XCDs works its rows off that much faster)
    const int latency = (max_rows * nq <= (long long)DAG_LATENCY_QUEUE_ROWS * DAG_QUEUES || max_count <= 1) ? 1 : 0;
#ifdef PSOAP_FOLLOW
    // following strip solves (scheme 2) where they were measured to win (profiles/r3_follow_table.txt: N = 2000 .. 8192, B =
    // 1 .. 32, against scheme 1 with its PARTs just in time): up to eight matrices everywhere -- single evaluations 12-37 %
    // faster, eight matrices 1-21 % -- and up to 24 small ones (at most 20 block rows: N = 2000, 12 / 16 / 24 matrices 12 /
    // 10 / 3 % faster; from N = 4096 on scheme 1 is 1-2 % ahead at 12 and 16)
    int Pmax = 0;
    for (int P : Ps) Pmax = P > Pmax ? P : Pmax;
    if (latency == 1 && (Ps.size() <= (size_t)DAG_FOLLOW_MAX_MATS || (Ps.size() <= 24 && Pmax <= DAG_FOLLOW_SMALL_ROWS))) return 2;
#endif
    return latency;
}
// fixed_share > 0: the fixed plan (dag_fixed_plan) -- scheme 0, every matrix cut as ONE matrix on `fixed_share` workgroups
inline DagPlan dag_build_tasks(const std::vector<int>& Ps, int workers, int scheme = -1, int Mt = 0, int Ms = 0,
                               int fixed_share = 0)
{
    DagPlan plan;
    const int B = (int)Ps.size();
    if (fixed_share > 0) scheme = 0;
    if (scheme < 0) scheme = dag_auto_scheme(Ps);
#ifndef PSOAP_FOLLOW
    if (scheme == 2) scheme = 1;       // the following scheme needs the kernels built with -DPSOAP_FOLLOW
#endif
    plan.scheme = scheme;
    // workgroups of XCDs whose own queue is empty steal, so the workers are shared by the queues in use
    const int nq = dag_queue_count(B);
    const int used = B < nq ? (B > 0 ? B : 1) : nq;
    const int per_queue = workers / used > 0 ? workers / used : 1;
    for (int g = 0; g < DAG_QUEUES; ++g) {
        plan.queues.first[g] = (unsigned int)plan.tasks.size();
        std::vector<int> mats;
        if (g < nq)
            for (int b = g; b < B; b += nq) mats.push_back(b);
        dag_build_queue(plan, mats, Ps, per_queue, (B + nq - 1) / nq, scheme, Mt, Ms, fixed_share);
        if (scheme >= 1) {
            // Latency scheme: hand the tasks out in order of READINESS instead of block row by block row.
            // A task over panels [pa, pb) can run once block row pb-1 is finished ("stage" pb); within a
            // stage the diagonal final (the in-block Cholesky everybody waits for) comes first, then the
            // row's other finals, then the PARTs that just became ready, nearest block row first.  PARTs of
            // far-away rows thus run as soon as their panels exist instead of arriving in a burst when
            // their row comes up, and a worker rarely takes a ticket it then has to spin on.  Every wait
            // still targets a smaller ticket: a chain's PARTs have increasing pb, its final the largest.
            // (the update-only task of tile (q, q+1), which DIAG(q) waits for after its factorisation, goes in
            // front of it: every wait still targets a smaller ticket)
            auto cls = [](const DagTask& t) {
                const int ty = t.type & DAG_TYPE_MASK;
                if (ty == DAG_OFF && (t.type & DAG_NOSOLVE) && !(t.type & DAG_WAITNEXT)) return -1;   // update-only: in front of its DIAG
                if (ty == DAG_PART && t.pb == 0 && t.q == t.j && t.q <= 1) return -2;  // the running sum DIAG(0) / DIAG(1) start from
                return ty == DAG_DIAG ? 0 : (ty == DAG_OFF ? 1 : 2);      // PART and DAG_SCHUR: whatever is left of a stage
            };
            // (scheme 2, PSOAP_DAG_EARLY=1: a final that covers two panels -- a following strip solve or the diagonal task that
            // follows one, from block row 4 on -- starts with the older panel, i.e. could be picked up a stage early.  That
            // paid while the PARTs were in readiness order (the finals queued behind a whole round of them, 110 us at
            // N = 6000, q = 8); with the PARTs handed out just in time -- below -- it only makes the finals hold their
            // workgroups longer: N = 6000: 2.68 -> 2.59 ms for one evaluation, 6.46 -> 6.17 for four WITHOUT it.  Off.)
            // (latency schemes, PARTs: not before block row q - jit is the current one.  In pure readiness order the early stages
            // hold every far row's first PARTs -- ~300 tasks per stage at N = 6000 against ~50 at the end -- and the
            // finals of the next rows queue up behind them: 90 us per block row over the first third of the matrix
            // instead of 45.  Just in time, every stage holds about one block row's worth of PARTs.)
            // (scheme 1 as well -- it is what 17 .. 32 matrices of N <= 4096 and 17 .. 24 of N = 6000 get: 1-5 % there;
            // PSOAP_DAG_JIT1=0 keeps its PARTs in readiness order)
            static const bool jit1 = !(getenv("PSOAP_DAG_JIT1") && getenv("PSOAP_DAG_JIT1")[0] == '0');
            const int jit = (scheme == 2 || (scheme == 1 && jit1)) ? dag_jit_rows() : 0;
            // (the PARTs of the Schur tiles of predict, rows q >= P, keep their place: they are the filler work)
            auto jit_part = [jit, &Ps](const DagTask& t) {
                return jit > 0 && (t.type & DAG_TYPE_MASK) == DAG_PART && (t.type & DAG_CHAIN) && (int)t.q < Ps[t.b];
            };
            // (measured: 6 block rows ahead; 4 .. 12 within 2 %, a lead that grows with the row index 5-8 % worse)
            // (... and a chain's parts one after the other over the stages of that lead, not all in its first one: a part
            // waits for its predecessor, and a workgroup that holds a waiting part works on nothing else)
            std::vector<int> chain_parts(plan.n_ctrs + 1, 1);
            if (jit > 0)
                for (size_t i = plan.queues.first[g]; i < plan.tasks.size(); ++i) {
                    const DagTask& t = plan.tasks[i];
                    if ((t.type & DAG_TYPE_MASK) != DAG_PART && (t.type & DAG_CHAIN) && t.S > 1) chain_parts[t.ctr] = t.S - 1;
                }
            auto stage = [jit, &jit_part, &chain_parts](const DagTask& t) {
                const int ty = t.type & DAG_TYPE_MASK;
                if (jit_part(t)) {
                    const int spread = jit > 2 ? (int)t.S * (jit - 2) / chain_parts[t.ctr] : 0;
                    return std::max((int)t.pb, (int)t.q - jit + spread);
                }
                const bool follows = (ty == DAG_OFF && (t.type & DAG_WAITNEXT)) || (ty == DAG_DIAG && (t.type & DAG_NOSOLVE));
                static const bool early = getenv("PSOAP_DAG_EARLY") && getenv("PSOAP_DAG_EARLY")[0] == '1';   // experiments
                return (early && follows && t.q >= 4 && t.pb - t.pa >= 2) ? t.pb - 1 : (int)t.pb;   // (q >= 4: what it follows is a stage early too)
            };
            // (the finals of a stage row by row -- only scheme 2 has finals of two rows in one stage, and those of the
            // lower row follow those of the upper one)
            std::stable_sort(plan.tasks.begin() + plan.queues.first[g], plan.tasks.end(),
                             [&](const DagTask& a, const DagTask& b) {
                                 if (stage(a) != stage(b)) return stage(a) < stage(b);
                                 const int ca = cls(a), cb = cls(b);
                                 if ((ca == 2) != (cb == 2)) return cb == 2;     // (finals behind the PARTs of their stage: 3-6 % slower)
                                 // (tried: within a stage the PARTs that wait for nothing ahead of the ones that need the row
                                 // just finishing -- 3-5 % slower: those are the chains of the nearest rows)
                                 if (a.q != b.q) return a.q < b.q;
                                 if (ca != cb) return ca < cb;
                                 // (just in time: the chains of a row's tiles side by side -- first parts, second parts, ...
                                 // -- not tile after tile: a part waits for its predecessor)
                                 if (jit_part(a) && jit_part(b) && a.S != b.S) return a.S < b.S;
                                 return false;
                             });
        }
    }
    plan.queues.first[DAG_QUEUES] = (unsigned int)plan.tasks.size();
    plan.queues.follow_first = (unsigned int)dag_follow_first_row();
#ifdef PSOAP_POOL
    if (scheme >= 1) dag_build_pool(plan);
#endif
    return plan;
}

// How many persistent workgroups a batch gets: all the device admits (two per compute unit), or ONE per compute
// unit when the batch is bound by the row-to-row chains of its matrices and not by MFMA throughput.  The
// in-block factorisation on a chain is VALU / LDS work in dependent steps; an MFMA-streaming workgroup on the
// same compute unit owns the double-precision pipe for 64 cycles per instruction and slows it up to 2x
// (priorities only decide who issues next).  With one workgroup per compute unit the chain runs undisturbed:
// N = 6000: 4.1 -> 3.5 ms for one evaluation, N = 2000, B = 8: 1.40 -> 1.27 ms -- but the device then
// delivers roughly half the throughput, so the switch is by work: measured over N = 2000 .. 8192, B = 1 .. 32
// (tools/latency_quick.py under PSOAP_DAG_WORKERS), one per compute unit wins while
//     algorithmic flops of the batch  <=  3.3e9 x block rows of its largest matrix
// (chain time ~ 75 us per block row against ~35 TFLOP/s of the half-populated device), i.e. B N^2 <~ 7.7e7.
// Round 3 (following scheme): several matrices -- 2.0e9 instead of 3.3e9: their strip solves hold workgroups while they
// follow the factorisation, which a second workgroup per compute unit makes up for earlier (N = 6000, B = 2: 4.29 -> 3.99
// ms; N = 4096, B = 4: 2.87 -> 2.70 ms; N = 4096, B = 2 stays with one: 1.92 against 2.15 ms).
// Round 4: one workgroup per compute unit now means the kernels compiled for one wave per SIMD (512 registers per lane,
// nothing of the chain phases in scratch memory: 3-7 % faster), which moves the crossover for several matrices back to
// 3.3e9: N = 4096, 3 / 4 matrices 2.39 -> 2.16 / 2.74 -> 2.62 ms, N = 6000, 2 matrices 3.94 -> 3.83 ms; beyond (N = 4096:
// 6, N = 6000: 3, N = 8192: 2 matrices) two per compute unit stay 5-12 % ahead.
inline int dag_pick_workers(double flops, int Pmax, int compute_units, int max_workers, int n_mats = 1)
{
    if (const char* e = getenv("PSOAP_DAG_WORKERS"))      // experiments
        if (atoi(e) > 0) return atoi(e);
    if (max_workers <= compute_units) return max_workers;
    (void)n_mats;
    return flops <= 3.3e9 * (double)Pmax ? compute_units : max_workers;
}
inline double dag_batch_flops(const std::vector<int>& Ps, int Mt = 0)
{
    double f = 0.0;
    for (int P : Ps) {
        const double n = 128.0 * P, r = 128.0 * Mt;
        f += n * n * n / 3.0 + n * n * r;
    }
    return f;
}

// The task list every lane of a stream runs (one matrix), cut as if `lanes` matrices shared `workers` workgroups, with its
// bursts marked (DagTask::b, which the lanes do not need: the matrix index is the lane).  A burst is what the workgroups of
// an XCD draw from one lane before they move on to the next: one block row -- scheme 0: the tasks emitted for row q (the
// PARTs that pre-accumulate DIAG(q+1), the row's strip solves with their PARTs, the owned DIAG(q+1)); schemes 1, 2:
// whatever precedes a diagonal final.  bursts == false: every ticket ends one (the lanes ticket by ticket in turn).
// The list depends on P and the scheme ONLY -- the split factors are those of a nominal 32 lanes whatever the stream's
// lane count -- so a proposal's result is bit-identical for every lane count, batch size, submission order and world size.
// Scheme 0 splits half as eagerly as a plain launch does (a row's tiles are cut while `tiles x parts` stays below HALF the
// workgroups' share of one lane): with other matrices in other phases always in flight, sparse block
// rows need not fill the device by themselves, and every part saved is a partial tile that does not travel (measured:
// 38.5 -> 38.2 ms per 32-walker step).
inline DagPlan dag_build_lane_plan(int P, int lanes, int workers, int scheme, bool bursts = true)
{
    (void)lanes;
    // (scheme 0: dag_nominal_share -- half the workgroups' share of one of 32 lanes: the fixed plan, also what a batch
    // launch gets under PSOAP_FIXED_PLAN=1)
    const int share15 = workers / STREAM_NOMINAL_LANES > 0 ? workers / STREAM_NOMINAL_LANES : 1;
    DagPlan plan = scheme == 0 ? dag_build_tasks(std::vector<int>(1, P), share15, 0, 0, 0, dag_nominal_share(workers))
                               : dag_build_tasks(std::vector<int>(1, P), share15, scheme);
    const bool rows = plan.scheme == 0 && DAG_TILE_DEPS;
    // (schemes 1, 2: the list is in order of readiness already and its tasks are short -- the lanes ticket by ticket in turn
    // measured 8 % faster at N = 2000, the same at N = 4096)
    if (plan.scheme != 0) bursts = false;
    auto section = [](const DagTask& t) { return (t.q == t.j && t.q > 0) ? (int)t.q - 1 : (int)t.q; };
    for (size_t i = 0; i < plan.tasks.size(); ++i) {
        const bool last = i + 1 == plan.tasks.size();
        bool end = !bursts || last;
        if (!end) {
            const DagTask& nx = plan.tasks[i + 1];
            end = rows ? section(nx) != section(plan.tasks[i]) : (nx.type & DAG_TYPE_MASK) == DAG_DIAG;
        }
        plan.tasks[i].b = (unsigned short)(end ? STREAM_BURST_END : 0);
    }
    return plan;
}

// Waves per SIMD of the throughput-scheme kernels (k_chol_dag<C, false, false, *>).  -DPSOAP_WPE3 (round 5, experiment): three
// workgroups per compute unit, 168 registers per lane -- see DESIGN.md / LABNOTES.md for what it measured.
#ifdef PSOAP_WPE3
constexpr int DAG_WPE_TP = 3;
#else
constexpr int DAG_WPE_TP = 2;
#endif

// uniform batch: B matrices of P block rows each
inline DagPlan dag_build_tasks(int B, int P, int workers, int scheme = -1, int Mt = 0, int Ms = 0)
{
    return dag_build_tasks(std::vector<int>((size_t)(B > 0 ? B : 0), P), workers, scheme, Mt, Ms);
}


}  // namespace psoap
