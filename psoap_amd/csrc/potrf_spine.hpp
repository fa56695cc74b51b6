// potrf_spine.hpp -- the diagonal task of the latency scheme: form the 128 x 128 diagonal tile
//     T = part - strip^T strip
// in registers and factor it, with the serial pivot chain on a wave of its own.
//
// Same mathematics and outputs as potrf_blocked<.., FUSED = true> (U11 into the matrix, W = U11^-T k-major
// into Wm, z = W r into r, sum log U_ii and z^T z into the accumulators), different division of labour.
// In potrf_blocked every wave owns two blocks of every block row, so the in-wave factorisation of the next
// diagonal block (16 dependent pivots, 2.6 us -- 40 % of the eight-step loop) can only start when wave 0 is
// through with its share of the trailing update, and the other three waves wait for it.  Here
//
//   wave 0 ("spine")  owns the eight diagonal blocks and does nothing else: right
//                     after block row bb is published it updates diagonal block bb+1 with it (one 16 x 16 x 16
//                     product), factors it, parks W^T for the others and raises an LDS flag; the remaining
//                     diagonal updates with row bb follow while the workers finish row bb+1;
//   waves 1..3        own the 56 off-diagonal blocks and the 8 right-hand-side blocks (block (I, J) belongs to
//                     wave 1 + (I + J) mod 3: 21 / 22 / 21 blocks, register slots fixed at compile time per wave): trailing update with row bb (phase C),
//                     then -- behind the flag -- block row bb+1 = W_(bb+1) x (their blocks of that row)
//                     (phase B), one workgroup barrier per step.
//
// The published block row lives in one of two LDS buffers (step parity): a worker already in phase B of
// step bb+1 must not overwrite what the spine still reads for its deferred updates with row bb.
#pragma once
#include "potrf_blocked.hpp"

// variant matrix (tools/lat_variants.py): the routine's global accesses as global_* instead of flat_* instructions

namespace psoap {
namespace ps {

using pb::BLK;
constexpr int OFF_ROW = 0;              // 2 x 9 blocks: row bb of U (J > bb) / W (J <= bb) / z (slot 8), buffer bb & 1
constexpr int OFF_V = 18 * BLK;         // W_bb^T
static_assert(OFF_V + BLK <= pb::OFF_TR, "spine layout must stay below the transpose scratch it shares with potrf_blocked");
// progress flag (diagonal blocks factored and announced so far): one int in the routine's LDS scratch, pb::OFF_FLAG --
// not a __shared__ variable of its own, so that the routine can run inside a non-kernel function (gemm_core.hpp, SmemArg)
template <class SM>
__device__ __forceinline__ lds_int* flag_ptr(SM sm) { return (lds_int*)sm.ptr(pb::OFF_FLAG); }

__device__ __forceinline__ int row_off(int bb) { return OFF_ROW + (bb & 1) * 9 * BLK; }
// Wave W (1..3) owns block (I, J), J != I, when (I + J) mod 3 == W - 1; column 8 is the right-hand side
// (r_k in its first column: it turns into z exactly like a block of U).  In block row I these are the
// columns J = j0 + 3 s, s = 0..2, with j0 = (W - 1 - I) mod 3 -- register slot 3 I + s.  Slots whose column
// falls outside 0..8 or on the diagonal are never touched (no register is spent on them): 21 / 22 / 21 blocks.
constexpr int col0(int W, int I) { return ((W - 1 - I) % 3 + 3) % 3; }
constexpr bool owned(int W, int I, int s) { return col0(W, I) + 3 * s <= 8 && col0(W, I) + 3 * s != I; }

template <class SM>
__device__ __forceinline__ void wait_flag(int target, int lane, SM sm)
{
    if (lane == 0)
        while (__hip_atomic_load(flag_ptr(sm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// Workgroup barrier for LDS traffic only: waits for this wave's LDS operations, not for its global stores
// (__syncthreads() also drains vmcnt: with the block-row outputs stored inside the step loop every barrier
// would wait for the memory round trip of those stores).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ d4 neg(const d4& v) { return d4{-v[0], -v[1], -v[2], -v[3]}; }

// Where the fused diagonal task publishes its progress for the strip solves that follow it (dag_kernel.hpp: mailbox,
// dag_pss).  mb == nullptr: nothing is published.
constexpr int MB_BLOCKS = 10;      // per step b: U_bJ (J = 0..7, written for J > b), V_b = W_bb^T (8), the rhs block z_b (9)
struct SpinePub {
    double* mb;        // this block's half of the matrix's mailbox: slot (b, J) at ((b * MB_BLOCKS) + J) * 256
    int* step_w;       // MatFlags::step_w
    int base;          // 8 q
};
// The second level of following: the strip above this diagonal tile -- tile (q-1, q) -- is being solved by a task that
// itself follows the factorisation of block q-1 and stores its row blocks one by one (dag_pss, `xpub`): the symmetric
// update here consumes them as they arrive (16 rows = one LDS stage of the update), and this task also applies that
// tile's contribution to its own right-hand side block (z of block q-1 comes step by step from the mailbox of the
// factorisation above).  xstep == nullptr: the strip is final when wait_dep() returns (every other use).
struct SpineFollow {
    const double* zmb;     // the mailbox half of block q-1
    int* xstep;            // MatFlags::xcol[q]: 8 (q-1) + b + 1 once row block b of tile (q-1, q) is in memory (two waves)
    int base;              // 8 (q-1)
    unsigned int* err;     // DagCtl::error (a wait that gives up)
    const double* strip0;  // != nullptr: the final covers TWO panels -- tile (q-2, q), final when wait_dep() returns, is
                           // applied first (eight plain stages), then the following stages of tile (q-1, q)
};
// row blocks delivered so far of the tile(s) whose two-flag records are xa and xb (the same record twice for one tile)
__device__ __forceinline__ int x_steps(int* xa, int* xb, int base, int lane)
{
    int x = 0x7fffffff;
    if (lane < 4) x = poll_word((lane < 2 ? xa : xb) + (lane & 1));
    x = min(x, __shfl_xor(x, 1, 64));
    x = min(x, __shfl_xor(x, 2, 64));
    return __builtin_amdgcn_readfirstlane(x) - base;
}
__device__ __forceinline__ int x_wait(int* xa, int* xb, int base, unsigned int* err, int need, int lane)
{
    int v = x_steps(xa, xb, base, lane);
    for (long long spins = 0; v < need; ++spins) {
        if (spins > 4000000) {            // seconds: give up, the results are invalid and reported as such
            if (lane == 0) __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return 8;
        }
        __builtin_amdgcn_s_sleep(1);
        v = x_steps(xa, xb, base, lane);
    }
    return v;
}
__device__ __forceinline__ int x_steps(const SpineFollow& xf, int lane) { return x_steps(xf.xstep, xf.xstep, xf.base, lane); }
__device__ __forceinline__ int x_wait(const SpineFollow& xf, int need, int lane)
{
    return x_wait(xf.xstep, xf.xstep, xf.base, xf.err, need, lane);
}
__device__ __forceinline__ d4 x_zblock(const SpineFollow& xf, int b, int lane)
{
    const double* src = xf.zmb + ((size_t)b * MB_BLOCKS + 9) * 256;
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = src[r * 64 + lane];
    return v;
}
// one block, accumulator-linear, with agent-scope (write-through) stores: no L2 write-back is needed before the flag
__device__ __forceinline__ void pub_block(const SpinePub& pub, int b, int J, int lane, const d4& v)
{
    double* dst = pub.mb + ((size_t)b * MB_BLOCKS + J) * 256;
#pragma unroll
    for (int r = 0; r < 4; ++r) __hip_atomic_store(dst + r * 64 + lane, v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// This wave's share of steps < n_steps is in the mailbox.  `younger`: how many of its vector-memory operations were
// issued AFTER the stores the flag vouches for (they may still be in flight: stores complete in issue order, and these
// waves issue nothing but stores inside the step loop).  Waiting for everything instead stalled every wave of the
// factorisation by ~1.3 us per step -- the write-through stores take longer than a step to be acknowledged.
// The count is exact by construction: every operation between the vouched-for stores and the wait is a relaxed ATOMIC
// store (pub_block; pb::emit_w<.., COUNTED>), four per block, which the compiler can neither merge nor drop -- a wait
// for `younger` outstanding operations would otherwise let vouched-for stores stay in flight if fewer had been emitted.
// -DPSOAP_PUB_WAIT_ALL: wait for everything (the knob to rule the counting out when a result is in doubt).
__device__ __forceinline__ void pub_flag(const SpinePub& pub, int wave, int n_steps, int lane, int younger = 0)
{
#ifdef PSOAP_PUB_WAIT_ALL
    younger = 0;
#endif
    switch (younger) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
    if (lane == 0) __hip_atomic_store(pub.step_w + wave, pub.base + n_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// blocks of row I that worker wave W publishes (its owned blocks right of the diagonal, the right-hand side included)
constexpr int pub_count(int W, int I)
{
    int n = 0;
    for (int s = 0; s < 3; ++s) {
        const int J = ((W - 1 - I) % 3 + 3) % 3 + 3 * s;
        if (J <= 8 && J != I && J > I) ++n;
    }
    return n;
}

// ---- the three worker waves ------------------------------------------------------------------------------
template <int W, class WaitFn, class SM>
__device__ __forceinline__ void worker(double* Km, int ld, int k0, double* Wm, double* Rv,
                                       const double* __restrict__ part, const double* __restrict__ strip,
                                       WaitFn& wait_dep, SM sm, const SpinePub& pub, const SpineFollow& xf)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = lane >> 4, c = lane & 15;
    d4 blk[24];
#pragma unroll
    for (int I = 0; I < 8; ++I)
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (owned(W, I, s)) {
                const int J = col0(W, I) + 3 * s;
                d4 v = {0.0, 0.0, 0.0, 0.0};
                if (J > I && J < 8) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = part[(size_t)(16 * I + q + 4 * r) * NB + 16 * J + c];
                }
                blk[3 * I + s] = v;
            }
    wait_dep();
    // the right-hand side blocks (final only now: the task above updated them last)
#pragma unroll
    for (int I = 0; I < 8; ++I)
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (owned(W, I, s) && col0(W, I) + 3 * s == 8) {
#pragma unroll
                for (int r = 0; r < 4; ++r) blk[3 * I + s][r] = (c == 0) ? Rv[k0 + 16 * I + q + 4 * r] : 0.0;
            }
    // ---- T -= strip^T strip on the upper blocks this wave owns (K = 128 in eight LDS stages)
    if (xf.xstep) {
        // ... behind the strip solve that is producing the strip (SpineFollow): a following stage holds one of its row
        // blocks.  A stage is requested ahead of the products of the one before when it is known to be there, behind
        // them otherwise; the right-hand side blocks take that tile's contribution here as well (r_I -= X_I^T z: one
        // more column).  With an older panel in front (strip0) its eight stages come first, unconditionally.
        const int fr = lane & 15, fk = lane >> 4;
        const int n0 = xf.strip0 ? NB / KB : 0, nch = n0 + NB / KB;
        int avail = 0;
        d4 zc = {0.0, 0.0, 0.0, 0.0}, zn = zc;
        if (n0) {
            stage_glds_one(xf.strip0, (size_t)ld, 0, 0, tid, sm);
        } else {
            avail = x_wait(xf, 1, lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            stage_glds_one(strip, (size_t)ld, 0, 0, tid, sm);
            zc = x_zblock(xf, 0, lane);
        }
        __syncthreads();
#pragma unroll 1
        for (int ch = 0; ch < nch; ++ch) {
            const int cur = ch & 1;
            const int need = ch + 2 - n0;          // steps of the strip solve the next stage needs (<= 0: none)
            bool early = false;
            if (ch + 1 < nch) {
                if (need <= 0) {
                    stage_glds_one(xf.strip0, (size_t)ld, (ch + 1) * KB, cur ^ 1, tid, sm);
                    early = true;
                } else {
                    if (avail < need) avail = x_steps(xf, lane);
                    if (avail >= need) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        stage_glds_one(strip, (size_t)ld, (need - 1) * KB, cur ^ 1, tid, sm);
                        zn = x_zblock(xf, need - 1, lane);
                        early = true;
                    }
                }
            }
            const int base = cur * LDS_BUFFER + fk * LDS_LD + fr;
            const bool with_rhs = ch >= n0;
#pragma unroll
            for (int I = 0; I < 8; ++I) {
                double x[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) x[ks] = -sm[base + ks * 4 * LDS_LD + 16 * I];
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    if (owned(W, I, s) && col0(W, I) + 3 * s > I) {
                        const int J = col0(W, I) + 3 * s;
                        if (J == 8 && !with_rhs) continue;
                        double y[4];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) y[ks] = J < 8 ? sm[base + ks * 4 * LDS_LD + 16 * J] : zc[ks];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks)
                            blk[3 * I + s] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ks], y[ks], blk[3 * I + s], 0, 0, 0);
                    }
            }
            if (ch + 1 < nch && !early) {
                avail = x_wait(xf, need, lane);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                stage_glds_one(strip, (size_t)ld, (need - 1) * KB, cur ^ 1, tid, sm);
                zn = x_zblock(xf, need - 1, lane);
            }
            zc = zn;
            __syncthreads();
        }
    } else if (strip) {
    stage_glds_one(strip, (size_t)ld, 0, 0, tid, sm);
    __syncthreads();
    {
        const int fr = lane & 15, fk = lane >> 4;
#pragma unroll 1
        for (int ch = 0; ch < NB / KB; ++ch) {
            const int cur = ch & 1;
            if (ch + 1 < NB / KB) stage_glds_one(strip, (size_t)ld, (ch + 1) * KB, cur ^ 1, tid, sm);
            const int base = cur * LDS_BUFFER + fk * LDS_LD + fr;
#pragma unroll
            for (int I = 0; I < 7; ++I) {
                double x[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) x[ks] = -sm[base + ks * 4 * LDS_LD + 16 * I];
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    if (owned(W, I, s) && col0(W, I) + 3 * s > I && col0(W, I) + 3 * s < 8) {
                        const int J = col0(W, I) + 3 * s;
                        double y[4];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) y[ks] = sm[base + ks * 4 * LDS_LD + 16 * J];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks)
                            blk[3 * I + s] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ks], y[ks], blk[3 * I + s], 0, 0, 0);
                    }
            }
            __syncthreads();
        }
    }
    }
    __syncthreads();   // [S0] the spine has reset the flag; the stage buffers are free for the block rows

#pragma unroll 1
    for (int bb = 0; bb < 8; ++bb) {
        // ---- B: block row bb (behind the flag of diagonal block bb).  The (up to three) products of the row
        // are issued k-step by k-step so that consecutive MFMAs are independent.
        wait_flag(bb + 1, lane, sm);
        {
            const int base = row_off(bb);
            const d4 x = pb::load_blk(OFF_V, lane, sm);
#pragma unroll
            for (int I = 0; I < 8; ++I) {
                if (I != bb) continue;
                d4 res[3];
#pragma unroll
                for (int s = 0; s < 3; ++s) res[s] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int s = 0; s < 3; ++s)
                        if (owned(W, I, s))
                            res[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ks], blk[3 * I + s][ks], res[s], 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    if (owned(W, I, s)) {
                        const int J = col0(W, I) + 3 * s;
                        pb::store_blk(base + J * BLK, lane, res[s], sm);
                        blk[3 * I + s] = res[s];
                        if (pub.mb && J > I) pub_block(pub, I, J < 8 ? J : 9, lane, res[s]);   // U_bJ (z_b) for the followers
                    }
            }
        }
        lds_barrier();     // [S1 + bb]
        // ---- C: trailing update of the rows below with row bb
        {
            const int base = row_off(bb);
#pragma unroll
            for (int I = 1; I < 8; ++I) {
                if (I <= bb) continue;
                const d4 xs = neg(pb::load_blk(base + I * BLK, lane, sm));           // -U_bI
                // upper blocks, A_IJ -= U_bI^T U_bJ: always active here (J > I > bb), interleaved like phase B
                d4 y[3];
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    if (owned(W, I, s) && col0(W, I) + 3 * s > I) y[s] = pb::load_blk(base + (col0(W, I) + 3 * s) * BLK, lane, sm);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int s = 0; s < 3; ++s)
                        if (owned(W, I, s) && col0(W, I) + 3 * s > I)
                            blk[3 * I + s] = __builtin_amdgcn_mfma_f64_16x16x4f64(xs[ks], y[s][ks], blk[3 * I + s], 0, 0, 0);
                // lower blocks, G_IJ -= U_bI^T W_bJ: active from step J on
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    if (owned(W, I, s) && col0(W, I) + 3 * s < I) {
                        const int J = col0(W, I) + 3 * s;
                        if (J <= bb) blk[3 * I + s] = pb::mma16(xs, pb::load_blk(base + J * BLK, lane, sm), blk[3 * I + s]);
                    }
            }
        }
        // this wave's blocks of rows < bb are in the mailbox by now; younger than those: row bb's blocks (phase B above) and,
        // in wave 1, the four stores of W_(bb-1) at the end of the previous step
        if (pub.mb && bb > 0) pub_flag(pub, W, bb, lane, 4 * pub_count(W, bb) + (W == 1 ? 4 : 0));
        // W_bb itself (parked by the spine in the row buffer) goes out to memory from here, off the spine's chain
        if (W == 1) pb::emit_w<SM, true>(pb::load_blk(row_off(bb) + bb * BLK, lane, sm), bb, bb, lane, 1, Wm, sm);
    }
    if (pub.mb) pub_flag(pub, W, 8, lane);
    double zz = 0.0;
    // ---- outputs of this wave: its blocks of W (strictly lower), of U11 (strictly upper) and of z
#pragma unroll
    for (int I = 0; I < 8; ++I)
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (owned(W, I, s)) {
                const int J = col0(W, I) + 3 * s;
                if (J < I) {
                    pb::emit_w(blk[3 * I + s], I, J, lane, W, Wm, sm);
                } else if (J < 8) {
                    const d4& v = blk[3 * I + s];
#pragma unroll
                    for (int r = 0; r < 4; ++r) Km[(size_t)(k0 + 16 * I + q + 4 * r) * ld + k0 + 16 * J + c] = v[r];
                } else {
                    // z (column 0 of the rhs block) back into r, and its share of z^T z
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c == 0) {
                            const double z = blk[3 * I + s][r];
                            Rv[k0 + 16 * I + q + 4 * r] = z;
                            zz = fma(z, z, zz);
                        }
                }
            }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) zz += __shfl_xor(zz, off, 64);
    if (lane == 0) sm[pb::OFF_RED + W] = zz;
}

// ---- wave 0 --------------------------------------------------------------------------------------------
// factor diagonal block bb (d), park W_bb and W_bb^T (the X operand of W_bb Y) in LDS, announce
template <class SM>
__device__ __forceinline__ void spine_factor(d4& d, int bb, int lane, int& bad, double& logsum, SM sm, const SpinePub& pub)
{
    const int q = lane >> 4, c = lane & 15;
    d4 wdiag;
    pb::chol16(d, wdiag, lane, bad, sm);
    pb::store_blk(row_off(bb) + bb * BLK, lane, wdiag, sm);
#pragma unroll
    for (int r = 0; r < 4; ++r) sm[pb::OFF_TR + (q + 4 * r) * 17 + c] = wdiag[r];
    __builtin_amdgcn_wave_barrier();
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = sm[pb::OFF_TR + c * 17 + q + 4 * r];
    __builtin_amdgcn_wave_barrier();
    pb::store_blk(OFF_V, lane, v, sm);
    if (pub.mb) pub_block(pub, bb, 8, lane, v);                  // V_b = W_bb^T for the followers
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(flag_ptr(sm), bb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    double dg = 1.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) dg = (c == q + 4 * r) ? d[r] : dg;
    logsum += log(dg);
}

template <class WaitFn, class SM>
__device__ __forceinline__ void spine(double* Km, int ld, int k0, const double* __restrict__ part,
                                      const double* __restrict__ strip, WaitFn& wait_dep, unsigned long long* tl, SM sm,
                                      const SpinePub& pub, const SpineFollow& xf)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = lane >> 4, c = lane & 15;
    d4 d[8];
#pragma unroll
    for (int I = 0; I < 8; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[I][r] = part[(size_t)(16 * I + q + 4 * r) * NB + 16 * I + c];
    wait_dep();
    if (xf.xstep) {
        // the same stages as the workers', requested as the strip solve above delivers them (worker<W>)
        const int fr = lane & 15, fk = lane >> 4;
        const int n0 = xf.strip0 ? NB / KB : 0, nch = n0 + NB / KB;
        int avail = 0;
        if (n0) {
            stage_glds_one(xf.strip0, (size_t)ld, 0, 0, tid, sm);
        } else {
            avail = x_wait(xf, 1, lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            stage_glds_one(strip, (size_t)ld, 0, 0, tid, sm);
        }
        __syncthreads();
#pragma unroll 1
        for (int ch = 0; ch < nch; ++ch) {
            const int cur = ch & 1;
            const int need = ch + 2 - n0;
            bool early = false;
            if (ch + 1 < nch) {
                if (need <= 0) {
                    stage_glds_one(xf.strip0, (size_t)ld, (ch + 1) * KB, cur ^ 1, tid, sm);
                    early = true;
                } else {
                    if (avail < need) avail = x_steps(xf, lane);
                    if (avail >= need) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        stage_glds_one(strip, (size_t)ld, (need - 1) * KB, cur ^ 1, tid, sm);
                        early = true;
                    }
                }
            }
            const int base = cur * LDS_BUFFER + fk * LDS_LD + fr;
#pragma unroll
            for (int I = 0; I < 8; ++I) {
                double x[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) x[ks] = sm[base + ks * 4 * LDS_LD + 16 * I];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) d[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x[ks], x[ks], d[I], 0, 0, 0);
            }
            if (ch + 1 < nch && !early) {
                avail = x_wait(xf, need, lane);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                stage_glds_one(strip, (size_t)ld, (need - 1) * KB, cur ^ 1, tid, sm);
            }
            __syncthreads();
        }
    } else if (strip) {
        stage_glds_one(strip, (size_t)ld, 0, 0, tid, sm);
        __syncthreads();
        const int fr = lane & 15, fk = lane >> 4;
#pragma unroll 1
        for (int ch = 0; ch < NB / KB; ++ch) {
            const int cur = ch & 1;
            if (ch + 1 < NB / KB) stage_glds_one(strip, (size_t)ld, (ch + 1) * KB, cur ^ 1, tid, sm);
            const int base = cur * LDS_BUFFER + fk * LDS_LD + fr;
#pragma unroll
            for (int I = 0; I < 8; ++I) {
                double x[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) x[ks] = sm[base + ks * 4 * LDS_LD + 16 * I];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) d[I] = __builtin_amdgcn_mfma_f64_16x16x4f64(-x[ks], x[ks], d[I], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    if (tl && lane == 0) tl[5] = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) __hip_atomic_store(flag_ptr(sm), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();   // [S0]
    int bad = 0;
    double logsum = 0.0;
    spine_factor(d[0], 0, lane, bad, logsum, sm, pub);
#pragma unroll 1
    for (int bb = 0; bb < 7; ++bb) {
        lds_barrier();     // [S1 + bb] block row bb is published
        const int base = row_off(bb);
        // ---- the next diagonal block first: update with row bb, factor, announce
#pragma unroll
        for (int I = 1; I < 8; ++I)
            if (I == bb + 1) {
                const d4 xi = pb::load_blk(base + I * BLK, lane, sm);
                d[I] = pb::mma16(neg(xi), xi, d[I]);
                spine_factor(d[I], I, lane, bad, logsum, sm, pub);
            }
        // ---- deferred: the other diagonal blocks, while the workers finish row bb+1
#pragma unroll
        for (int I = 2; I < 8; ++I)
            if (I > bb + 1) {
                const d4 xi = pb::load_blk(base + I * BLK, lane, sm);
                d[I] = pb::mma16(neg(xi), xi, d[I]);
            }
        if (pub.mb) pub_flag(pub, 0, bb + 1, lane, 4);  // V_0 .. V_bb are published; only V_(bb+1), just issued, may be in flight
    }
    lds_barrier();         // [S1 + 7]
    if (pub.mb) pub_flag(pub, 0, 8, lane);
    if (tl && lane == 0) tl[6] = __builtin_amdgcn_s_memrealtime();
    // ---- outputs: the diagonal blocks of U11; log-determinant share and the not-positive-definite flag for
    // the caller (through LDS: the accumulators are updated after the closing barrier)
#pragma unroll
    for (int I = 0; I < 8; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) Km[(size_t)(k0 + 16 * I + q + 4 * r) * ld + k0 + 16 * I + c] = d[I][r];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) logsum += __shfl_xor(logsum, off, 64);
    if (lane == 0) {
        sm[pb::OFF_RED + 0] = logsum;
        sm[pb::OFF_RED + 4] = bad ? 1.0 : 0.0;
    }
}

}  // namespace ps

// All 256 threads call this.  part: the tile's running sum (row-major 128 x 128, read before wait_dep());
// strip: the tile above the diagonal (k-major, leading dimension ld), final once wait_dep() returns.
// wait_dep() is called by every thread exactly once (it may contain a workgroup barrier).
template <class WaitFn, class SM = SmemKernel>
__device__ __forceinline__ void potrf_spine_fused(double* Km, int ld, int k0, double* Wm, double* Rv, MatAcc* acc,
                                                  const double* __restrict__ part, const double* __restrict__ strip,
                                                  WaitFn wait_dep, unsigned long long* tl = nullptr, SM sm = SM(),
                                                  ps::SpinePub pub = ps::SpinePub{nullptr, nullptr, 0},
                                                  ps::SpineFollow xf = ps::SpineFollow{nullptr, nullptr, 0, nullptr, nullptr})
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave == 0) ps::spine(Km, ld, k0, part, strip, wait_dep, tl, sm, pub, xf);
    else if (wave == 1) ps::worker<1>(Km, ld, k0, Wm, Rv, part, strip, wait_dep, sm, pub, xf);
    else if (wave == 2) ps::worker<2>(Km, ld, k0, Wm, Rv, part, strip, wait_dep, sm, pub, xf);
    else ps::worker<3>(Km, ld, k0, Wm, Rv, part, strip, wait_dep, sm, pub, xf);
    __syncthreads();   // all outputs issued, the reductions are in LDS
    if (threadIdx.x == 0) {
        const double l = sm[pb::OFF_RED + 0];
        const double qd = (sm[pb::OFF_RED + 1] + sm[pb::OFF_RED + 2]) + sm[pb::OFF_RED + 3];
        // this block's record (common.hpp, MatAcc): written, never read here -- no chain from block row to block row
        acc_store(acc, k0 / NB, l, qd, sm[pb::OFF_RED + 4] != 0.0);
    }
    // (the LDS scratch goes back to the tile engine behind the caller's next barrier: dag_drain)
    if (tl && threadIdx.x == 0) tl[1] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace psoap
