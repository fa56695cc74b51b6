// potrf_blocked.hpp -- in-block Cholesky of one 128 x 128 diagonal tile by 256 threads (4 waves),
// blocked 16 x 16 with the inner products on fp64 MFMA.
//
// Produces, like the unblocked version it replaces (k_potrf_diag: 128 pivot barriers):
//   U11 (upper factor, back into the matrix), W = U11^-T (lower triangular) in the k-major form
//   Wt[e][i] = W[i][e] that the strip solve multiplies with, z = W r_k, sum(log U_ii), z^T z.
//
// Data layout: the tile is an 8 x 8 grid of 16 x 16 blocks plus a ninth block column whose first
// column is the right-hand side r_k; each block lives in ONE wave's registers in the MFMA
// accumulator layout (lane (q = l>>4, c = l&15), register r holds element (q + 4r, c)).  With both
// operands in that layout, four v_mfma_f64_16x16x4_f64 compute C += X^T Y straight from registers
// (k runs over the rows, permuted identically on both sides).  Block (I, J) belongs to wave
// (3I + J) & 3, so every block row is spread over all four waves and wave 0 owns the diagonal.
// Upper positions (J >= I) hold A -> U; strictly lower positions accumulate
// G_IJ = -sum_{m<I} U_mI^T W_mJ, from which W_IJ = W_II G_IJ; column 8 turns r_k into z exactly like
// a block of U (z_I = W_II (r_I - sum_{m<I} U_mI^T z_m)), so the solve costs no extra code.
//
// Step bb = 0..7 (two workgroup barriers each):
//   A  wave 0 factors the diagonal block inside the wave: 16 pivots on the augmented block
//      [A | I], one LDS broadcast line per pivot (the pivot itself travels by v_readlane so its
//      reciprocal square root overlaps the LDS round trip), no workgroup barrier -> U_bb,
//      W_bb = U_bb^-T; W_bb^T is parked in LDS for the other waves.
//   B  row bb is finished by all waves:  U_bJ = W_bb A_bJ (J > bb, incl. the rhs column),
//      W_bJ = W_bb G_bJ (J < bb); finished blocks are published in LDS (W goes out to memory after the loop).
//   C  trailing update by all waves:  A_IJ -= U_bI^T U_bJ (bb < I <= J),  G_IJ -= U_bI^T W_bJ (J <= bb < I).
#pragma once
#include "gemm_core.hpp"

namespace psoap {

// timing stamps of the ABLATE == 9 diagnostic build (microbench only): [wave*8 + step][6], waves 0 and 1
__device__ unsigned long long g_potrf_stamps[16 * 6];

namespace pb {
constexpr int BLK = 256;                     // doubles per 16 x 16 block
constexpr int OFF_UROW = 0;                  // 9 blocks: row bb of U incl. the rhs column (accumulator-linear: r*64 + lane)
constexpr int OFF_WROW = 9 * BLK;            // 8 blocks: row bb of W
constexpr int OFF_V = 17 * BLK;              // W_bb^T
constexpr int OFF_DUMMY = 18 * BLK;          // target of predicated-off publications
constexpr int OFF_TR = 19 * BLK;             // 4 x (16 x 17) transpose scratch, one per wave
constexpr int OFF_LINE = OFF_TR + 4 * 272;   // 32: broadcast line of the in-wave factorisation
constexpr int OFF_RED = OFF_LINE + 32;       // 8: reductions
constexpr int OFF_FLAG = OFF_RED + 6;         // one int: progress flag of potrf_spine (diagonal blocks announced)
constexpr int LDS_DOUBLES = OFF_RED + 8;
static_assert((size_t)LDS_DOUBLES * sizeof(double) <= GEMM_LDS_BYTES, "blocked potrf scratch must fit the GEMM LDS");

template <class SM = SmemKernel>
__device__ __forceinline__ d4 load_blk(int off, int lane, SM sm = SM())
{
    d4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = sm[off + r * 64 + lane];
    return v;
}
template <class SM = SmemKernel>
__device__ __forceinline__ void store_blk(int off, int lane, const d4& v, SM sm = SM())
{
#pragma unroll
    for (int r = 0; r < 4; ++r) sm[off + r * 64 + lane] = v[r];
}
// c += x^T y, all three blocks in the accumulator layout
__device__ __forceinline__ d4 mma16(const d4& x, const d4& y, d4 c)
{
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s], y[s], c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ double bcast_lane(double v, int src_lane)   // src_lane wave-uniform
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d) to full precision for a positive, normal d: hardware estimate + two Newton steps
// (the library rsqrt()/sqrt() cost ~25 dependent instructions each and sit on the pivot chain)
__device__ __forceinline__ double rsqrt_chain(double d)
{
    double y = __builtin_amdgcn_rsq(d);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double t = d * y;
        const double e = fma(-t, y, 1.0);
        y = fma(0.5 * y, e, y);
    }
    return y;
}

// In-wave Cholesky of a 16 x 16 block t (accumulator layout) with the row eliminations mirrored on an
// identity block: on return t = U (strictly lower entries zeroed) and w = U^-T.  Branch-free per
// pivot: all six LDS reads are issued right behind the line write and overlap the rsqrt chain.
template <class SM = SmemKernel>
__device__ __forceinline__ void chol16(d4& t, d4& w, int lane, int& bad, SM sm = SM())
{
    const int q = lane >> 4, c = lane & 15;
    auto* line = sm.ptr(OFF_LINE);
    d4 e;
#pragma unroll
    for (int r = 0; r < 4; ++r) e[r] = (q + 4 * r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
#pragma unroll 1
        for (int rq = 0; rq < 4; ++rq) {
            const int jj = 4 * rr + rq;     // pivot row jj lives in lanes q == rq, register rr
            const double d = bcast_lane(t[rr], rq * 16 + jj);
            if (q == rq) {
                line[c] = t[rr];
                line[16 + c] = e[rr];
            }
            __builtin_amdgcn_wave_barrier();
            const double la = line[c], le = line[16 + c];
            double lm[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) lm[r] = line[q + 4 * r];
            __builtin_amdgcn_sched_barrier(0);
            if (!(d > 0.0)) bad = 1;
            const double inv = rsqrt_chain(d);
            const double pA = la * inv, pE = le * inv;
            const double ujj = d * inv;       // U_jj = sqrt(d)
            // rows below the pivot row are eliminated; rows above it get the multiplier 0 and stay as they
            // are (no selects on the results); the pivot row itself is one register in the lanes q == rq
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double m = (q + 4 * r > jj) ? lm[r] * inv : 0.0;   // U[jj][i], i = q + 4r
                t[r] = fma(-m, pA, t[r]);
                e[r] = fma(-m, pE, e[r]);
            }
            if (q == rq) {
                t[rr] = (c == jj) ? ujj : pA;
                e[rr] = pE;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (c < q + 4 * r) t[r] = 0.0;
    w = e;
}

// one finished block of W into the k-major operand Wt[e][i] = W[i][e] of the strip solve, transposed
// through a per-wave LDS scratch so the global stores are 128-byte row segments
// COUNTED: the caller counts this routine's global stores in an s_waitcnt vmcnt(N) (potrf_spine.hpp, pub_flag): exactly
// four store instructions, which relaxed wavefront-scope atomic stores guarantee -- same instruction as a plain store,
// but the compiler may neither merge nor split nor drop them.
template <class SM = SmemKernel, bool COUNTED = false>
__device__ __forceinline__ void emit_w(const d4& w, int I, int J, int lane, int wave, double* Wm, SM sm = SM())
{
    const int q = lane >> 4, c = lane & 15;
    auto* tr = sm.ptr(OFF_TR + wave * 272);
#pragma unroll
    for (int r = 0; r < 4; ++r) tr[(q + 4 * r) * 17 + c] = w[r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        double* dst = &Wm[(size_t)(16 * J + q + 4 * r) * NB + 16 * I + c];
        const double v = tr[c * 17 + q + 4 * r];
        if constexpr (COUNTED) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else *dst = v;
    }
    __builtin_amdgcn_wave_barrier();
}

// trailing update of block row I (> bb) with block row bb: both of the wave's blocks, plus its rhs
// block when it owns that row's
template <int I>
__device__ __forceinline__ void trail_row(d4 (&blk)[16], d4 (&rhs)[2], int bb, int wave, int lane, bool own_rhs)
{
    const d4 xi = load_blk(OFF_UROW + I * BLK, lane);
    d4 y[2], xs[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int J = ((wave - 3 * I) & 3) + 4 * h;
        const bool up = (J >= I), lo = (J <= bb);
        // a block outside the active set (bb < J < I) gets a zero X operand and a finite dummy Y: adds exactly 0
        const int yoff = up ? OFF_UROW + J * BLK : (lo ? OFF_WROW + J * BLK : OFF_UROW + I * BLK);
        const double sc = (up || lo) ? -1.0 : 0.0;
        y[h] = load_blk(yoff, lane);
#pragma unroll
        for (int r = 0; r < 4; ++r) xs[h][r] = xi[r] * sc;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) blk[2 * I + h] = mma16(xs[h], y[h], blk[2 * I + h]);
    if (own_rhs) {
        d4 xn;
#pragma unroll
        for (int r = 0; r < 4; ++r) xn[r] = -xi[r];
        rhs[I >> 2] = mma16(xn, load_blk(OFF_UROW + 8 * BLK, lane), rhs[I >> 2]);
    }
}

}  // namespace pb

// All 256 threads call this.  Uses the first pb::LDS_DOUBLES doubles of the dynamic LDS (free
// between two tile_gemm_tn calls).  The tile (k0, k0) of Km must hold the updated symmetric block
// (its upper triangle is read); Wm's strictly upper part must be zero (it is never written).
// ABLATE: timing diagnostics for the microbenchmark only (1 no in-wave factorisation, 2 no MFMA
// phases, 3 no W output, 9 phase stamps); 0 is the shipped routine.
struct PotrfNoWait {
    __device__ __forceinline__ void operator()() const {}
};

// FUSED (the diagonal task of the latency scheme): the tile does not come from the matrix but is formed here,
// in the registers the factorisation works on --
//     T = part - strip^T strip,
// `part` = the running sum the tile's PART chain left in a workspace slot (row-major 128 x 128: K minus the
// contributions of all block rows but the last), `strip` = the 128 x 128 tile right above the diagonal
// (k-major, leading dimension ld), i.e. the final K = 128 symmetric update.  `part` is read BEFORE
// wait_dep() -- the wait for the strip -- and the update runs on the 16 x 16 blocks each wave owns (12 of
// its 16 slots can hold an upper block; dead ones get a zero multiplier), so compared with
// "tile engine -> store -> drain -> reload" the row-to-row path loses a round trip through memory, the
// B-operand staging and a quarter of the MFMAs.
template <int ABLATE = 0, bool FUSED = false, class WaitFn = PotrfNoWait>
__device__ __forceinline__ void potrf_blocked(double* Km, int ld, int k0, double* Wm, double* Rv, MatAcc* acc,
                                              const double* __restrict__ part = nullptr,
                                              const double* __restrict__ strip = nullptr, WaitFn wait_dep = WaitFn(),
                                              unsigned long long* tl = nullptr)
{
    using namespace pb;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // opaque: no per-lane offset of this routine outlives it (spills)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: block ownership tests become s_cbranch
    const int q = lane >> 4, c = lane & 15;
    const int rhs_row = (3 * wave) & 3;   // this wave owns the rhs blocks of block rows rhs_row and rhs_row + 4
    d4 blk[16];                           // slot 2I + h holds block (I, J), J = ((wave - 3I) & 3) + 4h
    d4 rhs[2];                            // slot k holds block (rhs_row + 4k, 8): r_k in column 0
    {
        const double* src = FUSED ? part : Km + (size_t)k0 * ld + k0;
        const size_t lds = FUSED ? (size_t)NB : (size_t)ld;
#pragma unroll
        for (int I = 0; I < 8; ++I)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int J = ((wave - 3 * I) & 3) + 4 * h;
                d4 v = {0.0, 0.0, 0.0, 0.0};
                if (J >= I) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = src[(size_t)(16 * I + q + 4 * r) * lds + 16 * J + c];
                }
                blk[2 * I + h] = v;
            }
    }
    if (FUSED) {
        wait_dep();                                   // the strip (and the rhs block) are final from here on
        // K = 128 in eight LDS stages of 16 k-rows, double-buffered, one operand
        stage_glds_one(strip, (size_t)ld, 0, 0, tid);
        __syncthreads();
        const int fr = lane & 15, fk = lane >> 4;
#pragma unroll 1
        for (int ch = 0; ch < NB / KB; ++ch) {
            const int cur = ch & 1;
            if (ch + 1 < NB / KB) stage_glds_one(strip, (size_t)ld, (ch + 1) * KB, cur ^ 1, tid);
            const int base = cur * LDS_BUFFER + fk * LDS_LD + fr;
#pragma unroll
            for (int I = 0; I < 8; ++I) {
                // slot h = 1 of rows 0..3 is always an upper block (J >= 4 > I); slot h = 0 of rows 4..7 never
                // is (J <= 3 < I); the remaining slot of each row is upper or not depending on the wave
                const int J0 = (wave - 3 * I) & 3;
                const int Jc = (I < 4) ? J0 : J0 + 4;             // the conditional slot's column block
                const double sc = (Jc >= I) ? -1.0 : 0.0;         // dead block: zero multiplier, adds exactly 0
                double x[4], xc[4], yc[4], y1[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    x[ks] = psoap_smem[base + ks * 4 * LDS_LD + 16 * I];
                    yc[ks] = psoap_smem[base + ks * 4 * LDS_LD + 16 * Jc];
                    if (I < 4) y1[ks] = psoap_smem[base + ks * 4 * LDS_LD + 16 * (J0 + 4)];
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    xc[ks] = x[ks] * sc;
                    x[ks] = -x[ks];
                }
                const int hc = (I < 4) ? 0 : 1;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    blk[2 * I + hc] = __builtin_amdgcn_mfma_f64_16x16x4f64(xc[ks], yc[ks], blk[2 * I + hc], 0, 0, 0);
                    if (I < 4)
                        blk[2 * I + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[ks], y1[ks], blk[2 * I + 1], 0, 0, 0);
                }
            }
            __syncthreads();
        }
    }
    if (FUSED && tl && tid == 0) tl[5] = __builtin_amdgcn_s_memrealtime();     // debug stamps (DAG task log)
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) rhs[k][r] = (c == 0) ? Rv[k0 + 16 * (rhs_row + 4 * k) + q + 4 * r] : 0.0;
    int bad = 0;
    double logsum = 0.0;
    d4 wdiag = {0.0, 0.0, 0.0, 0.0};

#define PSOAP_STAMP(k)                                                                              \
    if (ABLATE == 9 && lane == 0 && wave < 2) g_potrf_stamps[(wave * 8 + bb) * 6 + (k)] = __builtin_amdgcn_s_memtime();
    // runtime loop over the 8 block steps; the register slots stay statically indexed through
    // unrolled loops over I guarded by scalar comparisons with bb
#pragma unroll 1
    for (int bb = 0; bb < 8; ++bb) {
        PSOAP_STAMP(0)
        // ---- A: diagonal block (bb, bb), inside wave 0: slot 2bb + (bb >> 2)
        if (wave == 0) {
            d4 t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int I = 0; I < 8; ++I)
                if (I == bb) t = blk[2 * I + (I >> 2)];
            if (ABLATE != 1) chol16(t, wdiag, lane, bad); else wdiag = t;
#pragma unroll
            for (int I = 0; I < 8; ++I)
                if (I == bb) blk[2 * I + (I >> 2)] = t;
            // one diagonal element per lane at most: a single log
            double dg = 1.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) dg = (c == q + 4 * r) ? t[r] : dg;
            logsum += log(dg);
            store_blk(OFF_WROW + bb * BLK, lane, wdiag);
            // W_bb^T for the other waves (X operand of  W_bb Y = (W_bb^T)^T Y)
#pragma unroll
            for (int r = 0; r < 4; ++r) psoap_smem[OFF_TR + (q + 4 * r) * 17 + c] = wdiag[r];
            __builtin_amdgcn_wave_barrier();
            d4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = psoap_smem[OFF_TR + c * 17 + q + 4 * r];
            __builtin_amdgcn_wave_barrier();
            store_blk(OFF_V, lane, v);
        }
        PSOAP_STAMP(1)
        __syncthreads();
        PSOAP_STAMP(2)
        // ---- B: finish block row bb.  Both of a wave's blocks go through the same straight-line MFMA
        // sequence (uniform branches around each would serialise LDS latency + 4 dependent MFMAs per
        // block); the diagonal block is kept by a select and publishes into a dummy slot.
        if (ABLATE != 2) {
            const d4 x = load_blk(OFF_V, lane);
#pragma unroll
            for (int I = 0; I < 8; ++I) {
                if (I != bb) continue;
                d4 res[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) res[h] = mma16(x, blk[2 * I + h], d4{0.0, 0.0, 0.0, 0.0});
                if ((I & 3) == rhs_row) {
                    rhs[I >> 2] = mma16(x, rhs[I >> 2], d4{0.0, 0.0, 0.0, 0.0});   // z of this block row
                    store_blk(OFF_UROW + 8 * BLK, lane, rhs[I >> 2]);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int J = ((wave - 3 * I) & 3) + 4 * h;
                    const int off = (J > bb) ? OFF_UROW + J * BLK : ((J < bb) ? OFF_WROW + J * BLK : OFF_DUMMY);
                    store_blk(off, lane, res[h]);
                    if (J != bb) blk[2 * I + h] = res[h];
                }
                if (ABLATE != 3) {
                    if (wave == 0) emit_w(wdiag, bb, bb, lane, 0, Wm);
                }
            }
        }
        PSOAP_STAMP(3)
        __syncthreads();
        PSOAP_STAMP(4)
        // ---- C: trailing update
        // (one uniform branch per block row: fully straight-line code over all rows lets the compiler
        // hoist every row's operand loads and spills the 144 block registers)
        if (ABLATE != 2) {
#define PSOAP_ROW(I_) if (bb < I_) trail_row<I_>(blk, rhs, bb, wave, lane, (I_ & 3) == rhs_row);
            PSOAP_ROW(1) PSOAP_ROW(2) PSOAP_ROW(3) PSOAP_ROW(4) PSOAP_ROW(5) PSOAP_ROW(6) PSOAP_ROW(7)
#undef PSOAP_ROW
        }
        PSOAP_STAMP(5)
        // (the next step's phase A only touches OFF_LINE/OFF_TR/OFF_V and one WROW block that phase C
        // of this step does not read -- J <= bb there -- so no barrier is needed here)
    }
#undef PSOAP_STAMP

    if (FUSED && tl && tid == 0) tl[6] = __builtin_amdgcn_s_memrealtime();
    // the strictly lower blocks of W = U11^-T stay in their registers once their block row is finished:
    // written out here, off the eight-step critical loop (transposes + global stores per step; -6 % on a
    // single N = 6000 evaluation, -1 % on the 32-walker batch)
    if (ABLATE != 3) {
#pragma unroll
        for (int I = 1; I < 8; ++I)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int J = ((wave - 3 * I) & 3) + 4 * h;
                if (J < I) emit_w(blk[2 * I + h], I, J, lane, wave, Wm);
            }
    }
    // U11 back to the matrix
#pragma unroll
    for (int I = 0; I < 8; ++I)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int J = ((wave - 3 * I) & 3) + 4 * h;
            if (J >= I) {
                const d4& v = blk[2 * I + h];
#pragma unroll
                for (int r = 0; r < 4; ++r) Km[(size_t)(k0 + 16 * I + q + 4 * r) * ld + k0 + 16 * J + c] = v[r];
            }
        }
    // z (column 0 of the rhs blocks) back into r, and z^T z
    double zz = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (c == 0) {
                const double z = rhs[k][r];
                Rv[k0 + 16 * (rhs_row + 4 * k) + q + 4 * r] = z;
                zz = fma(z, z, zz);
            }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        logsum += __shfl_xor(logsum, off, 64);   // only wave 0 holds diagonal blocks
        zz += __shfl_xor(zz, off, 64);
    }
    if (lane == 0) {
        psoap_smem[OFF_RED + wave] = zz;
        if (wave == 0) psoap_smem[OFF_RED + 4] = logsum;
    }
    const int anybad = __syncthreads_or(bad);
    if (tid == 0) {
        const double l = psoap_smem[OFF_RED + 4];
        const double qd = ((psoap_smem[OFF_RED + 0] + psoap_smem[OFF_RED + 1]) + psoap_smem[OFF_RED + 2]) +
                          psoap_smem[OFF_RED + 3];
        // this block's record (common.hpp, MatAcc): written, never read here -- no chain from block row to block row
        acc_store(acc, k0 / NB, l, qd, anybad != 0);
    }
    __syncthreads();   // the LDS scratch is handed back to the tile engine
    if (FUSED && tl && tid == 0) tl[1] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace psoap
