// gemm_core.hpp -- the fp64 MFMA tile engine shared by every blocked kernel.
//
// One 256-thread workgroup (4 waves as 2 x 2) accumulates a 128 x 128 tile
//     acc[m][n] += sum_k Aop[k][m] * Bop[k][n]
// where both operands are "k-major" strips in global memory: row k is 128
// contiguous doubles (1 KiB) at Aop + k*lda.  That is how an upper Cholesky
// factor stored row-major presents its block columns, so no transposes are ever
// materialised.  K must be a multiple of KB (16).
//
// Pipeline per KB-chunk: 8 x global_load_dwordx4 per thread into registers for
// chunk c+1 (one wave covers one full 1 KiB row per instruction) -> 64 x
// v_mfma_f64_16x16x4_f64 per wave on chunk c out of LDS -> ds_write_b128 of
// chunk c+1 into the other LDS buffer -> one barrier.  LDS rows are padded to
// 144 doubles so the four k-rows of a fragment read land on disjoint banks.
//
// Fragment maps (cdna_hip_programming.md section 3, f64 16x16x4):
//   A lane l: A[i = l&15][k = l>>4];  B lane l: B[k = l>>4][j = l&15]
//   D lane l, reg r: D[row = (l>>4) + 4r][col = l&15]
// Each wave owns a 64 x 64 sub-tile = 4 x 4 MFMA tiles = 16 accumulators (128 VGPRs).
#pragma once
#include "common.hpp"

namespace psoap {

struct Tile {
    d4 acc[4][4];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = d4{0.0, 0.0, 0.0, 0.0};
    }
};

struct Staging {
    d2 a[4], b[4];
};

// Row blocks (16 rows) of a tile dealt to the two wave rows by the WORK a triangular solve has for them instead of by
// position (0-3 / 4-7): wave row 0 owns {0, 1, 7, 6}, wave row 1 {2, 3, 5, 4} (slots 0..3) -- two from the top counted
// up, two from the bottom counted down, so that a wave's four operand addresses are two scalar bases with fixed
// offsets.  (ROWMAP) the update + following solve of dag_pss.
__device__ __forceinline__ constexpr int tile_rowblock(int wr, int m) { return m < 2 ? 2 * wr + m : 7 - 2 * wr - (m - 2); }

__device__ __forceinline__ void stage_load(Staging& s, const double* __restrict__ A, size_t lda,
                                           const double* __restrict__ B, size_t ldb, int k, int tid)
{
    const int col2 = tid & 63, row0 = tid >> 6;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const size_t row = (size_t)(k + row0 + 4 * it);
        s.a[it] = *reinterpret_cast<const d2*>(A + row * lda + 2 * col2);
        s.b[it] = *reinterpret_cast<const d2*>(B + row * ldb + 2 * col2);
    }
}

// All tile kernels share ONE dynamic LDS array and address it with integer offsets, so every
// access stays in the LDS address space (ds_read/ds_write; a pointer that is selected at run
// time degrades to flat_* accesses whose waits also drain the global prefetch).
extern __shared__ __attribute__((aligned(16))) double psoap_smem[];

// How a routine reaches that array.  Kernels (and everything inlined into them) name it directly (SmemKernel): its
// offset inside the workgroup's LDS is a link-time constant that folds into the DS instructions.  A routine that is
// compiled as a real function (dag_diag_fast, dag_kernel.hpp) must NOT name it -- nor any other __shared__ variable:
// a non-kernel function has no LDS layout of its own, and hipcc then lowers every such reference to a run-time
// lookup in a per-kernel table (llvm.amdgcn.dynlds.offset.table, indexed by a hidden kernel-id SGPR).  Such a
// routine is handed the base as an explicit address_space(3) pointer by the kernel that calls it (SmemArg) and
// reaches LDS through nothing else.
typedef __attribute__((address_space(3))) double lds_double;
typedef __attribute__((address_space(3))) int lds_int;
struct SmemKernel {
    __device__ __forceinline__ double& operator[](int i) const { return psoap_smem[i]; }
    __device__ __forceinline__ lds_double* ptr(int i) const { return (lds_double*)(psoap_smem + i); }
};
struct SmemArg {
    lds_double* base;
    __device__ __forceinline__ lds_double& operator[](int i) const { return base[i]; }
    __device__ __forceinline__ lds_double* ptr(int i) const { return base + i; }
};

constexpr int LDS_OPERAND = KB * LDS_LD;     // doubles per staged operand chunk
constexpr int LDS_BUFFER = 2 * LDS_OPERAND;  // A chunk followed by B chunk

__device__ __forceinline__ void stage_store(const Staging& s, int buf, int tid)
{
    const int col2 = tid & 63, row0 = tid >> 6;
    const int base = buf * LDS_BUFFER;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int off = base + (row0 + 4 * it) * LDS_LD + 2 * col2;
        *reinterpret_cast<d2*>(&psoap_smem[off]) = s.a[it];
        *reinterpret_cast<d2*>(&psoap_smem[off + LDS_OPERAND]) = s.b[it];
    }
}

// m_first > 0 (wave-uniform): the wave's first m_first 16-row blocks issue no MFMA in this chunk
template <bool PARTIAL_M = false, class SM = SmemKernel, bool ROWMAP = false>
__device__ __forceinline__ void tile_mma_chunk(Tile& t, int buf, int wr, int wc, int lane, int m_first = 0, SM sm = SM())
{
    // the fragment offsets are recomputed per stage from the lane id (opaque to the optimiser) instead of
    // being hoisted out of the K-loop: as loop invariants they are the values hipcc spills first when the
    // kernel around the loop grows (a scratch reload in front of the fragment reads also waits for the
    // LDS-DMA of the next stage -- vmcnt counts both -- and serialises it with the MFMAs)
    // (the lane id itself comes from the exec mask -- all 64 lanes are active here -- so not even it has to
    // stay in a register across the loop)
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    (void)lane;
    const int fr = l & 15, fk = l >> 4;
    const int baseA = buf * LDS_BUFFER + fk * LDS_LD + (ROWMAP ? 32 * wr : wr * 64) + fr;
    const int baseA2 = buf * LDS_BUFFER + fk * LDS_LD + 16 * (7 - 2 * wr) + fr;      // ROWMAP: slots 2, 3 count down from here
    const int baseB = buf * LDS_BUFFER + LDS_OPERAND + fk * LDS_LD + wc * 64 + fr;
#pragma unroll
    for (int ks = 0; ks < KB / 4; ++ks) {
        double a[4], b[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            a[m] = ROWMAP ? (m < 2 ? sm[baseA + ks * 4 * LDS_LD + m * 16] : sm[baseA2 + ks * 4 * LDS_LD - (m - 2) * 16])
                          : sm[baseA + ks * 4 * LDS_LD + m * 16];
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = sm[baseB + ks * 4 * LDS_LD + n * 16];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (PARTIAL_M && m < m_first) continue;
#pragma unroll
            for (int n = 0; n < 4; ++n)
                t.acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], t.acc[m][n], 0, 0, 0);
        }
    }
}

// Needs GEMM_LDS_BYTES of dynamic LDS at launch.  All 256 threads must call this.
// Structural-zero skipping (the MFMA pipe is the contended resource, so idle quadrants are
// worth skipping even though every wave still stages operands and meets the barriers):
//   skip_lower_left : wave (wr=1, wc=0) issues no MFMA -- the strictly lower quadrant of a
//                     symmetric diagonal tile is never read back;
//   k_limit_upper   : waves wr=0 issue no MFMA for k >= k_limit_upper -- rows 0..63 of a product
//                     with a lower-triangular left factor (W = U11^-T) only involve k < 64.
__device__ __forceinline__ void tile_gemm_tn_reg(Tile& t, const double* __restrict__ A, size_t lda,
                                             const double* __restrict__ B, size_t ldb, int K,
                                             bool skip_lower_left = false, int k_limit_upper = 0x7fffffff)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    if (K <= 0) return;
    Staging s;
    stage_load(s, A, lda, B, ldb, 0, tid);
    stage_store(s, 0, tid);
    __syncthreads();
    const int nchunk = K / KB;
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        const bool more = (c + 1 < nchunk);
        if (more) stage_load(s, A, lda, B, ldb, (c + 1) * KB, tid);
        const bool idle = (skip_lower_left && wr == 1 && wc == 0) || (wr == 0 && c * KB >= k_limit_upper);
        if (!idle) tile_mma_chunk(t, cur, wr, wc, lane);
        if (more) stage_store(s, cur ^ 1, tid);
        __syncthreads();
    }
}

// lane id from the exec mask (all 64 lanes are active wherever this is used): nothing to keep in a register
// across a K-loop, nothing for the register allocator to spill and reload inside it
__device__ __forceinline__ int hw_lane()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// cache policy of the operand loads (the aux immediate of global_load_lds: 16 = sc1, past the compute unit's vector L1 and
// served by the XCD's L2).  -DPSOAP_SC1_LOADS: experiments on what a workgroup that the device's scheduler moved to
// another compute unit reads there; default 0, plain loads.
#ifdef PSOAP_SC1_LOADS
#define PSOAP_GLDS_AUX 16
#else
#define PSOAP_GLDS_AUX 0
#endif

// LDS-DMA staging: one global_load_lds_dwordx4 per wave moves one 1 KiB operand row (128 doubles)
// straight into its padded LDS row -- no staging VGPRs, no ds_write pass.  Wave w fills rows
// w, w+4, w+8, w+12 of both operand chunks.
template <class SM = SmemKernel>
__device__ __forceinline__ void stage_glds(const double* __restrict__ A, size_t lda, const double* __restrict__ B,
                                           size_t ldb, int k, int buf, int tid, SM sm = SM())
{
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = wave + 4 * it;
        const int off = buf * LDS_BUFFER + row * LDS_LD;
        __builtin_amdgcn_global_load_lds((glb_ptr)(A + (size_t)(k + row) * lda + 2 * lane), (lds_ptr)sm.ptr(off), 16, 0, PSOAP_GLDS_AUX);
        __builtin_amdgcn_global_load_lds((glb_ptr)(B + (size_t)(k + row) * ldb + 2 * lane),
                                         (lds_ptr)sm.ptr(off + LDS_OPERAND), 16, 0, PSOAP_GLDS_AUX);
    }
}

// the same with the wave index given as a scalar and the lane taken from the exec mask: no thread-id register
// lives across the K-loop (tile_gemm_tn with wave_s >= 0)
template <class SM = SmemKernel>
__device__ __forceinline__ void stage_glds_w(const double* __restrict__ A, size_t lda, const double* __restrict__ B,
                                             size_t ldb, int k, int buf, int wave, SM sm = SM())
{
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int lane = hw_lane();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = wave + 4 * it;
        const int off = buf * LDS_BUFFER + row * LDS_LD;
        __builtin_amdgcn_global_load_lds((glb_ptr)(A + (size_t)(k + row) * lda + 2 * lane), (lds_ptr)sm.ptr(off), 16, 0, PSOAP_GLDS_AUX);
        __builtin_amdgcn_global_load_lds((glb_ptr)(B + (size_t)(k + row) * ldb + 2 * lane),
                                         (lds_ptr)sm.ptr(off + LDS_OPERAND), 16, 0, PSOAP_GLDS_AUX);
    }
}

// one operand only (the symmetric update of a diagonal tile multiplies a strip with itself): the A half
// of LDS buffer `buf`
template <class SM = SmemKernel>
__device__ __forceinline__ void stage_glds_one(const double* __restrict__ A, size_t lda, int k, int buf, int tid,
                                               SM sm = SM())
{
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = wave + 4 * it;
        __builtin_amdgcn_global_load_lds((glb_ptr)(A + (size_t)(k + row) * lda + 2 * lane),
                                         (lds_ptr)sm.ptr(buf * LDS_BUFFER + row * LDS_LD), 16, 0, PSOAP_GLDS_AUX);
    }
}

// SW (scalar wave): the caller passes the wave index as a scalar it keeps (wave_s) and the staging takes the lane from
// the exec mask (stage_glds_w) -- no thread-id register lives across the K-loop.  SW = false is the plain form.
template <bool SW = false, class SM = SmemKernel, bool ROWMAP = false>
__device__ __forceinline__ void tile_gemm_tn(Tile& t, const double* __restrict__ A, size_t lda,
                                                  const double* __restrict__ B, size_t ldb, int K,
                                                  bool skip_lower_left = false, int k_limit_upper = 0x7fffffff,
                                                  int wave_s = -1, SM sm = SM())
{
    if constexpr (SW) {
        const int wave = wave_s;
        const int wr = wave >> 1, wc = wave & 1;
        if (K <= 0) return;
        stage_glds_w(A, lda, B, ldb, 0, 0, wave, sm);
        __syncthreads();
        const int nchunk = K / KB;
        for (int c = 0; c < nchunk; ++c) {
            const int cur = c & 1;
            if (c + 1 < nchunk) stage_glds_w(A, lda, B, ldb, (c + 1) * KB, cur ^ 1, wave, sm);
            const bool idle = (skip_lower_left && wr == 1 && wc == 0) || (wr == 0 && c * KB >= k_limit_upper);
            if (!idle) tile_mma_chunk<false, SM, ROWMAP>(t, cur, wr, wc, 0, 0, sm);
            __syncthreads();
        }
        return;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar offsets
    const int wr = wave >> 1, wc = wave & 1;
    if (K <= 0) return;
    stage_glds(A, lda, B, ldb, 0, 0, tid);
    __syncthreads();
    const int nchunk = K / KB;
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        if (c + 1 < nchunk) stage_glds(A, lda, B, ldb, (c + 1) * KB, cur ^ 1, tid);
        const bool idle = (skip_lower_left && wr == 1 && wc == 0) || (wr == 0 && c * KB >= k_limit_upper);
        if (!idle) tile_mma_chunk(t, cur, wr, wc, lane);
        __syncthreads();
    }
}

// X = W T for a LOWER-TRIANGULAR left factor given k-major (A[k][i] = W[i][k], zero for k > i) and
// K = 128: the strip solve with the explicit inverse W = U11^-T.  Row block mb (16 rows) only involves
// k < 16 (mb + 1), so chunk c (16 k-rows) skips the MFMAs of every row block above the diagonal:
// 36 of the 64 (chunk, row block) pairs remain (a plain K = 128 product issues all 64).
__device__ __forceinline__ void tile_gemm_tn_lower(Tile& t, const double* __restrict__ A, size_t lda,
                                                   const double* __restrict__ B, size_t ldb)
{
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    stage_glds(A, lda, B, ldb, 0, 0, tid);
    __syncthreads();
    constexpr int nchunk = NB / KB;
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        if (c + 1 < nchunk) stage_glds(A, lda, B, ldb, (c + 1) * KB, cur ^ 1, tid);
        const int m_first = c - 4 * wr;          // row blocks mb = 4 wr + m < c lie above the diagonal
        if (m_first < 4) tile_mma_chunk<true>(t, cur, wr, wc, lane, m_first);
        __syncthreads();
    }
}

// The same product with the row blocks dealt to the two wave rows by WORK instead of by position.  Row block
// rb (16 rows) takes part in chunks c <= rb, i.e. rb + 1 of the 8 chunks: in the natural order the waves of
// the lower half (row blocks 4..7) issue 26 (row block, chunk) pairs and those of the upper half 10, and the
// product takes as long as the busy half needs.  Here wave row 0 owns row blocks {0, 1, 6, 7} and wave row 1
// {2, 3, 4, 5}: 18 pairs each (288 MFMAs per wave instead of 416 / 160).  Accumulator (m, n, reg) of wave
// (wr, wc) is element (16 lower_rowblock(wr, m) + (lane >> 4) + 4 reg, 64 wc + 16 n + (lane & 15)).
// Used by the strip solve on the row-to-row critical path only (dag_diag_fast): inside the kernel body the
// different loop shape upset hipcc's register allocation of the c = 2 kernels (measured +3 %).
__device__ __forceinline__ constexpr int lower_rowblock(int wr, int m) { return wr == 0 ? (m < 2 ? m : m + 4) : m + 2; }

template <class SM = SmemKernel>
__device__ __forceinline__ void tile_gemm_tn_lower_balanced(Tile& t, const double* __restrict__ A, size_t lda,
                                                            const double* __restrict__ B, size_t ldb, SM sm = SM())
{
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    stage_glds(A, lda, B, ldb, 0, 0, tid, sm);
    __syncthreads();
    constexpr int nchunk = NB / KB;
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        if (c + 1 < nchunk) stage_glds(A, lda, B, ldb, (c + 1) * KB, cur ^ 1, tid, sm);
        const int fr = tid & 15, fk = (tid & 63) >> 4;
        const int baseA = cur * LDS_BUFFER + fk * LDS_LD + fr;
        const int baseB = cur * LDS_BUFFER + LDS_OPERAND + fk * LDS_LD + wc * 64 + fr;
        // wave-uniform: which of this wave's four row blocks still reach below the diagonal of the factor
        bool live[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) live[m] = c <= lower_rowblock(wr, m);
        if (live[0] || live[1] || live[2] || live[3]) {
#pragma unroll
            for (int ks = 0; ks < KB / 4; ++ks) {
                double a[4], b[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) a[m] = sm[baseA + ks * 4 * LDS_LD + 16 * lower_rowblock(wr, m)];
#pragma unroll
                for (int n = 0; n < 4; ++n) b[n] = sm[baseB + ks * 4 * LDS_LD + n * 16];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (!live[m]) continue;
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        t.acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], t.acc[m][n], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
}

// element coordinates of accumulator (m, n, reg) inside the 128 x 128 tile
__device__ __forceinline__ int tile_row(int wr, int m, int lane, int r) { return wr * 64 + m * 16 + (lane >> 4) + 4 * r; }
__device__ __forceinline__ int tile_col(int wc, int n, int lane) { return wc * 64 + n * 16 + (lane & 15); }

// The 16 elements a lane holds of one 16-row block of a tile in memory -- rows 4 apart (r = 0 .. 3, `row4` doubles from one
// to the next), column blocks 16 doubles apart (n = 0 .. 3) -- as ONE statement: sixteen loads and the wait for them.
// Behind a K-loop hipcc has two or three vector registers to spare and schedules the plain C++ loop as 64 load / wait / use
// round trips to memory per tile (29 us of a partial tile's hand-over, tools/predict_timeline.py; the read-modify-write
// epilogues of the staged kernels likewise); a statement it cannot split costs four.  v[n][r].
__device__ __forceinline__ void tile_load16(const double* p0, size_t row4, double (&v)[4][4])
{
    const double *p1 = p0 + row4, *p2 = p1 + row4, *p3 = p2 + row4;
    asm volatile("global_load_dwordx2 %0, %16, off\n\tglobal_load_dwordx2 %1, %17, off\n\t"
                 "global_load_dwordx2 %2, %18, off\n\tglobal_load_dwordx2 %3, %19, off\n\t"
                 "global_load_dwordx2 %4, %16, off offset:128\n\tglobal_load_dwordx2 %5, %17, off offset:128\n\t"
                 "global_load_dwordx2 %6, %18, off offset:128\n\tglobal_load_dwordx2 %7, %19, off offset:128\n\t"
                 "global_load_dwordx2 %8, %16, off offset:256\n\tglobal_load_dwordx2 %9, %17, off offset:256\n\t"
                 "global_load_dwordx2 %10, %18, off offset:256\n\tglobal_load_dwordx2 %11, %19, off offset:256\n\t"
                 "global_load_dwordx2 %12, %16, off offset:384\n\tglobal_load_dwordx2 %13, %17, off offset:384\n\t"
                 "global_load_dwordx2 %14, %18, off offset:384\n\tglobal_load_dwordx2 %15, %19, off offset:384\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3]), "=&v"(v[1][0]), "=&v"(v[1][1]),
                   "=&v"(v[1][2]), "=&v"(v[1][3]), "=&v"(v[2][0]), "=&v"(v[2][1]), "=&v"(v[2][2]), "=&v"(v[2][3]),
                   "=&v"(v[3][0]), "=&v"(v[3][1]), "=&v"(v[3][2]), "=&v"(v[3][3])
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
}

}  // namespace psoap
