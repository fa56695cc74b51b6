// chol_kernels.hpp -- batched left-looking blocked Cholesky K = U^T U with the
// triangular solve and log-determinant folded in.
//
// Replaces, for a whole batch of proposals at once, the LAPACK calls of the
// reference's lnlike_* (psoap/covariance.py:325-331,348-354,370-376):
//   cho_factor (dpotrf, upper)  -> panel_update (MFMA) + potrf_diag + trsm_strip (MFMA)
//   logdet = sum 2 log diag     -> accumulated in potrf_diag
//   cho_solve + dot             -> z = U^-T r folded into the factorisation; r^T K^-1 r = z^T z
//
// Matrix layout in HBM: per proposal one (Npad x ld) row-major array, Npad = ld =
// N rounded up to 128; only the upper triangle is referenced; the padding is the
// identity, which leaves logdet and the solve unchanged.  The factor U overwrites K.
//
// Why left-looking: the panel update
//     P = K[k0:k0+128, k0:] - U[0:k0, k0:k0+128]^T U[0:k0, k0:]
// is one long-K GEMM per 128 x 128 tile that reads each operand row once and
// touches the output tile once, i.e. 32 flop per HBM byte at fp64 (a right-looking
// K=128 trailing update re-reads and re-writes the trailing matrix every step:
// 16 flop/B, below the MI355X fp64 balance point of ~13-15 flop/B once L2 misses
// are counted).
#pragma once
#include "gemm_core.hpp"

namespace psoap {

// r = fl - mu (padded with zeros); accumulators cleared.   covariance.py:331 (fl - mu_GP)
__global__ void k_init_rhs(double* __restrict__ R, int Npad, int N, const double* __restrict__ fl, double mu,
                           MatAcc* __restrict__ acc)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Npad) R[(size_t)b * Npad + i] = (i < N) ? (fl[i] - mu) : 0.0;
    // (the per-block records are all written by the factorisation -- common.hpp, MatAcc; only the LAST one is cleared: the
    // staged path beyond 254 block rows (N > 32640, kernel boundaries between the diagonal blocks) adds the rest up in it)
    if (i == 0) acc[(size_t)b * ACC_ROWS + ACC_ROWS - 1] = MatAcc{0.0, 0.0, 0.0, 0.0};
}

// Left-looking update of block row k0 (tiles j0 = k0 + 128*blockIdx.x), K-loop over the k0 finished rows.
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_panel_update(double* __restrict__ Kbase, size_t mat_stride,
                                                                 int ld, int k0)
{
    double* Km = Kbase + (size_t)blockIdx.y * mat_stride;
    const int j0 = k0 + NB * blockIdx.x;
    Tile t;
    t.zero();
    tile_gemm_tn(t, Km + k0, (size_t)ld, Km + j0, (size_t)ld, k0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        double* p0 = Km + (size_t)(k0 + tile_row(wr, m, lane, 0)) * ld + j0 + tile_col(wc, 0, lane);
        double v[4][4];
        tile_load16(p0, (size_t)4 * ld, v);
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) p0[(size_t)4 * r * ld + 16 * n] = v[n][r] - t.acc[m][n][r];
    }
}

// ---------------------------------------------------------------------------------------------
// potrf_diag: factor the 128 x 128 diagonal block at (k0,k0), produce
//   U11 (upper, back to K), W = U11^-T in k-major form Wt[e][i] = W[i][e] (operand of trsm_strip),
//   z = W r_k (overwrites r[k0:k0+128]), logdet/quad/info accumulators.
// One 512-thread workgroup per matrix.  The augmented block [A | I] (128 x 256) lives
// entirely in registers, 64 values per thread on a 16 x 32 thread grid (rows ty+16a,
// columns tx+32b); the same row eliminations that turn A into U turn I into U^-T.
// Each of the 128 pivot steps broadcasts one row through a double-buffered LDS line
// (one barrier per step); phases of 16 steps are unrolled so finished rows/column
// blocks are skipped statically.
// ---------------------------------------------------------------------------------------------
template <int JA>
__device__ __forceinline__ void potrf_phase(double (&R)[8][8], double (*rowbuf)[256], int ty, int tx, int& bad)
{
    constexpr int BA0 = JA >> 1;      // first live 32-column block of the A part
    constexpr int BE1 = 4 + (JA >> 1);  // last live block of the E part
#pragma unroll 1
    for (int jr = 0; jr < 16; ++jr) {
        const int j = 16 * JA + jr;
        const int cur = j & 1;
        if (ty == jr) {
#pragma unroll
            for (int b = BA0; b <= BE1; ++b) rowbuf[cur][tx + 32 * b] = R[JA][b];
        }
        __syncthreads();
        const double d = rowbuf[cur][j];
        if (!(d > 0.0)) bad = 1;
        const double inv = rsqrt(d);
        double p[8];
#pragma unroll
        for (int b = BA0; b <= BE1; ++b) p[b] = rowbuf[cur][tx + 32 * b] * inv;
#pragma unroll
        for (int a = JA; a < 8; ++a) {
            double mval = rowbuf[cur][ty + 16 * a] * inv;
            if (a == JA && ty <= jr) mval = 0.0;
#pragma unroll
            for (int b = BA0; b <= BE1; ++b) R[a][b] = fma(-mval, p[b], R[a][b]);
        }
        if (ty == jr) {
#pragma unroll
            for (int b = BA0; b <= BE1; ++b) R[JA][b] = p[b];
            if (tx == (j & 31)) R[JA][JA >> 1] = sqrt(d);  // U_jj (block j>>5 == JA>>1 inside a phase)
        }
    }
}

__global__ __launch_bounds__(512, 2) void k_potrf_diag(double* __restrict__ Kbase, size_t mat_stride, int ld, int k0,
                                                      double* __restrict__ Wt, double* __restrict__ Rbase, int Npad,
                                                      MatAcc* __restrict__ acc, size_t wt_stride = (size_t)NB * NB)
{
    __shared__ double rowbuf[2][256];
    __shared__ double rk[NB];
    __shared__ double red[2][8];
    const int tid = threadIdx.x;
    const int ty = tid >> 5, tx = tid & 31;
    const int bidx = blockIdx.x;
    double* Km = Kbase + (size_t)bidx * mat_stride;
    double* Rv = Rbase + (size_t)bidx * Npad;
    double R[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int i = ty + 16 * a;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int t = tx + 32 * b;
            // the lower half of the block is never used as a pivot or multiplier; mirror for tidiness
            R[a][b] = (t >= i) ? Km[(size_t)(k0 + i) * ld + k0 + t] : Km[(size_t)(k0 + t) * ld + k0 + i];
            R[a][4 + b] = (t == i) ? 1.0 : 0.0;
        }
    }
    if (tid < NB) rk[tid] = Rv[k0 + tid];
    int bad = 0;
    potrf_phase<0>(R, rowbuf, ty, tx, bad);
    potrf_phase<1>(R, rowbuf, ty, tx, bad);
    potrf_phase<2>(R, rowbuf, ty, tx, bad);
    potrf_phase<3>(R, rowbuf, ty, tx, bad);
    potrf_phase<4>(R, rowbuf, ty, tx, bad);
    potrf_phase<5>(R, rowbuf, ty, tx, bad);
    potrf_phase<6>(R, rowbuf, ty, tx, bad);
    potrf_phase<7>(R, rowbuf, ty, tx, bad);
    __syncthreads();

    // outputs: U11, Wt, z, accumulators
    double* Wm = Wt + (size_t)bidx * wt_stride;     // (a chunk handle: the persistent kernel's spacing, tile 0 of each matrix)
    double logpart = 0.0, quadpart = 0.0;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int i = ty + 16 * a;
        double zp = 0.0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int t = tx + 32 * b;
            if (t >= i) Km[(size_t)(k0 + i) * ld + k0 + t] = R[a][b];
            if (t == i) logpart += log(R[a][b]);
            Wm[(size_t)t * NB + i] = R[a][4 + b];  // Wt[e=t][i] = (U11^-T)[i][e]
            zp = fma(R[a][4 + b], rk[t], zp);
        }
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) zp += __shfl_xor(zp, off, 64);
        if (tx == 0) {
            Rv[k0 + i] = zp;
            quadpart = fma(zp, zp, quadpart);
        }
    }
    // deterministic block reduction: wave butterflies, then a fixed-order sum over the 8 waves
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        logpart += __shfl_xor(logpart, off, 64);
        quadpart += __shfl_xor(quadpart, off, 64);
    }
    const int wave = tid >> 6;
    if ((tid & 63) == 0) {
        red[0][wave] = logpart;
        red[1][wave] = quadpart;
    }
    const int anybad = __syncthreads_or(bad);
    if (tid == 0) {
        double l = 0.0, q = 0.0;
        for (int w = 0; w < 8; ++w) {
            l += red[0][w];
            q += red[1][w];
        }
        // this block's record; blocks beyond the last record share it (consecutive launches: ordered by kernel boundaries)
        MatAcc* rec = acc + (size_t)bidx * ACC_ROWS;
        const int qb = k0 / NB;
        if (qb < ACC_ROWS - 1) {
            rec[qb] = MatAcc{l, q, anybad ? 1.0 : 0.0, 0.0};
        } else {
            rec[ACC_ROWS - 1].logdet_half += l;
            rec[ACC_ROWS - 1].quad += q;
            if (anybad) rec[ACC_ROWS - 1].info = 1.0;
        }
    }
}

// trsm_strip: X = U11^-T K[k0:k0+128, j0:j0+128] for every tile right of the diagonal block
// (a K=128 MFMA GEMM against Wt), written over K, plus the right-looking update of the
// right-hand side  r[j0:j0+128] -= X^T z_k  so the triangular solve costs no extra pass.
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_trsm_strip(double* __restrict__ Kbase, size_t mat_stride,
                                                               int ld, int k0, const double* __restrict__ Wt,
                                                               double* __restrict__ Rbase, int Npad,
                                                               size_t wt_stride = (size_t)NB * NB)
{
    __shared__ double zk[NB];
    __shared__ double colsum[NB];
    const int bidx = blockIdx.y;
    double* Km = Kbase + (size_t)bidx * mat_stride;
    double* Rv = Rbase + (size_t)bidx * Npad;
    const int j0 = k0 + NB * (blockIdx.x + 1);
    const int tid = threadIdx.x;
    if (tid < NB) zk[tid] = Rv[k0 + tid];
    Tile t;
    t.zero();
    tile_gemm_tn_lower(t, Wt + (size_t)bidx * wt_stride, (size_t)NB, Km + (size_t)k0 * ld + j0, (size_t)ld);
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    double part[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tile_row(wr, m, lane, r);
            const double z = zk[row];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const double x = t.acc[m][n][r];
                Km[(size_t)(k0 + row) * ld + j0 + tile_col(wc, n, lane)] = x;
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
    // columns at or beyond Npad are appended right-hand sides (predict path), not part of r
    if (wr == 0 && lane < 16 && j0 < Npad) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = tile_col(wc, n, lane);
            Rv[j0 + col] -= part[n] + colsum[col];
        }
    }
}

// the sums over a matrix's P block records, in block order (the order every path adds them in: results are bit-identical
// whichever kernel reports them)
__host__ __device__ inline MatAcc acc_total(const MatAcc* rec, int P)
{
    MatAcc a{0.0, 0.0, 0.0, 0.0};
    if (P > ACC_ROWS) P = ACC_ROWS;
    for (int q = 0; q < P; ++q) {
        a.logdet_half += rec[q].logdet_half;
        a.quad += rec[q].quad;
        if (rec[q].info != 0.0) a.info = 1.0;
    }
    return a;
}

// one matrix's totals into a record of their own (the callers that only look at `info`: predict, calibration)
__global__ void k_acc_total(const MatAcc* __restrict__ rec, int P, MatAcc* __restrict__ out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *out = acc_total(rec, P);
}

// lnp = -0.5 * (z^T z + 2 sum log U_ii)   (covariance.py:329-331); -inf when not positive definite
__global__ void k_finalize(const MatAcc* __restrict__ acc, double* __restrict__ out, int B,
                           const int* __restrict__ too_fast, int P)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        const MatAcc a = acc_total(acc + (size_t)b * ACC_ROWS, P);
        // too_fast: an orbit proposal with |v| >= c (sample_parallel.py:186-187)
        const bool bad = (a.info != 0.0) || (too_fast != nullptr && too_fast[b] != 0);
        out[b] = bad ? -INFINITY : -0.5 * (a.quad + 2.0 * a.logdet_half);
    }
}

}  // namespace psoap
