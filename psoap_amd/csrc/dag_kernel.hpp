// dag_kernel.hpp -- the whole batched factorisation as ONE persistent launch.
//
// The staged path (chol_kernels.hpp) issues three kernels per 128-row panel; every panel
// boundary quantises the work into rounds of resident workgroups and puts the diagonal-block
// factorisation on the critical path of the whole batch.  Here the same tile operations are
// tasks of a dependency graph executed by persistent 256-thread workgroups (2 per CU):
//
//   DIAG(b,q)    tile (q,q) of matrix b: left-looking MFMA update over the q finished block
//                rows, then an in-block Cholesky of the 128 x 128 tile that also yields
//                U11^-T (operand of the strip solve), z_q = U11^-T r_q and the logdet/quad sums.
//   OFF(b,q,j)   tile (q,j), j > q: the same MFMA update, then X = U11^-T (tile) as a K=128 MFMA
//                product, then r[j-block] -= X^T z_q.
//
// Tasks are drawn from one atomic ticket counter in the order (q, DIAGs first, then b, j), so a
// task only ever waits for tasks with smaller tickets, which are held by running workgroups:
// no co-residency assumption, no deadlock, and while one matrix waits for its diagonal block the
// workgroups work on the other matrices of the batch.  Hand-offs follow the agent-scope
// release/acquire recipe of cdna_hip_programming.md Guideline 16: plain stores, every wave drains
// (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane release fence + drain, relaxed agent-scope
// atomic on the counter; consumers poll relaxed, ONE acquire fence, drain, barrier, plain vector
// loads.  Every spin is bounded and reports through DagCtl::error instead of hanging the GPU.
//
// Per matrix three words: rows_done (block rows fully finished), potrf_done (diagonal blocks
// factored), cnt (finished tasks of the current block row).  The update of block row q reads
// rows < q; it is split so that rows < q-1 are consumed before waiting for row q-1 (look-ahead).
#pragma once
#include "chol_kernels.hpp"
#include "fill_kernels.hpp"

namespace psoap {

struct alignas(64) MatFlags {
    int rows_done;
    int potrf_done;
    int cnt;
    int pad[13];
};

struct alignas(64) DagCtl {
    unsigned int ticket;
    unsigned int error;
    unsigned int pad[14];
};

constexpr long long DAG_MAX_SPINS = 2000000;  // x (s_sleep + atomic round trip) ~ seconds

#define PSOAP_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Poll of a flag word.  Measured on gfx950 (ROCm 7.2): a relaxed agent-scope LOAD (global_load sc1)
// is served by the polling XCD's L2 and can return the value from before another XCD's sc1 store
// indefinitely once the line is resident there; a read-modify-write executes at the memory side and
// always observes the latest value, so the poll is an atomic add of zero.
__device__ __forceinline__ int dag_peek(int* flag) { return __hip_atomic_fetch_add(flag, 0, PSOAP_RLX_AGENT); }

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
__device__ __forceinline__ void dag_task_done(MatFlags* f, int q, int ntasks_row)
{
    const int old = __hip_atomic_fetch_add(&f->cnt, 1, PSOAP_RLX_AGENT);
    if (old + 1 == ntasks_row) {
        // last finisher of the row: order after every other task's release, then publish the row
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&f->cnt, 0, PSOAP_RLX_AGENT);
        __hip_atomic_store(&f->rows_done, q + 1, PSOAP_RLX_AGENT);
    }
}

// left-looking update with look-ahead: rows [0, k0-128) need rows_done >= q-1, the last 128 need q
__device__ __forceinline__ void dag_update(Tile& t, double* Km, int ld, int k0, int j0, int q, MatFlags* f,
                                           DagCtl* ctl)
{
    t.zero();
    if (q == 0) return;
    const int k1 = k0 - NB;
    if (k1 > 0) {
        dag_wait_ge(&f->rows_done, q - 1, ctl, 1u);
        tile_gemm_tn(t, Km + k0, (size_t)ld, Km + j0, (size_t)ld, k1);
    }
    dag_wait_ge(&f->rows_done, q, ctl, 2u);
    tile_gemm_tn(t, Km + (size_t)k1 * ld + k0, (size_t)ld, Km + (size_t)k1 * ld + j0, (size_t)ld, NB);
}

// Fused kernel-matrix fill: tile (k0, j0) <- K(i, j) - acc, with K evaluated on the fly in the
// MFMA accumulator layout (same arithmetic as k_fill_sym: squared-exponential sum, diagonal rule,
// sigma^2 on the diagonal, identity padding).  The covariance matrix is therefore never
// materialised in HBM: each tile is written exactly once, already updated.
template <int C>
__device__ __forceinline__ void dag_store_updated(const Tile& t, double* Km, int ld, int k0, int j0,
                                                  const double* __restrict__ lw, const GpDev& g, double dsum,
                                                  const double* __restrict__ sigma, int N)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    double xj[4][C];
    int jj[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        jj[n] = j0 + tile_col(wc, n, lane);
#pragma unroll
        for (int c = 0; c < C; ++c) xj[n][c] = (jj[n] < N) ? lw[(size_t)c * N + jj[n]] : 0.0;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        double xi[4][C];
        int ii[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ii[r] = k0 + tile_row(wr, m, lane, r);
#pragma unroll
            for (int c = 0; c < C; ++c) xi[r][c] = (ii[r] < N) ? lw[(size_t)c * N + ii[r]] : 0.0;
        }
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ii[r], j = jj[n];
                double v;
                if (i < N && j < N) {
                    if (i == j) {
#pragma clang fp contract(off)
                        const double sg = sigma[i];
                        v = dsum + sg * sg;
                    } else {
                        v = kern_elem<C>(xi[r], xj[n], g);
                    }
                } else {
                    v = (i == j) ? 1.0 : 0.0;
                }
                Km[(size_t)i * ld + j] = v - t.acc[m][n][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------
// In-block Cholesky of a 128 x 128 tile by 256 threads (16 x 16 grid: rows ty+16a, cols tx+16b,
// 64 values per thread).  The register image M starts as the symmetric tile; the upper triangle
// becomes U, and the strictly lower triangle is re-used in place for the strictly lower part of
// W = U^-T (its diagonal is 1/U_jj, kept in dinv): slot (i, c<i) is first written at pivot step c
// (W_ic = -U_ci / U_cc ...) and afterwards receives the same row eliminations as the upper part.
// One LDS broadcast line and one barrier per pivot.
// ---------------------------------------------------------------------------------------------
template <int JA>
__device__ __forceinline__ void potrf256_phase(double (&M)[8][8], double (*rowbuf)[NB], double* dinv, int ty, int tx,
                                               int& bad)
{
#pragma unroll 1
    for (int jr = 0; jr < 16; ++jr) {
        const int j = 16 * JA + jr;
        const int cur = j & 1;
        if (ty == jr) {
#pragma unroll
            for (int b = 0; b < 8; ++b) rowbuf[cur][tx + 16 * b] = M[JA][b];
        }
        __syncthreads();
        const double d = rowbuf[cur][j];
        if (!(d > 0.0)) bad = 1;
        const double inv = rsqrt(d);
        double p[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) p[b] = rowbuf[cur][tx + 16 * b] * inv;
        const bool pivcol = (tx == jr);  // column j lives in block b == JA at tx == jr
#pragma unroll
        for (int a = JA; a < 8; ++a) {
            double mval = rowbuf[cur][ty + 16 * a] * inv;  // U_ji, i = ty + 16a > j
            if (a == JA && ty <= jr) mval = 0.0;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                if (b == JA) {
                    // first touch of W_ij at column j: -U_ji / U_jj; other columns: plain elimination
                    const double upd = fma(-mval, p[b], M[a][b]);
                    const bool live = !(a == JA && ty <= jr);
                    M[a][b] = (pivcol && live) ? (-mval * inv) : upd;
                } else {
                    M[a][b] = fma(-mval, p[b], M[a][b]);
                }
            }
        }
        if (ty == jr) {
#pragma unroll
            for (int b = 0; b < 8; ++b) M[JA][b] = p[b];   // scaled pivot row: U_j,c>j and W_j,c<j
            if (pivcol) M[JA][JA] = sqrt(d);               // U_jj
            if (tx == 0) dinv[j] = inv;                    // W_jj
        }
    }
}

// Factor tile (k0,k0) of Km in place; write W^T (k-major) to Wm; z = W r_k into Rv[k0..]; sums to acc.
// All 256 threads call this; uses the three LDS arrays passed in.
__device__ __forceinline__ void potrf256(double* Km, int ld, int k0, double* Wm, double* Rv, MatAcc* acc,
                                         double (*rowbuf)[NB], double* dinv, double* rk, double (*red)[4])
{
    const int tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    double M[8][8];
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int i = ty + 16 * a;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int c = tx + 16 * b;
            // upper triangle from memory; the lower slots start as the mirror (their first use overwrites them)
            M[a][b] = (c >= i) ? Km[(size_t)(k0 + i) * ld + k0 + c] : Km[(size_t)(k0 + c) * ld + k0 + i];
        }
    }
    if (tid < NB) rk[tid] = Rv[k0 + tid];
    int bad = 0;
    // the factorisation is a latency-bound chain on the batch's critical path: let its waves win
    // issue arbitration against the MFMA workgroup sharing the CU
    __builtin_amdgcn_s_setprio(3);
    potrf256_phase<0>(M, rowbuf, dinv, ty, tx, bad);
    potrf256_phase<1>(M, rowbuf, dinv, ty, tx, bad);
    potrf256_phase<2>(M, rowbuf, dinv, ty, tx, bad);
    potrf256_phase<3>(M, rowbuf, dinv, ty, tx, bad);
    potrf256_phase<4>(M, rowbuf, dinv, ty, tx, bad);
    potrf256_phase<5>(M, rowbuf, dinv, ty, tx, bad);
    potrf256_phase<6>(M, rowbuf, dinv, ty, tx, bad);
    potrf256_phase<7>(M, rowbuf, dinv, ty, tx, bad);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();

    double logpart = 0.0, quadpart = 0.0;
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int i = ty + 16 * a;
        double zp = 0.0;
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const int c = tx + 16 * b;
            double w;  // W[i][c] = (U^-T)[i][c]
            if (c > i) {
                Km[(size_t)(k0 + i) * ld + k0 + c] = M[a][b];
                w = 0.0;
            } else if (c == i) {
                Km[(size_t)(k0 + i) * ld + k0 + c] = M[a][b];
                logpart += log(M[a][b]);
                w = dinv[i];
            } else {
                w = M[a][b];
            }
            Wm[(size_t)c * NB + i] = w;  // k-major operand: Wt[e=c][i]
            zp = fma(w, rk[c], zp);
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) zp += __shfl_xor(zp, off, 64);
        if (tx == 0) {
            Rv[k0 + i] = zp;
            quadpart = fma(zp, zp, quadpart);
        }
    }
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
        const double l = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3];
        const double qd = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        // MatAcc is handed from block row to block row across workgroups: agent-scope accesses only
        const double l0 = __hip_atomic_load(&acc->logdet_half, PSOAP_RLX_AGENT);
        const double q0 = __hip_atomic_load(&acc->quad, PSOAP_RLX_AGENT);
        __hip_atomic_store(&acc->logdet_half, l0 + l, PSOAP_RLX_AGENT);
        __hip_atomic_store(&acc->quad, q0 + qd, PSOAP_RLX_AGENT);
        if (anybad) __hip_atomic_store(&acc->info, 1.0, PSOAP_RLX_AGENT);
    }
}

// strip solve + right-hand-side update for tile (k0, j0); the updated tile is already in memory
__device__ __forceinline__ void dag_trsm(Tile& t, double* Km, int ld, int k0, int j0, const double* Wm, double* Rv,
                                         int Npad, double* zk, double* colsum)
{
    const int tid = threadIdx.x;
    if (tid < NB) zk[tid] = Rv[k0 + tid];
    t.zero();
    tile_gemm_tn(t, Wm, (size_t)NB, Km + (size_t)k0 * ld + j0, (size_t)ld, NB);
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
    if (wr == 0 && lane < 16 && j0 < Npad) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = tile_col(wc, n, lane);
            Rv[j0 + col] -= part[n] + colsum[col];
        }
    }
}

template <int C>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_chol_dag(double* Kbase, size_t mat_stride, int ld, int P, int B,
                                                             double* Wt, double* Rbase, int Npad, MatAcc* acc,
                                                             MatFlags* flags, DagCtl* ctl, unsigned long long* tlog,
                                                             const double* __restrict__ lwl,
                                                             const double* __restrict__ gp,
                                                             const double* __restrict__ sigma, int N)
{
    __shared__ double rowbuf[2][NB];
    __shared__ double dinv[NB];
    __shared__ double vec1[NB];   // r_k (DIAG) / z_k (OFF)
    __shared__ double vec2[NB];   // column sums (OFF)
    __shared__ double red[2][4];
    __shared__ unsigned int s_ticket;
    const unsigned int total = (unsigned int)B * (unsigned int)(P * (P + 1) / 2);
    Tile t;
    for (;;) {
        if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&ctl->ticket, 1u, PSOAP_RLX_AGENT);
        __syncthreads();
        const unsigned int ticket = s_ticket;
        __syncthreads();  // s_ticket is rewritten at the top of the next iteration
        if (ticket >= total) return;
        if (__hip_atomic_fetch_or(&ctl->error, 0u, PSOAP_RLX_AGENT) != 0u) return;
        int q, dummy;
        decode_upper((int)(ticket / (unsigned int)B), P, q, dummy);
        const int row_first = (q * P - q * (q - 1) / 2) * B;
        const int rem = (int)ticket - row_first;
        const int ntasks_row = P - q;
        int b, j;
        if (rem < B) {
            b = rem;
            j = q;
        } else {
            const int r2 = rem - B;
            b = r2 / (ntasks_row - 1);
            j = q + 1 + r2 % (ntasks_row - 1);
        }
        double* Km = Kbase + (size_t)b * mat_stride;
        double* Rv = Rbase + (size_t)b * Npad;
        double* Wm = Wt + (size_t)b * NB * NB;
        MatFlags* f = flags + b;
        const int k0 = q * NB, j0 = j * NB;

        if (tlog && threadIdx.x == 0) tlog[ticket * 4 + 0] = __builtin_amdgcn_s_memrealtime();
        dag_update(t, Km, ld, k0, j0, q, f, ctl);
        {
            GpDev g;
            load_gp(gp + (size_t)b * 2 * C, C, g);
            double dsum = g.a2[0];
            {
#pragma clang fp contract(off)
                for (int c = 1; c < C; ++c) dsum = dsum + g.a2[c];
            }
            dag_store_updated<C>(t, Km, ld, k0, j0, lwl + (size_t)b * C * N, g, dsum, sigma, N);
        }
        dag_drain();  // the tile is re-read below in another layout by other waves of this block
        if (tlog && threadIdx.x == 0) tlog[ticket * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        if (j == q) {
            potrf256(Km, ld, k0, Wm, Rv, acc + b, rowbuf, dinv, vec1, red);
            dag_drain();
            if (tlog && threadIdx.x == 0) tlog[ticket * 4 + 2] = __builtin_amdgcn_s_memrealtime();
            if (threadIdx.x == 0) {
                dag_release_fence();
                __hip_atomic_store(&f->potrf_done, q + 1, PSOAP_RLX_AGENT);
                dag_task_done(f, q, ntasks_row);
            }
        } else {
            dag_wait_ge(&f->potrf_done, q + 1, ctl, 3u + 16u * (unsigned int)q + 4096u * (unsigned int)b);
            if (tlog && threadIdx.x == 0) tlog[ticket * 4 + 2] = __builtin_amdgcn_s_memrealtime();
            dag_trsm(t, Km, ld, k0, j0, Wm, Rv, Npad, vec1, vec2);
            dag_drain();
            if (threadIdx.x == 0) {
                dag_release_fence();
                dag_task_done(f, q, ntasks_row);
            }
        }
        if (tlog && threadIdx.x == 0) tlog[ticket * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace psoap
