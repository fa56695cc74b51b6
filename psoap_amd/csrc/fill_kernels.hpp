// fill_kernels.hpp -- squared-exponential kernel-matrix fill (HBM-write-bound).
//
// Replaces the Cython loops of the reference's psoap/matrix_functions.pyx:
//   fill_V11_f :19-59, fill_V11_f_g :99-146, fill_V11_f_g_h :149-201 (symmetric, C = 1,2,3)
//   fill_V12_f :61-96 (rectangular)
// and fuses `V11[diag] += sigma**2` (psoap/covariance.py:322,344,367).
//
// Layout: one 256-thread workgroup per 128 x 128 tile.  The tile's row
// ln-wavelengths (C vectors x 128) are staged in LDS; each lane keeps the
// ln-wavelengths of its two adjacent columns in registers and stores 16 B per row
// (a wave writes four 256-byte row segments per instruction).  Arithmetic mirrors the reference exactly
// (no FMA contraction in p*r*r and in the component sum), so the only
// difference from libm is the device exp().
#pragma once
#include "common.hpp"

namespace psoap {

struct GpDev {      // per-matrix hyper-parameters prepared on the device
    double a2[3];   // amp^2                       (pyx:28)
    double p2[3];   // -0.5 * c_kms^2 / l^2        (pyx:29)
};

__device__ inline void load_gp(const double* __restrict__ gp, int C, GpDev& g)
{
#pragma clang fp contract(off)
    for (int c = 0; c < 3; ++c) {
        if (c < C) {
            double amp = gp[2 * c], l = gp[2 * c + 1];
            g.a2[c] = amp * amp;
            g.p2[c] = -0.5 * (C_KMS * C_KMS) / (l * l);
        } else {
            g.a2[c] = 0.0;
            g.p2[c] = 0.0;
        }
    }
}

template <int C>
__device__ __forceinline__ double kern_elem(const double (&xi)[C], const double (&xj)[C], const GpDev& g)
{
#pragma clang fp contract(off)
    double cov = 0.0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        double r = xj[c] - xi[c];
        double t = g.a2[c] * exp(g.p2[c] * r * r);
        cov = (c == 0) ? t : cov + t;
    }
    return cov;
}

// exp() of NE arguments at once, operation for operation the sequence the device math library's
// double-precision exp compiles to on gfx950 (Cody-Waite reduction by ln2 in two parts, degree-11
// Horner polynomial, ldexp; constants read back from the compiled library code), so the results are
// bit-identical to exp() for every x <= 0 and for NaN -- the only arguments the kernel produces; the
// library's x > 1024 -> inf select is dropped.  The point of the batch: the NE Horner chains are
// independent (an epilogue runs with one wave per SIMD, so a single chain is latency-bound), and each
// polynomial constant is materialised once per step instead of once per call.
template <int NE>
__device__ __forceinline__ void exp_nonpos_batch(const double (&x)[NE], double (&z)[NE])
{
    constexpr double LOG2E = 0x1.71547652b82fep+0;      // 0x3ff71547652b82fe
    constexpr double NLN2_HI = -0x1.62e42fefa39efp-1;   // 0xbfe62e42fefa39ef
    constexpr double NLN2_LO = -0x1.abc9e3b39803fp-56;  // 0xbc7abc9e3b39803f
    constexpr double CK[10] = {
        0x1.ade156a5dcb37p-26,  // 0x3e5ade156a5dcb37
        0x1.28af3fca7ab0cp-22,  // 0x3e928af3fca7ab0c
        0x1.71dee623fde64p-19,  // 0x3ec71dee623fde64
        0x1.a01997c89e6b0p-16,  // 0x3efa01997c89e6b0
        0x1.a01a014761f6ep-13,  // 0x3f2a01a014761f6e
        0x1.6c16c1852b7b0p-10,  // 0x3f56c16c1852b7b0
        0x1.1111111122322p-7,   // 0x3f81111111122322
        0x1.55555555502a1p-5,   // 0x3fa55555555502a1
        0x1.5555555555511p-3,   // 0x3fc5555555555511
        0x1.000000000000bp-1,   // 0x3fe000000000000b
    };
    double dn[NE], t[NE], p[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) dn[e] = __builtin_rint(x[e] * LOG2E);
#pragma unroll
    for (int e = 0; e < NE; ++e) t[e] = __builtin_fma(NLN2_HI, dn[e], x[e]);
#pragma unroll
    for (int e = 0; e < NE; ++e) t[e] = __builtin_fma(NLN2_LO, dn[e], t[e]);
#pragma unroll
    for (int e = 0; e < NE; ++e) p[e] = __builtin_fma(CK[0], t[e], CK[1]);
    // (-DPSOAP_EXP_PROBE: a TIMING probe only -- six of the eleven Horner steps dropped, the operation count a 64-entry
    // table + degree-5 polynomial would have, WRONG values: what a table-driven exp could gain at most.  DESIGN.md 3.)
#ifdef PSOAP_EXP_PROBE
    constexpr int K0 = 8;
#else
    constexpr int K0 = 2;
#endif
#pragma unroll
    for (int k = K0; k < 10; ++k)
#pragma unroll
        for (int e = 0; e < NE; ++e) p[e] = __builtin_fma(t[e], p[e], CK[k]);
#pragma unroll
    for (int e = 0; e < NE; ++e) p[e] = __builtin_fma(t[e], p[e], 1.0);
#pragma unroll
    for (int e = 0; e < NE; ++e) p[e] = __builtin_fma(t[e], p[e], 1.0);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const double r = __builtin_ldexp(p[e], (int)dn[e]);
        z[e] = (x[e] < -1075.0) ? 0.0 : r;
    }
}

// Four squared-exponential elements (the four accumulator registers of one MFMA block) with the
// underflow shortcut: exp(a) is exactly +0 for a < -745.14, so when every lane of the wave is below
// -746 for all four the terms are +0 without evaluating exp (a wave-uniform branch; same bits as
// kern_elem).  With resolved spectra most 16 x 16 patches far from the band diagonals underflow.
template <int C>
__device__ __forceinline__ void kern_elem4_skip(const double (&xi)[4][C], const double (&xj)[C], const GpDev& g,
                                                double (&cov)[4])
{
#pragma clang fp contract(off)
#pragma unroll
    for (int c = 0; c < C; ++c) {
        double a[4], e[4];
        bool live = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double d = xj[c] - xi[r][c];
            a[r] = g.p2[c] * d * d;
            live = live || !(a[r] <= -746.0);
        }
        if (__builtin_amdgcn_ballot_w64(live) != 0ull) {
            exp_nonpos_batch<4>(a, e);
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = g.a2[c] * e[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) cov[r] = (c == 0) ? e[r] : cov[r] + e[r];
    }
}

// Symmetric fill of a batch of padded matrices.
//   Kbase + b*mat_stride : (Npad x ld) row-major; rows/cols >= N become identity
//   lwl  : (B, C, N) device;  gp : (B, 2C) device;  sigma : (N) device or nullptr
//   upper_only != 0: only tiles with tj >= ti are written (Cholesky input);
//   otherwise grid.x spans all P*P tiles (full symmetric matrix, fill_V11_* contract).
template <int C>
__global__ __launch_bounds__(256) void k_fill_sym(double* __restrict__ Kbase, size_t mat_stride, int ld, int N,
                                                  int P, const double* __restrict__ lwl,
                                                  const double* __restrict__ gp,
                                                  const double* __restrict__ sigma, int upper_only)
{
    __shared__ double xrow[C][NB];
    __shared__ double srow[NB];
    const int b = blockIdx.y;
    int ti, tj;
    if (upper_only) {
        decode_upper(blockIdx.x, P, ti, tj);
    } else {
        ti = blockIdx.x / P;
        tj = blockIdx.x % P;
    }
    const int tid = threadIdx.x;
    const int i0 = ti * NB, j0 = tj * NB;
    const double* lw = lwl + (size_t)b * C * N;
    GpDev g;
    load_gp(gp + (size_t)b * 2 * C, C, g);
    double dsum = g.a2[0];
    {
#pragma clang fp contract(off)
        for (int c = 1; c < C; ++c) dsum = dsum + g.a2[c];
    }

    if (tid < NB) {
        int i = i0 + tid;
#pragma unroll
        for (int c = 0; c < C; ++c) xrow[c][tid] = (i < N) ? lw[(size_t)c * N + i] : 0.0;
        double s = (sigma != nullptr && i < N) ? sigma[i] : 0.0;
        srow[tid] = s * s;
    }
    // Wave w owns the 32-column block w of the tile; per step it covers a 16 x 32 patch: lane = (row group
    // 0..3) x (column pair 0..15), four rows (rg + 4 s) and two columns per lane.  A compact patch is what
    // makes the underflow shortcut bite: the covariance is banded (|dx| beyond ~72 pixels gives exp() = +0
    // exactly), and a 16 x 32 patch misses the bands far more often than a 4 x 128 strip does (round 1's
    // shape: 4.3 TB/s at c = 2, bound by exp() throughput, not by the stores).  A store instruction writes
    // four 256-byte row segments.
    const int w = tid >> 6, rg = (tid >> 4) & 3, cp = tid & 15;
    const int ja = j0 + 32 * w + 2 * cp, jb = ja + 1;
    double xa[C], xb[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        xa[c] = (ja < N) ? lw[(size_t)c * N + ja] : 0.0;
        xb[c] = (jb < N) ? lw[(size_t)c * N + jb] : 0.0;
    }
    __syncthreads();

    double* Km = Kbase + (size_t)b * mat_stride;
    // Four rows x two columns per lane and step: the eight exponentials of a component go through one
    // batched, underflow-skipping evaluation.
#pragma unroll 1
    for (int g4 = 0; g4 < NB / 16; ++g4) {
        double va[4], vb[4];
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma clang fp contract(off)
            double a[8], e[8];
            bool live = false;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const double xi = xrow[c][16 * g4 + rg + 4 * s4];
                const double da = xa[c] - xi, db = xb[c] - xi;
                a[2 * s4] = g.p2[c] * da * da;
                a[2 * s4 + 1] = g.p2[c] * db * db;
                live = live || !(a[2 * s4] <= -746.0) || !(a[2 * s4 + 1] <= -746.0);
            }
            if (__builtin_amdgcn_ballot_w64(live) != 0ull) {
                exp_nonpos_batch<8>(a, e);
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = g.a2[c] * e[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) e[k] = 0.0;
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                va[s4] = (c == 0) ? e[2 * s4] : va[s4] + e[2 * s4];
                vb[s4] = (c == 0) ? e[2 * s4 + 1] : vb[s4] + e[2 * s4 + 1];
            }
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int r = 16 * g4 + rg + 4 * s4;
            const int i = i0 + r;
            d2 v;
            if (i < N) {
                double x = (ja < N) ? va[s4] : 0.0;
                double y = (jb < N) ? vb[s4] : 0.0;
                // diagonal rule (pyx:56-57,144,201) + noise (covariance.py:322)
                if (ja == i) x = dsum + srow[r];
                if (jb == i) y = dsum + srow[r];
                v.x = x;
                v.y = y;
            } else {
                v.x = (ja == i) ? 1.0 : 0.0;
                v.y = (jb == i) ? 1.0 : 0.0;
            }
            // (non-temporal stores measured slower here: 5.4 vs 5.75 TB/s at c = 1, no difference at c = 2, 3)
            *reinterpret_cast<d2*>(Km + (size_t)i * ld + ja) = v;
        }
    }
}

// Rectangular cross fill (pyx:61-96): out (M x ld), out[i,j] = a2 * exp(p2 * (xcol[j]-xrow[i])^2).
// grid.x = tiles over columns, grid.y = tiles over rows.
__global__ __launch_bounds__(256) void k_fill_cross(double* __restrict__ out, int ld, int M, int Ncol,
                                                    const double* __restrict__ xrow_g,
                                                    const double* __restrict__ xcol_g, double amp, double l)
{
    __shared__ double xrow[NB];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.y * NB, j0 = blockIdx.x * NB;
    GpDev g;
    double gp[2] = {amp, l};
    load_gp(gp, 1, g);
    if (tid < NB) xrow[tid] = (i0 + tid < M) ? xrow_g[i0 + tid] : 0.0;
    const int col2 = tid & 63, w = tid >> 6;
    const int ja = j0 + 2 * col2, jb = ja + 1;
    double xa[1], xb[1];
    xa[0] = (ja < Ncol) ? xcol_g[ja] : 0.0;
    xb[0] = (jb < Ncol) ? xcol_g[jb] : 0.0;
    __syncthreads();
    if (ja >= ld) return;
#pragma unroll 4
    for (int it = 0; it < NB / 4; ++it) {
        const int r = w + 4 * it;
        const int i = i0 + r;
        if (i >= M) break;
        double xi[1] = {xrow[r]};
        d2 v;
        v.x = (ja < Ncol) ? kern_elem<1>(xi, xa, g) : 0.0;
        v.y = (jb < Ncol) ? kern_elem<1>(xi, xb, g) : 0.0;
        *reinterpret_cast<d2*>(out + (size_t)i * ld + ja) = v;
    }
}

// Device-side Doppler shift (replicate_wls + lredshift, psoap/data.py:37,61):
//   lwl_out[b,c,i] = lwl[i] + (-vel[b,c,epoch[i]]) / c_kms
__global__ void k_doppler_shift(double* __restrict__ lwl_out, const double* __restrict__ lwl,
                                const int32_t* __restrict__ epoch, const double* __restrict__ vel, int N,
                                int n_epochs, int BC)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int bc = blockIdx.y;
    if (i < N && bc < BC) {
        double v = vel[(size_t)bc * n_epochs + epoch[i]];
        lwl_out[(size_t)bc * N + i] = lwl[i] + (-v) / C_KMS;
    }
}

}  // namespace psoap
