// common.hpp -- shared types and constants for the gfx950 GP-likelihood kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psoap {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// psoap/matrix_functions.pyx:16-17 (reference tree): c_kms and c_kms**2
constexpr double C_KMS = 2.99792458e5;

constexpr int NB = 128;       // panel width == tile edge of every blocked kernel
constexpr int KB = 16;        // k-rows staged per LDS buffer in the MFMA tile loop
constexpr int LDS_LD = 144;   // LDS row stride (doubles): 128 + 16 keeps ds_read_b64 fragment
                              // reads of two consecutive k-rows on disjoint bank halves
constexpr int GEMM_THREADS = 256;
constexpr size_t GEMM_LDS_BYTES = (size_t)2 /*operands*/ * 2 /*buffers*/ * KB * LDS_LD * sizeof(double);

// per-matrix accumulator record kept in device memory
struct MatAcc {
    double logdet_half;  // sum_i log U_ii   (logdet K = 2 * this)
    double quad;         // z^T z, z = U^-T r
    double info;         // != 0  ->  a pivot was <= 0 or NaN (not positive definite)
    double pad;
};

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// row-major upper-triangle tile index t in [0, P(P+1)/2)  ->  (ti, tj), tj >= ti
__device__ inline void decode_upper(int t, int P, int& ti, int& tj)
{
    // row ti starts at offset ti*P - ti*(ti-1)/2
    double twoP1 = 2.0 * P + 1.0;
    int r = (int)((twoP1 - sqrt(twoP1 * twoP1 - 8.0 * (double)t)) * 0.5);
    if (r < 0) r = 0;
    if (r > P - 1) r = P - 1;
    while (r > 0 && r * P - r * (r - 1) / 2 > t) --r;
    while ((r + 1) * P - (r + 1) * r / 2 <= t) ++r;
    ti = r;
    tj = r + (t - (r * P - r * (r - 1) / 2));
}

}  // namespace psoap
