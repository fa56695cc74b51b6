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

// What the factorisation of ONE diagonal block contributes to lnprob.  A matrix owns ACC_ROWS of these records, one per
// block row (record q is written by the task that factors diagonal block q and by nobody else); whoever reports the
// result adds them up in the order q = 0, 1, ..., P-1 (acc_total).
// Rounds 1-5 kept ONE record per matrix that every diagonal task read, added to and wrote back -- a read-modify-write
// chain whose only ordering was the dependency of diagonal task q+1 on diagonal task q.  The second level of following
// (scheme 2, round 3) removed that dependency: task q+1 follows the strip solve that follows the steps of task q and
// never waits for anything task q publishes AFTER its accumulator update, so nothing ordered the two updates but time
// (task q+1 still has a block to factor, ~40 us) -- and with the records of four matrices in one 128-byte line a workgroup
// of another XCD could read the line for a neighbouring matrix (the result report of a stream lane), leave a snapshot of
// it in that XCD's L2, and a diagonal task that XCD ran next (a stolen one: its own lane had just finished) started from
// that snapshot: exactly one block row's contribution missing from lnprob -- the wrong value (2-3 % low) once in 20,000
// (N = 8192) ... 80,000 (N = 4096) matrices through a resident launch with the following scheme that round 5 found and
// fenced (profiles/r5_stream_scheme2.txt; every large mismatch there equals -0.5 (z_q^T z_q + 2 sum log U_ii) of ONE block
// q in {1, 29, 30, 31} to the last digit: profiles/r6_acc_forensics.txt).  No chain, no shared line: nothing to order.
struct MatAcc {
    double logdet_half;  // sum_i log U_ii over the block's 128 pivots   (logdet K = 2 * the sum over blocks)
    double quad;         // z_q^T z_q, z = U^-T r
    double info;         // != 0  ->  a pivot was <= 0 or NaN (not positive definite)
    double pad;
};
constexpr int ACC_ROWS = 256;      // records per matrix (block rows are 8-bit indices): 8 KiB, a whole number of 128-byte lines
static_assert(sizeof(MatAcc) == 32 && (ACC_ROWS * sizeof(MatAcc)) % 128 == 0, "a matrix's records share no cache line with another's");

// the record of diagonal block k0 / NB, written through (thread 0 of the factoring workgroup, or of a staged kernel)
__device__ __forceinline__ void acc_store(MatAcc* acc, int q, double logdet_half, double quad, bool bad)
{
    __hip_atomic_store(&acc[q].logdet_half, logdet_half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&acc[q].quad, quad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&acc[q].info, bad ? 1.0 : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Reading a word that other workgroups write (a progress flag, a ticket counter): the value the MEMORY holds now.
// Every poll of the persistent kernels goes through here.  Rounds 1-5 wrote their polls as `__hip_atomic_fetch_add(p, 0)`
// on the understanding that a read-modify-write executes at the memory side -- and hipcc turned every one of them into a
// load: LLVM folds an atomic read-modify-write whose operand is the operation's identity (add 0, or 0) and whose ordering
// is relaxed into an ATOMIC LOAD (`flat_load_dword ... sc1`, 2868 of them in the round-5 binary), which the XCD's L2 may
// serve.  With -DPSOAP_RMW_POLL the zero passes through an opaque statement, the fold cannot happen and the poll is the
// returning atomic add the design asked for (round 6: the A/B that decides which of the two the streams need).
template <class T>
__device__ __forceinline__ T rmw_read(T* p)      // always the returning atomic add (executed at the memory side, past every L2)
{
    T z = 0;
    asm volatile("" : "+v"(z));
    return __hip_atomic_fetch_add(p, z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T>
__device__ __forceinline__ T poll_word(T* p)
{
#ifdef PSOAP_RMW_POLL
    return rmw_read(p);
#else
    return __hip_atomic_fetch_add(p, (T)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

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
