// microbench_kernels.hpp -- measured ceilings quoted next to the spec peaks in bench.py:
// back-to-back v_mfma_f64_16x16x4_f64 issue rate and streaming HBM write / copy bandwidth.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "common.hpp"
#include "fill_kernels.hpp"
#include "gemm_core.hpp"
#include "potrf_blocked.hpp"

namespace psoap {

// Every SIMD of every CU issues independent fp64 MFMAs from registers (4 accumulator
// chains per wave, 2 waves per SIMD).  2048 flop per instruction.
__global__ __launch_bounds__(512) void k_mfma_f64_peak(double* out, int iters)
{
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    d4 acc0 = {0, 0, 0, 0}, acc1 = {1, 1, 1, 1}, acc2 = {2, 2, 2, 2}, acc3 = {3, 3, 3, 3};
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
    }
    d4 s = acc0 + acc1 + acc2 + acc3;
    if (s[0] + s[1] + s[2] + s[3] == 12345.678) out[0] = s[0];  // keep the chain alive
}

__global__ void k_stream_write(d2* __restrict__ dst, size_t n2, double v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    d2 val = {v, v};
    for (; i < n2; i += stride) dst[i] = val;
}

// store-path variants for finding the streaming-write ceiling: 0 plain, 1 nontemporal, 2 plain with
// 4 stores in flight per thread per iteration
template <int V>
__global__ void k_stream_write_v(d2* __restrict__ dst, size_t n2, double v)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    d2 val = {v, v};
    if (V == 2) {
        for (; i + 3 * stride < n2; i += 4 * stride) {
            dst[i] = val;
            dst[i + stride] = val;
            dst[i + 2 * stride] = val;
            dst[i + 3 * stride] = val;
        }
        for (; i < n2; i += stride) dst[i] = val;
    } else {
        for (; i < n2; i += stride) {
            if (V == 1) __builtin_nontemporal_store(val, &dst[i]);
            else dst[i] = val;
        }
    }
}

// narrower stores: W = 1 one dword per lane (256 B per wave-instruction, the shape MI355X_MICROARCH.md quotes 6.0-6.2 TB/s for),
// W = 2 one double per lane; each workgroup writes contiguous 1 KiB / 2 KiB pieces
template <int W>
__global__ void k_stream_write_narrow(unsigned int* __restrict__ dst, size_t n_dwords, unsigned int v)
{
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * W;
    const size_t stride = (size_t)gridDim.x * blockDim.x * W;
    for (; i + W <= n_dwords; i += stride) {
        if (W == 1) dst[i] = v;
        else *reinterpret_cast<uint2*>(dst + i) = make_uint2(v, v);
    }
}

__global__ void k_stream_copy(d2* __restrict__ dst, const d2* __restrict__ src, size_t n2)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n2; i += stride) dst[i] = src[i];
}

// Tile-engine ceiling: every workgroup runs tile_gemm_tn over K rows.  shared_operands = 1: all
// workgroups stream the same two strips (L2-resident after the first pass); 0: every workgroup
// streams its own B strip from HBM and shares the A strip with the 31 others of its group.
template <int ENG>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_tile_engine_bench(const double* __restrict__ A,
                                                                      const double* __restrict__ Bm, size_t ld, int K,
                                                                      int shared_operands, double* sink)
{
    Tile t;
    t.zero();
    // shared_operands & 2: give the two workgroups that share a CU different issue priorities
    const size_t acol = shared_operands ? 0 : (size_t)(blockIdx.x / 32) * NB;
    const size_t bcol = shared_operands ? NB : (size_t)(blockIdx.x % ((int)(ld / NB))) * NB;
    if (ENG == 1) tile_gemm_tn(t, A + acol, ld, Bm + bcol, ld, K);
    else tile_gemm_tn_reg(t, A + acol, ld, Bm + bcol, ld, K);
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) s += t.acc[m][n][0] + t.acc[m][n][1] + t.acc[m][n][2] + t.acc[m][n][3];
    if (s == 12345.678) sink[0] = s;
}

template <int ABLATE>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_potrf_bench(double* Km, int ld, double* Wm, double* Rv, MatAcc* acc,
                                                                const double* K0, int reps)
{
    for (int it = 0; it < reps; ++it) {
        for (int i = threadIdx.x; i < NB * NB; i += GEMM_THREADS) Km[(size_t)(i / NB) * ld + i % NB] = K0[i];
        if (threadIdx.x < NB) Rv[threadIdx.x] = 1.0;
        __syncthreads();
        potrf_blocked<ABLATE>(Km, ld, 0, Wm, Rv, acc);
        __syncthreads();
    }
}

// Ablations of the tile-engine loop (diagnostics): ABL bit 0 = no global loads / LDS stores after the
// first stage, bit 1 = no workgroup barrier, bit 2 = no LDS fragment reads (MFMAs on stale registers).
template <int ABL>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_tile_engine_ablate(const double* __restrict__ A, size_t ld, int K,
                                                                       double* sink)
{
    Tile t;
    t.zero();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const double* Ab = A + (size_t)(blockIdx.x / 32) * NB;
    const double* Bb = A + (size_t)(blockIdx.x % ((int)(ld / NB))) * NB;
    Staging s;
    stage_load(s, Ab, ld, Bb, ld, 0, tid);
    stage_store(s, 0, tid);
    stage_store(s, 1, tid);
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4;
    double a[4] = {1.0, 2.0, 3.0, 4.0}, b[4] = {1.5, 2.5, 3.5, 4.5};
    const int nchunk = K / KB;
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        if (!(ABL & 1)) stage_load(s, Ab, ld, Bb, ld, ((c + 1) % nchunk) * KB, tid);
        if (ABL & 4) {
#pragma unroll
            for (int ks = 0; ks < KB / 4; ++ks)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        t.acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], t.acc[m][n], 0, 0, 0);
        } else {
            tile_mma_chunk(t, cur, wr, wc, lane);
        }
        if (!(ABL & 1)) stage_store(s, cur ^ 1, tid);
        if (!(ABL & 2)) __syncthreads();
    }
    double sum = a[0] * fr + b[0] * fk;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) sum += t.acc[m][n][0] + t.acc[m][n][1] + t.acc[m][n][2] + t.acc[m][n][3];
    if (sum == 12345.678) sink[0] = sum;
}

#define MB_TRY(expr)                                                              \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            err = std::string(#expr) + ": " + hipGetErrorString(_e);              \
            return 1;                                                             \
        }                                                                         \
    } while (0)

inline int microbench_mfma(double* tflops, std::string& err)
{
    hipDeviceProp_t prop;
    int dev = 0;
    MB_TRY(hipGetDevice(&dev));
    MB_TRY(hipGetDeviceProperties(&prop, dev));
    const int blocks = prop.multiProcessorCount * 2;  // 2 x 512 threads = 4 waves per SIMD
    const int iters = 20000;
    double* d = nullptr;
    MB_TRY(hipMalloc(&d, 64));
    hipEvent_t e0, e1;
    MB_TRY(hipEventCreate(&e0));
    MB_TRY(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_mfma_f64_peak, dim3(blocks), dim3(512), 0, 0, d, 1000);  // warm-up
    MB_TRY(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_mfma_f64_peak, dim3(blocks), dim3(512), 0, 0, d, iters);
    MB_TRY(hipEventRecord(e1, 0));
    MB_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    MB_TRY(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)blocks * 8 /*waves*/ * iters * 4.0 * 2048.0;
    *tflops = flops / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d);
    return 0;
}

inline int microbench_tile_engine(int shared_operands, double* tflops, std::string& err)
{
    hipDeviceProp_t prop;
    int dev = 0;
    MB_TRY(hipGetDevice(&dev));
    MB_TRY(hipGetDeviceProperties(&prop, dev));
    // + 64: ONE workgroup per compute unit (what a workgroup achieves while its neighbour is outside its K-loop)
    const int blocks = prop.multiProcessorCount * ((shared_operands & 64) ? 1 : 2);
    shared_operands &= ~64;
    const int K = 4096;
    const size_t ld = (size_t)blocks * NB;          // one 128-column strip per workgroup
    double* M = nullptr;
    MB_TRY(hipMalloc(&M, sizeof(double) * ld * K));
    MB_TRY(hipMemset(M, 0, sizeof(double) * ld * K));
    hipLaunchKernelGGL(k_stream_write, dim3(2048), dim3(256), 0, 0, reinterpret_cast<d2*>(M), ld * K / 2, 1.0e-3);
    MB_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_engine_bench<0>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES));
    MB_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_engine_bench<1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES));
    hipEvent_t e0, e1;
    MB_TRY(hipEventCreate(&e0));
    MB_TRY(hipEventCreate(&e1));
    if (shared_operands >= 16) {   // 16 + ABL: loop ablations
        const int abl = shared_operands - 16;
#define PSOAP_TE(AB)                                                                                          \
    MB_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_engine_ablate<AB>),                        \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES));              \
    hipLaunchKernelGGL(k_tile_engine_ablate<AB>, dim3(blocks), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, M, ld, 256, M); \
    MB_TRY(hipEventRecord(e0, 0));                                                                            \
    hipLaunchKernelGGL(k_tile_engine_ablate<AB>, dim3(blocks), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, M, ld, K, M);   \
    MB_TRY(hipEventRecord(e1, 0));
        switch (abl) {
            case 0: { PSOAP_TE(0) } break;
            case 1: { PSOAP_TE(1) } break;
            case 2: { PSOAP_TE(2) } break;
            case 3: { PSOAP_TE(3) } break;
            case 4: { PSOAP_TE(4) } break;
            case 5: { PSOAP_TE(5) } break;
            case 6: { PSOAP_TE(6) } break;
            default: { PSOAP_TE(7) } break;
        }
#undef PSOAP_TE
    } else {
    if (shared_operands >= 8) {   // 8, 9: LDS-DMA staging
    hipLaunchKernelGGL(k_tile_engine_bench<1>, dim3(blocks), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, M, M, ld, 256,
                       shared_operands & 1, M);
    MB_TRY(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_tile_engine_bench<1>, dim3(blocks), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, M, M, ld, K,
                       shared_operands & 1, M);
    MB_TRY(hipEventRecord(e1, 0));
    } else {
    hipLaunchKernelGGL(k_tile_engine_bench<0>, dim3(blocks), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, M, M, ld, 256,
                       shared_operands, M);
    MB_TRY(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_tile_engine_bench<0>, dim3(blocks), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, M, M, ld, K,
                       shared_operands, M);
    MB_TRY(hipEventRecord(e1, 0));
    }
    }
    MB_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    MB_TRY(hipEventElapsedTime(&ms, e0, e1));
    *tflops = 2.0 * NB * NB * (double)K * blocks / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(M);
    return 0;
}

// exp_nonpos_batch against the device library's exp(), bit for bit (the fused-fill epilogue relies on it)
__global__ void k_exp_check(const double* __restrict__ x, long long n, unsigned long long* mismatches)
{
    const long long i0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i0 + 3 >= n) return;
    double a[4], z[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] = x[i0 + e];
    exp_nonpos_batch<4>(a, z);
    unsigned long long bad = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double want = exp(a[e]);
        bad += (__double_as_longlong(want) != __double_as_longlong(z[e])) ? 1ull : 0ull;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// How fp64 MFMAs and fp64 vector arithmetic of two waves on one SIMD share the machine: one 512-thread
// workgroup per compute unit, waves 0-3 issue independent MFMAs (bit 0 of mode), waves 4-7 evaluate the
// epilogue's batched exp (bit 1).  Every wave stamps its own start / end (100 MHz counter) and its SIMD id.
__global__ __launch_bounds__(512) void k_mfma_valu_mix(int mode, int iters_m, int iters_e, unsigned long long* stamps,
                                                       double* sink)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double keep = 0.0;
    if ((mode & 32) && wave >= 4) __builtin_amdgcn_s_setprio(3);    // vector waves at raised priority
    if (wave < 4) {
        if (mode & 1) {
            const double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
            d4 acc0 = {0, 0, 0, 0}, acc1 = {1, 1, 1, 1}, acc2 = {2, 2, 2, 2}, acc3 = {3, 3, 3, 3};
            for (int i = 0; i < iters_m; ++i) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
            }
            d4 s4 = acc0 + acc1 + acc2 + acc3;
            keep = s4[0] + s4[1] + s4[2] + s4[3];
        }
    } else if (mode & 8) {       // fp32 FMA chains: 4 chains x 23 + 8, the instruction count of one exp batch
        float x[4] = {1e-3f * threadIdx.x, 0.5f, 3.0f, 40.0f}, z[4];
        for (int i = 0; i < iters_e; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = __builtin_fmaf(x[e], 0.999f, -1e-4f);
#pragma unroll
            for (int k = 0; k < 22; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = __builtin_fmaf(z[e], 0.999f, -1e-4f);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = x[e] * 0.9999f - 1e-6f * z[e];
        }
        keep = x[0] + x[1] + x[2] + x[3];
    } else if (mode & 16) {      // 32-bit integer multiply-add chains, same count
        unsigned int x[4] = {threadIdx.x, 5u, 3u, 40u}, z[4];
        for (int i = 0; i < iters_e; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = x[e] * 2654435761u + 12345u;
#pragma unroll
            for (int k = 0; k < 22; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = (z[e] ^ (z[e] >> 7)) + 0x9e3779b9u;
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = x[e] + z[e];
        }
        keep = (double)(x[0] + x[1] + x[2] + x[3]);
    } else if (mode & 2) {
        double x[4] = {-1e-3 * threadIdx.x, -0.5 - 1e-3 * threadIdx.x, -3.0 - 1e-3 * threadIdx.x, -40.0 - 1e-3 * threadIdx.x};
        double z[4];
        for (int i = 0; i < iters_e; ++i) {
            if (mode & 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) z[e] = __builtin_fma(x[e], 0.999, -1e-4);   // plain fp64 FMAs instead
#pragma unroll
                for (int k = 0; k < 22; ++k)
#pragma unroll
                    for (int e = 0; e < 4; ++e) z[e] = __builtin_fma(z[e], 0.999, -1e-4);
            } else {
                exp_nonpos_batch<4>(x, z);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = x[e] * 0.9999 - 1e-6 * z[e];
        }
        keep = x[0] + x[1] + x[2] + x[3];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (keep == 12345.678) sink[0] = keep;
    if ((threadIdx.x & 63) == 0) {
        const unsigned int simd = __builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));   // HW_ID.SIMD_ID
        stamps[((size_t)blockIdx.x * 8 + wave) * 2 + 0] = t1 - t0;
        stamps[((size_t)blockIdx.x * 8 + wave) * 2 + 1] = simd;
    }
}

// out[0] = mean MFMA-wave time (us), out[1] = mean exp-wave time (us), out[2] = fraction of workgroups whose
// waves w and w + 4 report the same SIMD
inline int microbench_mix(int mode, int iters_m, int iters_e, double* out, std::string& err)
{
    hipDeviceProp_t prop;
    int dev = 0;
    MB_TRY(hipGetDevice(&dev));
    MB_TRY(hipGetDeviceProperties(&prop, dev));
    const int blocks = prop.multiProcessorCount;
    unsigned long long* d = nullptr;
    double* sink = nullptr;
    MB_TRY(hipMalloc(&d, sizeof(unsigned long long) * blocks * 16));
    MB_TRY(hipMalloc(&sink, sizeof(double)));
    std::vector<unsigned long long> h((size_t)blocks * 16);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_mfma_valu_mix, dim3(blocks), dim3(512), 0, 0, mode, iters_m, iters_e, d, sink);
        MB_TRY(hipGetLastError());
        MB_TRY(hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    }
    double tm = 0, te = 0, same = 0;
    for (int b = 0; b < blocks; ++b) {
        bool s = true;
        for (int w = 0; w < 4; ++w) {
            tm += (double)h[((size_t)b * 8 + w) * 2] / 100.0;
            te += (double)h[((size_t)b * 8 + w + 4) * 2] / 100.0;
            s = s && h[((size_t)b * 8 + w) * 2 + 1] == h[((size_t)b * 8 + w + 4) * 2 + 1];
        }
        same += s ? 1.0 : 0.0;
    }
    out[0] = tm / (4.0 * blocks);
    out[1] = te / (4.0 * blocks);
    out[2] = same / blocks;
    (void)hipFree(d);
    (void)hipFree(sink);
    return 0;
}

inline int microbench_exp_check(long long n, const double* x, long long* mismatches, std::string& err)
{
    double* dx = nullptr;
    unsigned long long* dm = nullptr;
    unsigned long long hm = 0;
    n -= n % 4;
    MB_TRY(hipMalloc(&dx, sizeof(double) * n));
    MB_TRY(hipMalloc(&dm, sizeof(unsigned long long)));
    MB_TRY(hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
    MB_TRY(hipMemset(dm, 0, sizeof(unsigned long long)));
    hipLaunchKernelGGL(k_exp_check, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, 0, dx, n, dm);
    MB_TRY(hipGetLastError());
    MB_TRY(hipMemcpy(&hm, dm, sizeof(hm), hipMemcpyDeviceToHost));
    (void)hipFree(dx);
    (void)hipFree(dm);
    *mismatches = (long long)hm;
    return 0;
}

inline int microbench_potrf(int ablate, double* usec, std::string& err)
{
    std::vector<double> K0(NB * NB);
    for (int i = 0; i < NB; ++i)
        for (int j = 0; j < NB; ++j) K0[i * NB + j] = (i == j ? 2.0 : 0.0) + 1.0 / (1.0 + (i > j ? i - j : j - i));
    double *dK0 = nullptr, *dK = nullptr, *dW = nullptr, *dR = nullptr;
    MatAcc* dAcc = nullptr;
    MB_TRY(hipMalloc(&dK0, sizeof(double) * NB * NB));
    MB_TRY(hipMalloc(&dK, sizeof(double) * NB * NB));
    MB_TRY(hipMalloc(&dW, sizeof(double) * NB * NB));
    MB_TRY(hipMalloc(&dR, sizeof(double) * NB));
    MB_TRY(hipMalloc(&dAcc, sizeof(MatAcc)));
    MB_TRY(hipMemcpy(dK0, K0.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice));
    MB_TRY(hipMemset(dW, 0, sizeof(double) * NB * NB));
    MB_TRY(hipMemset(dAcc, 0, sizeof(MatAcc)));
    const int reps = 200;
    hipEvent_t e0, e1;
    MB_TRY(hipEventCreate(&e0));
    MB_TRY(hipEventCreate(&e1));
#define PSOAP_PB(AB)                                                                                              \
    MB_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_potrf_bench<AB>),                                   \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES));                  \
    hipLaunchKernelGGL(k_potrf_bench<AB>, dim3(1), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, dK, NB, dW, dR, dAcc, dK0, 2); \
    MB_TRY(hipEventRecord(e0, 0));                                                                                \
    hipLaunchKernelGGL(k_potrf_bench<AB>, dim3(1), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, dK, NB, dW, dR, dAcc, dK0, reps); \
    MB_TRY(hipEventRecord(e1, 0));
    if (ablate == 0) { PSOAP_PB(0) } else if (ablate == 1) { PSOAP_PB(1) } else if (ablate == 2) { PSOAP_PB(2) } else if (ablate == 3) { PSOAP_PB(3) } else { PSOAP_PB(9) }
#undef PSOAP_PB
    MB_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    MB_TRY(hipEventElapsedTime(&ms, e0, e1));
    *usec = 1e3 * ms / reps;
    if (ablate == 9) {
        unsigned long long st[16 * 6];
        MB_TRY(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_potrf_stamps), sizeof st));
        for (int w = 0; w < 2; ++w)
            for (int bb = 0; bb < 8; ++bb) {
                const unsigned long long* p = st + (w * 8 + bb) * 6;
                printf("wave %d step %d: A %5llu | wait1 %5llu | B %5llu | wait2 %5llu | C %5llu cycles\n", w, bb,
                       p[1] - p[0], p[2] - p[1], p[3] - p[2], p[4] - p[3], p[5] - p[4]);
            }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(dK0); (void)hipFree(dK); (void)hipFree(dW); (void)hipFree(dR); (void)hipFree(dAcc);
    return 0;
}

inline int microbench_hbm(double* write_gbs, double* copy_gbs, std::string& err)
{
    const size_t bytes = (size_t)2 << 30;  // 2 GiB per buffer: far beyond the 256 MiB Infinity Cache
    d2 *a = nullptr, *b = nullptr;
    MB_TRY(hipMalloc(&a, bytes));
    MB_TRY(hipMalloc(&b, bytes));
    const size_t n2 = bytes / sizeof(d2);
    hipEvent_t e0, e1;
    MB_TRY(hipEventCreate(&e0));
    MB_TRY(hipEventCreate(&e1));
    float ms = 0.f;
    hipLaunchKernelGGL(k_stream_write, dim3(2048), dim3(256), 0, 0, a, n2, 1.0);
    // streaming-write ceiling: best of a few grid sizes / store flavours (it varies 3.9-5.3 TB/s with the
    // number of workgroups on this part; many short workgroups win)
    *write_gbs = 0.0;
    for (int variant = 0; variant < 2; ++variant)
        for (int g : {2048, 16384}) {
            MB_TRY(hipEventRecord(e0, 0));
            for (int r = 0; r < 5; ++r) {
                if (variant == 0) hipLaunchKernelGGL(k_stream_write_v<0>, dim3(g), dim3(256), 0, 0, b, n2, 2.0);
                else hipLaunchKernelGGL(k_stream_write_v<1>, dim3(g), dim3(256), 0, 0, b, n2, 2.0);
            }
            MB_TRY(hipEventRecord(e1, 0));
            MB_TRY(hipEventSynchronize(e1));
            MB_TRY(hipEventElapsedTime(&ms, e0, e1));
            const double gbs = 5.0 * bytes / (ms * 1e-3) / 1e9;
            if (gbs > *write_gbs) *write_gbs = gbs;
        }
    // (8 bytes per lane is the fastest store width on this part: PSOAP_WRITE_SWEEP below)
    for (int g : {16384, 65536}) {
        MB_TRY(hipEventRecord(e0, 0));
        for (int r = 0; r < 5; ++r)
            hipLaunchKernelGGL(k_stream_write_narrow<2>, dim3(g), dim3(256), 0, 0, (unsigned int*)b, bytes / 4, 7u);
        MB_TRY(hipEventRecord(e1, 0));
        MB_TRY(hipEventSynchronize(e1));
        MB_TRY(hipEventElapsedTime(&ms, e0, e1));
        const double gbs = 5.0 * bytes / (ms * 1e-3) / 1e9;
        if (gbs > *write_gbs) *write_gbs = gbs;
    }
    if (getenv("PSOAP_WRITE_SWEEP")) {
        const int grids[] = {1024, 2048, 4096, 8192, 16384};
        for (int variant = 0; variant < 3; ++variant)
            for (int g : grids) {
                MB_TRY(hipEventRecord(e0, 0));
                for (int r = 0; r < 5; ++r) {
                    if (variant == 0) hipLaunchKernelGGL(k_stream_write_v<0>, dim3(g), dim3(256), 0, 0, b, n2, 2.0);
                    else if (variant == 1) hipLaunchKernelGGL(k_stream_write_v<1>, dim3(g), dim3(256), 0, 0, b, n2, 2.0);
                    else hipLaunchKernelGGL(k_stream_write_v<2>, dim3(g), dim3(256), 0, 0, b, n2, 2.0);
                }
                MB_TRY(hipEventRecord(e1, 0));
                MB_TRY(hipEventSynchronize(e1));
                MB_TRY(hipEventElapsedTime(&ms, e0, e1));
                printf("write variant %d grid %5d: %.0f GB/s\n", variant, g, 5.0 * bytes / (ms * 1e-3) / 1e9);
            }
        for (int w = 1; w <= 2; ++w)
            for (int g : {2048, 4096, 8192, 16384, 65536}) {
                MB_TRY(hipEventRecord(e0, 0));
                for (int r = 0; r < 5; ++r) {
                    if (w == 1) hipLaunchKernelGGL(k_stream_write_narrow<1>, dim3(g), dim3(256), 0, 0, (unsigned int*)b, bytes / 4, 7u);
                    else hipLaunchKernelGGL(k_stream_write_narrow<2>, dim3(g), dim3(256), 0, 0, (unsigned int*)b, bytes / 4, 7u);
                }
                MB_TRY(hipEventRecord(e1, 0));
                MB_TRY(hipEventSynchronize(e1));
                MB_TRY(hipEventElapsedTime(&ms, e0, e1));
                printf("write %d dword(s) per lane grid %5d: %.0f GB/s\n", w, g, 5.0 * bytes / (ms * 1e-3) / 1e9);
            }
    }
    MB_TRY(hipEventRecord(e0, 0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_stream_copy, dim3(2048), dim3(256), 0, 0, b, a, n2);
    MB_TRY(hipEventRecord(e1, 0));
    MB_TRY(hipEventSynchronize(e1));
    MB_TRY(hipEventElapsedTime(&ms, e0, e1));
    *copy_gbs = 5.0 * 2.0 * bytes / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    return 0;
}

}  // namespace psoap
