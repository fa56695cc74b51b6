// psoap_bench.hip -- measurement kernels behind include/psoap_bench.h (libpsoap_bench.so).
// Not part of the product library: bench.py, tools/ and one GPU test load it to state measured ceilings
// (MFMA issue rate, streaming HBM bandwidth, the tile engine alone) beside the spec peaks, and to check the
// batched exp() of the fused-fill epilogue against the device library's bit for bit.
#include "../../include/psoap_bench.h"

#include <string>

#include "microbench_kernels.hpp"
#include "litmus_kernels.hpp"

using namespace psoap;

static thread_local std::string g_err;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            char _buf[512];                                                                        \
            snprintf(_buf, sizeof _buf, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                     __LINE__);                                                                    \
            g_err = _buf;                                                                          \
            return 1;                                                                              \
        }                                                                                          \
    } while (0)

#define FAIL(msg)                \
    do {                         \
        g_err = std::string(msg); \
        return 2;                \
    } while (0)

extern "C" const char* psoap_bench_last_error(void) { return g_err.c_str(); }

extern "C" int psoap_microbench_mfma_f64(int device, double* tflops)
{
    HIP_TRY(hipSetDevice(device));
    return microbench_mfma(tflops, g_err);
}

extern "C" int psoap_microbench_tile_engine(int device, int shared_operands, double* tflops)
{
    HIP_TRY(hipSetDevice(device));
    return microbench_tile_engine(shared_operands, tflops, g_err);
}

extern "C" int psoap_microbench_potrf(int device, int ablate, double* usec)
{
    HIP_TRY(hipSetDevice(device));
    return microbench_potrf(ablate, usec, g_err);
}

extern "C" int psoap_microbench_exp_check(int device, long long n, const double* x, long long* mismatches)
{
    if (n < 4 || !x || !mismatches) FAIL("psoap_microbench_exp_check: bad arguments");
    HIP_TRY(hipSetDevice(device));
    return microbench_exp_check(n, x, mismatches, g_err);
}

extern "C" int psoap_microbench_mix(int device, int mode, int iters_mfma, int iters_valu, double* out3)
{
    if (!out3 || iters_mfma < 0 || iters_valu < 0) FAIL("psoap_microbench_mix: bad arguments");
    HIP_TRY(hipSetDevice(device));
    return microbench_mix(mode, iters_mfma, iters_valu, out3, g_err);
}

extern "C" int psoap_microbench_hbm(int device, double* write_gbs, double* copy_gbs)
{
    HIP_TRY(hipSetDevice(device));
    return microbench_hbm(write_gbs, copy_gbs, g_err);
}

extern "C" int psoap_litmus_l2(int device, int plant_mode, int writer_mode, int reader_mode, int same_xcd, int iters,
                               int background, unsigned long long* out8)
{
    if (!out8 || iters < 1 || plant_mode < 0 || plant_mode > 3 || writer_mode < 0 || writer_mode > 2 || reader_mode < 0 ||
        reader_mode >= LIT_N_READ)
        FAIL("psoap_litmus_l2: bad arguments");
    HIP_TRY(hipSetDevice(device));
    return litmus_run(plant_mode, writer_mode, reader_mode, same_xcd, iters, background, out8, g_err);
}

extern "C" int psoap_litmus_writeback(int device, int iters, unsigned long long* out8)
{
    if (!out8 || iters < 1) FAIL("psoap_litmus_writeback: bad arguments");
    HIP_TRY(hipSetDevice(device));
    return litmus_wb_run(iters, out8, g_err);
}
