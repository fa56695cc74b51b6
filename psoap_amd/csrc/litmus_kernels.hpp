// litmus_kernels.hpp -- what a workgroup on one XCD reads of a line a workgroup on another XCD has just rewritten.
//
// Measurement only (libpsoap_bench.so, tools/litmus.py).  Round 6 needed the hardware's answer to three questions the
// persistent kernels' hand-offs rest on (dag_kernel.hpp; cdna_hip_programming.md Guideline 16):
//   1. can an agent-scope relaxed atomic LOAD (global_load ... sc1) be served from a copy the reader's own XCD L2 took
//      of the line BEFORE the other XCD's store -- and for how long;
//   2. does an agent-scope acquire (buffer_inv sc1) in front of the load change that, for sc1 loads, plain loads and
//      LDS-DMA loads;
//   3. is a returning atomic (the read-modify-write the polls were meant to be) always fresh.
// One READER wave and one WRITER wave (on another XCD, or on the same one) take turns on one 256-byte unit (two lines):
//   reader: (plant) touch the unit so that its XCD's L2 holds a copy -> tell the writer (sc1 store of `go`)
//   writer: poll `go` with a returning atomic -> store the unit's 32 words = i + 1 -> drain (vmcnt 0, a release fence
//           behind plain stores) -> returning atomic add on `done`
//   reader: poll `done` with a returning atomic -> read the unit in the mode under test -> stale if any word != i + 1;
//           if stale, keep reading in that mode and count the 100 MHz ticks until every word is fresh.
// The other workgroups of the launch either leave or stream a large buffer through the L2s (eviction pressure, as beside
// a factorisation).  Every access under test is inline assembly: the instruction is the one named.
#pragma once
#include <string>

#include "common.hpp"

namespace psoap {

struct LitmusOut {
    unsigned long long iters, plant_stale, read_stale, never_fresh, max_ticks, sum_ticks, reader_xcc, writer_xcc;
};

typedef unsigned long long u64;

__device__ __forceinline__ u64 lit_load_plain(const u64* p)
{
    u64 v;
    asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ u64 lit_load_sc1(const u64* p)
{
    u64 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ u64 lit_load_sc01(const u64* p)
{
    u64 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void lit_acquire()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// reader modes
enum : int { LIT_SC1 = 0, LIT_ACQ_SC1 = 1, LIT_ACQ_PLAIN = 2, LIT_RMW = 3, LIT_PLAIN = 4, LIT_ACQ_LDSDMA = 5, LIT_SC01 = 6, LIT_N_READ = 7 };
// writer modes: 0 sc1 stores, 1 plain stores + release fence, 2 sc0 sc1 stores
// plant modes: 0 none, 1 sc1 load, 2 plain load, 3 acquire + plain load

// one read of the unit in `mode`: true when every word this lane checks equals want (lanes >= 32 check nothing)
__device__ __forceinline__ bool lit_read_fresh(int mode, u64* unit, int lane, u64 want, unsigned int* lds)
{
    bool ok = true;
    if (mode == LIT_ACQ_SC1 || mode == LIT_ACQ_PLAIN || mode == LIT_ACQ_LDSDMA) lit_acquire();
    if (mode == LIT_ACQ_LDSDMA) {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* glb_ptr;
        // 64 lanes x 4 bytes = the unit's 64 dwords, lane l's dword lands at lds[l]
        __builtin_amdgcn_global_load_lds((glb_ptr)(reinterpret_cast<const unsigned int*>(unit) + lane), (lds_ptr)lds, 4, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int v = reinterpret_cast<volatile unsigned int*>(lds)[lane];
        ok = (lane & 1) ? (v == (unsigned int)(want >> 32)) : (v == (unsigned int)want);
    } else if (lane < 32) {
        u64 v;
        if (mode == LIT_SC1 || mode == LIT_ACQ_SC1) v = lit_load_sc1(unit + lane);
        else if (mode == LIT_SC01) v = lit_load_sc01(unit + lane);
        else if (mode == LIT_RMW) v = rmw_read(unit + lane);
        else v = lit_load_plain(unit + lane);
        ok = v == want;
    }
    return __all(ok);
}

__global__ __launch_bounds__(64) void k_litmus(u64* unit, unsigned int* sync, int plant_mode, int writer_mode, int reader_mode,
                                               int same_xcd, int iters, const double* bg, size_t bg_n, LitmusOut* out)
{
    // sync words, one 128-byte line each: [0..7] claims per XCD, [8] go, [9] done, [10] stop, [11] roles taken
    __shared__ unsigned int lds[64];
    const int lane = threadIdx.x;
    const unsigned int xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;
    int role = -1;
    if (lane == 0) {
        const unsigned int prev = atomicAdd(&sync[xcc * 32], 1u);
        if (xcc == 0 && prev == 0) role = 0;
        else if (!same_xcd && xcc == 1 && prev == 0) role = 1;
        else if (same_xcd && xcc == 0 && prev == 1) role = 1;
    }
    role = __builtin_amdgcn_readfirstlane(role);
    unsigned int* go = sync + 8 * 32;
    unsigned int* done = sync + 9 * 32;
    unsigned int* stop = sync + 10 * 32;
    if (role < 0) {
        // background: stream the buffer until told to stop
        if (bg_n == 0) return;
        double s = 0.0;
        for (int pass = 0; pass < 100000; ++pass) {
            const size_t chunk = bg_n / gridDim.x;
            const double* p = bg + (size_t)blockIdx.x * chunk;
            for (size_t i = lane; i < chunk; i += 64) s += p[i];
            unsigned int st = 0;
            if (lane == 0) st = rmw_read(stop);
            if (__builtin_amdgcn_readfirstlane((int)st) != 0) break;
        }
        if (s == 1.2345e300) out->iters = 0;
        return;
    }
    if (role == 1) {
        if (lane == 0) out->writer_xcc = xcc;
        for (int i = 0; i < iters; ++i) {
            int gave_up = 0;
            if (lane == 0) {
                long long spins = 0;
                while (rmw_read(go) != (unsigned int)(i + 1))
                    if (++spins > 4000000) { gave_up = 1; break; }      // (seconds: the other role is not there)
            }
            if (__builtin_amdgcn_readfirstlane(gave_up)) return;
            const u64 v = (u64)(i + 1) * 0x0000000100000001ull;      // both dwords carry i + 1
            if (lane < 32) {
                u64* p = unit + lane;
                if (writer_mode == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
                else if (writer_mode == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
                else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (writer_mode == 1) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (lane == 0) atomicAdd(done, 1u);
        }
        return;
    }
    // reader
    u64 plant_stale = 0, read_stale = 0, never = 0, max_t = 0, sum_t = 0;
    for (int i = 0; i < iters; ++i) {
        const u64 old = (u64)i * 0x0000000100000001ull, want = (u64)(i + 1) * 0x0000000100000001ull;
        if (plant_mode != 0) {
            bool ok = true;
            if (plant_mode == 3) lit_acquire();
            if (lane < 32) ok = (plant_mode == 1 ? lit_load_sc1(unit + lane) : lit_load_plain(unit + lane)) == old;
            if (!__all(ok)) ++plant_stale;
        }
        int gave_up = 0;
        if (lane == 0) {
            __hip_atomic_store(go, (unsigned int)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long spins = 0;
            while (rmw_read(done) != (unsigned int)(i + 1))
                if (++spins > 4000000) { gave_up = 1; break; }
        }
        if (__builtin_amdgcn_readfirstlane(gave_up)) { never = ~0ull; break; }
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        if (!lit_read_fresh(reader_mode, unit, lane, want, lds)) {
            ++read_stale;
            bool fresh = false;
            for (int k = 0; k < 4000 && !fresh; ++k) fresh = lit_read_fresh(reader_mode, unit, lane, want, lds);      // (~ 4 ms)
            const u64 dt = __builtin_amdgcn_s_memrealtime() - t0;
            if (!fresh) {
                ++never;
                // put the reader's view right again for the next round: a returning atomic on every word
                if (lane < 32) (void)rmw_read(unit + lane);
                lit_acquire();
            } else {
                max_t = dt > max_t ? dt : max_t;
                sum_t += dt;
            }
        }
    }
    if (lane == 0) {
        out->iters = (u64)iters;
        out->plant_stale = plant_stale;
        out->read_stale = read_stale;
        out->never_fresh = never;
        out->max_ticks = max_t;
        out->sum_ticks = sum_t;
        out->reader_xcc = xcc;
        __hip_atomic_store(stop, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


// Does the write-back of a line one XCD has PARTLY rewritten with plain stores carry the rest of that XCD's (older) copy
// of the line with it -- over what another XCD has written through to those other words in the meantime?  (The stream
// dispatcher clears a lane's flags, counters and hyper-parameters with plain stores, next to words of lanes in flight.)
//   A (XCD 0): plain-load all 16 words of the line (its L2 now holds the whole line) -> tell B
//   B (XCD 1): write words 1..15 = i + 1 through (sc1), drain -> tell A
//   A: plain-store word 0 = i + 1, release fence, drain -> tell B
//   B: read words 1..15 with returning atomics: any that fell back to i were overwritten by A's write-back.
__global__ __launch_bounds__(64) void k_litmus_wb(u64* line, unsigned int* sync, int iters, LitmusOut* out)
{
    const int lane = threadIdx.x;
    const unsigned int xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;
    int role = -1;
    if (lane == 0) {
        const unsigned int prev = atomicAdd(&sync[xcc * 32], 1u);
        if (xcc == 0 && prev == 0) role = 0;
        else if (xcc == 1 && prev == 0) role = 1;
    }
    role = __builtin_amdgcn_readfirstlane(role);
    if (role < 0) return;
    unsigned int* step = sync + 8 * 32;       // 3 i + 1: A has the line; 3 i + 2: B has written; 3 i + 3: A has written back
    auto wait_step = [&](unsigned int want) {
        int gave_up = 0;
        if (lane == 0) {
            long long spins = 0;
            while (rmw_read(step) < want)
                if (++spins > 4000000) { gave_up = 1; break; }
        }
        return __builtin_amdgcn_readfirstlane(gave_up) != 0;
    };
    u64 clobbered = 0, lost0 = 0;
    for (int i = 0; i < iters; ++i) {
        const u64 v = (u64)(i + 1);
        if (role == 0) {
            if (i > 0 && wait_step(3u * i)) return;
            lit_acquire();
            u64 x = 0;
            if (lane < 16) x = lit_load_plain(line + lane);
            if (x == 0x123456789abcdefull) out->iters = 0;
            if (lane == 0) atomicAdd(step, 1u);                     // -> 3 i + 1
            if (wait_step(3u * i + 2)) return;
            if (lane == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(line), "v"(v) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) atomicAdd(step, 1u);                     // -> 3 i + 3
        } else {
            if (wait_step(3u * i + 1)) return;
            if (lane >= 1 && lane < 16) {
                u64* p = line + lane;
                asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) atomicAdd(step, 1u);                     // -> 3 i + 2
            if (wait_step(3u * i + 3)) return;
            bool ok = true;
            if (lane < 16) ok = rmw_read(line + lane) == v;
            if (!__all(ok || lane == 0)) ++clobbered;
            if (!__all(ok || lane != 0)) ++lost0;
        }
    }
    if (role == 1 && lane == 0) {
        out->iters = (u64)iters;
        out->read_stale = clobbered;      // launches in which a word B wrote fell back
        out->plant_stale = lost0;         // ... in which A's own word did not arrive
        out->writer_xcc = xcc;
    }
}

inline int litmus_wb_run(int iters, unsigned long long* out8, std::string& err)
{
    u64* line = nullptr;
    unsigned int* sync = nullptr;
    LitmusOut* dout = nullptr;
#define LT_TRY(expr)                                                                                    \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            err = std::string(#expr) + " failed: " + hipGetErrorString(_e);                             \
            return 1;                                                                                   \
        }                                                                                               \
    } while (0)
    LT_TRY(hipMalloc(&line, 4096));
    LT_TRY(hipMalloc(&sync, 12 * 128));
    LT_TRY(hipMalloc(&dout, sizeof(LitmusOut)));
    LT_TRY(hipMemset(line, 0, 4096));
    LT_TRY(hipMemset(sync, 0, 12 * 128));
    LT_TRY(hipMemset(dout, 0, sizeof(LitmusOut)));
    LT_TRY(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_litmus_wb, dim3(64), dim3(64), 0, 0, line, sync, iters, dout);
    LT_TRY(hipGetLastError());
    LT_TRY(hipDeviceSynchronize());
    LitmusOut h;
    LT_TRY(hipMemcpy(&h, dout, sizeof h, hipMemcpyDeviceToHost));
    out8[0] = h.iters; out8[1] = h.plant_stale; out8[2] = h.read_stale; out8[3] = h.never_fresh;
    out8[4] = h.max_ticks; out8[5] = h.sum_ticks; out8[6] = h.reader_xcc; out8[7] = h.writer_xcc;
    (void)hipFree(line); (void)hipFree(sync); (void)hipFree(dout);
#undef LT_TRY
    return 0;
}

inline int litmus_run(int plant_mode, int writer_mode, int reader_mode, int same_xcd, int iters, int background,
                      unsigned long long* out8, std::string& err)
{
#define LT_TRY(expr)                                                                                    \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            err = std::string(#expr) + " failed: " + hipGetErrorString(_e);                             \
            return 1;                                                                                   \
        }                                                                                               \
    } while (0)
    u64* unit = nullptr;
    unsigned int* sync = nullptr;
    double* bg = nullptr;
    LitmusOut* dout = nullptr;
    const size_t bg_n = background ? ((size_t)1 << 26) : 0;      // 512 MiB streamed by the other workgroups
    LT_TRY(hipMalloc(&unit, 4096));
    LT_TRY(hipMalloc(&sync, 12 * 128));
    LT_TRY(hipMalloc(&dout, sizeof(LitmusOut)));
    if (bg_n) {
        LT_TRY(hipMalloc(&bg, sizeof(double) * bg_n));
        LT_TRY(hipMemset(bg, 0, sizeof(double) * bg_n));
    }
    LT_TRY(hipMemset(unit, 0, 4096));
    LT_TRY(hipMemset(sync, 0, 12 * 128));
    LT_TRY(hipMemset(dout, 0, sizeof(LitmusOut)));
    LT_TRY(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_litmus, dim3(background ? 1024 : 64), dim3(64), 0, 0, unit, sync, plant_mode, writer_mode, reader_mode,
                       same_xcd, iters, bg, bg_n, dout);
    LT_TRY(hipGetLastError());
    LT_TRY(hipDeviceSynchronize());
    LitmusOut h;
    LT_TRY(hipMemcpy(&h, dout, sizeof h, hipMemcpyDeviceToHost));
    out8[0] = h.iters; out8[1] = h.plant_stale; out8[2] = h.read_stale; out8[3] = h.never_fresh;
    out8[4] = h.max_ticks; out8[5] = h.sum_ticks; out8[6] = h.reader_xcc; out8[7] = h.writer_xcc;
    (void)hipFree(unit); (void)hipFree(sync); (void)hipFree(dout);
    if (bg) (void)hipFree(bg);
#undef LT_TRY
    return 0;
}

}  // namespace psoap
