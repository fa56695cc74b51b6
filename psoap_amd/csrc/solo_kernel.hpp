// solo_kernel.hpp -- many small matrices: ONE workgroup factors ONE matrix, start to finish.
//
// The dependency-graph kernel (dag_kernel.hpp) spreads every matrix over many workgroups and pays for it in hand-offs:
// progress words, release / acquire fences, partial tiles through HBM, workgroups that hold a ticket and wait.  At the
// reference's real sizes -- chunks of ~80 pixels x 10-20 epochs, N = 800 ... 2000 (scripts/psoap_generate_chunks.py:8-9,
// 73-96), dozens of chunks x the walkers of an ensemble step -- there are hundreds of matrices per step and each is small:
// N = 2000, 32 walkers reach 0.455 of the fp64 peak through the graph (K-loops of 1-15 panels, 37 % of the worker time
// outside them), and nothing in its scheduling moves that (profiles/r5_experiments.txt).  With as many matrices as
// workgroup slots the parallelism is BETWEEN the matrices: a workgroup takes a matrix from a ticket counter and runs the
// left-looking tile Cholesky of DESIGN.md 3 on it alone --
//     for block row q:   tile (q, q): update over rows < q, covariance on the fly, in-block Cholesky (potrf_blocked)
//                        tiles (q, j > q): update, covariance, strip solve with W = U11^-T, r_j -= X^T z_q
// -- in program order.  No flags, no fences, no atomics but the ticket: every dependency is a workgroup barrier behind a
// drained store queue (the vector L1 is shared by the workgroup's waves: LLVM's workgroup scope).  What the workgroup
// loses while it factors a diagonal block or evaluates exp() its neighbour on the compute unit (another matrix) gains:
// the fp64 pipe is shared by the two.  Same tile engine, same store routine, same in-block factorisation, same strip
// solve as the graph's tasks; tiles are stored plainly (dag_st<false>): nobody else reads them.
// Order of summation: every tile's update is ONE K-loop over all rows above (no split), so a value differs from the
// graph's in the last bits like one scheme's from another's, and is the same for every batch size.
#pragma once
#include "dag_kernel.hpp"

namespace psoap {

struct SoloCtl {
    unsigned int next;      // next matrix to hand out
    unsigned int pad[31];
};

__device__ __forceinline__ void solo_sync()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

template <int C>
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_chol_solo(const DagMat* __restrict__ mats, const unsigned int* __restrict__ order,
                                                              int n_mats, SoloCtl* ctl)
{
    __shared__ double vec1[NB];
    __shared__ double vec2[NB];
    __shared__ unsigned int s_ticket;
    const int wave_s = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    for (;;) {
        if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&ctl->next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned int ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)s_ticket);
        __syncthreads();
        if (ticket >= (unsigned int)n_mats) return;
        // (largest matrices first: the launch ends with the short ones)
        const int b = __builtin_amdgcn_readfirstlane((int)(order ? order[ticket] : ticket));
        const DagMat mat = mats[b];
        double* const Km = mat.K;
        double* const Rv = mat.R;
        double* const Wm = mat.Wt;
        const int ld = mat.ld, N = mat.N, Npad = mat.Npad, P = mat.P;
        GpDev g;
        load_gp(mat.gp, C, g);
        double dsum = g.a2[0];
        {
#pragma clang fp contract(off)
            for (int c = 1; c < C; ++c) dsum = dsum + g.a2[c];
        }
        const DagAug aug{P, 0, 0, nullptr, nullptr, nullptr, nullptr, 0};
        for (int q = 0; q < P; ++q) {
            const int k0 = q * NB;
            for (int j = q; j < P; ++j) {
                const int j0 = j * NB;
                Tile t;
                t.zero();
                if (q > 0) tile_gemm_tn<true>(t, Km + k0, (size_t)ld, Km + j0, (size_t)ld, k0, j == q, 0x7fffffff, wave_s);
                dag_store_updated<C, false, false, false, false, false>(t, Km + (size_t)k0 * ld + j0, (size_t)ld, k0, j0, mat.lw, g, dsum,
                                                                         mat.sigma, N, 1.0, Npad, aug, nullptr, true);
                solo_sync();      // the tile is re-read below in another layout by other waves
                if (j == q) {
                    __builtin_amdgcn_s_setprio(3);
                    potrf_blocked(Km, ld, k0, Wm, Rv, mat.acc);
                    __builtin_amdgcn_s_setprio(0);
                } else {
                    dag_trsm<true, double*, SmemKernel, false>(t, Km, ld, k0, j0, Wm, Rv, Npad, vec1, vec2);
                }
                solo_sync();
            }
        }
    }
}

}  // namespace psoap
