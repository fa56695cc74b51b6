// predict_kernels.hpp -- GP conditional mean / covariance (the predict_* family).
//
// Reference (psoap/covariance.py): predict_f :25-54, predict_f_g :81-148,
// predict_f_g_sum :151-187, predict_f_g_h :190-251, predict_f_g_h_sum :253-297.
//
// All of them are   mu = m0 + Cx B^-1 (fl - off),   Sigma = A - Cx B^-1 Cx^T
// with B the noisy data covariance.  On the device B = U^T U is factored by the same
// left-looking kernels as the likelihood, with Cx^T appended to B as extra column tiles
// ([B | Cx^T], Npad x (Npad + Rpad)): the panel update and the strip solve then turn those
// columns into W = U^-T Cx^T in place, the right-hand side r = fl - off becomes z = U^-T r,
// and
//     mu = m0 + W^T z            (k_gemv_t, deterministic two-stage reduction)
//     Sigma = A - W^T W          (k_syrk_sub, fp64 MFMA tiles, K = Npad)
// so no explicit inverse or second triangular solve is needed.
#pragma once
#include <atomic>
#include <thread>
#include <chrono>
#include <string>
#include <vector>

#include "chol_kernels.hpp"
#include "dag_kernel.hpp"
#include "fill_kernels.hpp"

namespace psoap {

struct GpHost {
    double v[6];
};

// out[i][col0 + j] = sum_{c<C} a_c^2 exp(p_c (xcol_c[j] - xrow_c[i])^2),  i < nrows, j < ncols
// (fill_V12_f semantics, pyx:61-96, summed over components left to right as the reference
// sums V12_f + V12_g (+ V12_h), covariance.py:171,279).  With sym_diag the i == j entries
// follow the fill_V11_f diagonal rule plus `nugget` (pyx:56-57; covariance.py:165).
// Scalar stores: column offsets in the predict layout are not 16-byte aligned.
template <int C>
__global__ __launch_bounds__(256) void k_fill_region(double* __restrict__ out, size_t ld, int col0, int nrows,
                                                     int ncols, const double* __restrict__ xrow, size_t xrow_stride,
                                                     const double* __restrict__ xcol, size_t xcol_stride, GpHost gph,
                                                     int sym_diag, double nugget)
{
    __shared__ double xr[C][NB];
    const int tid = threadIdx.x;
    const int i0 = blockIdx.y * NB, j0 = blockIdx.x * NB;
    GpDev g;
    load_gp(gph.v, C, g);
    double dsum = g.a2[0];
    {
#pragma clang fp contract(off)
        for (int c = 1; c < C; ++c) dsum = dsum + g.a2[c];
    }
    if (tid < NB) {
#pragma unroll
        for (int c = 0; c < C; ++c) xr[c][tid] = (i0 + tid < nrows) ? xrow[c * xrow_stride + i0 + tid] : 0.0;
    }
    const int col = tid & 127, half = tid >> 7;
    const int j = j0 + col;
    double xj[C];
#pragma unroll
    for (int c = 0; c < C; ++c) xj[c] = (j < ncols) ? xcol[c * xcol_stride + j] : 0.0;
    __syncthreads();
    if (j >= ncols) return;
    for (int it = 0; it < NB / 2; ++it) {
        const int r = half + 2 * it;
        const int i = i0 + r;
        if (i >= nrows) break;
        double xi[C];
#pragma unroll
        for (int c = 0; c < C; ++c) xi[c] = xr[c][r];
        double v = kern_elem<C>(xi, xj, g);
        if (sym_diag && i == j) {
#pragma clang fp contract(off)
            v = dsum + nugget;
        }
        out[(size_t)i * ld + col0 + j] = v;
    }
}

// S tile (ti, tj) -= sum_k W[k][128 ti + .] W[k][128 tj + .]   (all tiles; calibration's pass 1 / 2)
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_syrk_sub(const double* __restrict__ W, size_t ldw, int K,
                                                             double* __restrict__ S, size_t lds)
{
    const int ti = blockIdx.y, tj = blockIdx.x;
    Tile t;
    t.zero();
    tile_gemm_tn(t, W + (size_t)NB * ti, ldw, W + (size_t)NB * tj, ldw, K);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        double* p0 = S + (size_t)(NB * ti + tile_row(wr, m, lane, 0)) * lds + NB * tj + tile_col(wc, 0, lane);
        double v[4][4];
        tile_load16(p0, (size_t)4 * lds, v);
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) p0[(size_t)4 * r * lds + 16 * n] = v[n][r] - t.acc[m][n][r];
    }
}

// Symmetric form for predict's Sigma = A - W^T W (returned in full, covariance.py:143,251): one workgroup
// per UPPER tile (ti <= tj) computes S(ti, tj) -= W_ti^T W_tj and also writes its transpose into
// S(tj, ti), so the product costs N R^2 / 2 MFMA flops instead of N R^2 and Sigma is bitwise symmetric.
// The prior A must be present in the upper tiles (the lower ones are overwritten).
// t0: index of the launch's first upper tile (row-major over the upper triangle): the Sigma download streams by tile
// rows, so the product is launched in groups of tile rows, top to bottom -- rows < r of Sigma are final once the
// tile rows < r are done (their lower parts were mirrored in by the rows above).
__global__ __launch_bounds__(GEMM_THREADS, 2) void k_syrk_sub_sym(const double* __restrict__ W, size_t ldw, int K,
                                                                 double* __restrict__ S, size_t lds, int St, int t0)
{
    int ti, tj;
    decode_upper(blockIdx.x + t0, St, ti, tj);
    Tile t;
    t.zero();
    tile_gemm_tn(t, W + (size_t)NB * ti, ldw, W + (size_t)NB * tj, ldw, K);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        double* p0 = S + (size_t)(NB * ti + tile_row(wr, m, lane, 0)) * lds + NB * tj + tile_col(wc, 0, lane);
        double c[4][4];
        tile_load16(p0, (size_t)4 * lds, c);
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = tile_row(wr, m, lane, r), col = tile_col(wc, n, lane);
                const double v = c[n][r] - t.acc[m][n][r];
                p0[(size_t)4 * r * lds + 16 * n] = v;
                if (ti != tj) S[(size_t)(NB * tj + col) * lds + NB * ti + row] = v;
            }
    }
}

// partial[s][q] = sum_{k in slab s} W[k][q] z[k];  slabs of 256 rows, 128 columns per block
__global__ __launch_bounds__(256) void k_gemv_t_partial(const double* __restrict__ W, size_t ldw, int K, int ncols,
                                                        const double* __restrict__ z, double* __restrict__ partial)
{
    __shared__ double red[128];
    const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
    const int q = blockIdx.x * 128 + col;
    const int k0 = blockIdx.y * 256;
    double s = 0.0;
    if (q < ncols) {
        for (int k = k0 + half; k < k0 + 256 && k < K; k += 2) s = fma(W[(size_t)k * ldw + q], z[k], s);
    }
    if (half == 1) red[col] = s;
    __syncthreads();
    if (half == 0 && q < ncols) partial[(size_t)blockIdx.y * ncols + q] = s + red[col];
}

__global__ void k_gemv_finish(const double* __restrict__ partial, int nslab, int ncols,
                              const double* __restrict__ m0, double* __restrict__ mu)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ncols) return;
    double s = 0.0;
    for (int k = 0; k < nslab; ++k) s += partial[(size_t)k * ncols + q];
    mu[q] = m0[q] + s;
}

// diag(Sigma) alone, for callers that only take sqrt(diag) (scripts/psoap_retrieve_ST3.py:111):
//   var[q] = prior[q] - sum_k W[k][q]^2,  partial sums over slabs of 256 rows in a fixed order
__global__ __launch_bounds__(256) void k_colnorm_partial(const double* __restrict__ W, size_t ldw, int K, int ncols,
                                                         double* __restrict__ partial)
{
    __shared__ double red[128];
    const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
    const int q = blockIdx.x * 128 + col;
    const int k0 = blockIdx.y * 256;
    double s = 0.0;
    if (q < ncols) {
        for (int k = k0 + half; k < k0 + 256 && k < K; k += 2) {
            const double w = W[(size_t)k * ldw + q];
            s = fma(w, w, s);
        }
    }
    if (half == 1) red[col] = s;
    __syncthreads();
    if (half == 0 && q < ncols) partial[(size_t)blockIdx.y * ncols + q] = s + red[col];
}

__global__ void k_var_finish(const double* __restrict__ partial, int nslab, int ncols, const double* __restrict__ prior,
                             double* __restrict__ var)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ncols) return;
    double s = 0.0;
    for (int k = 0; k < nslab; ++k) s += partial[(size_t)k * ncols + q];
    var[q] = prior[q] - s;
}

__global__ void k_zero(double* __restrict__ p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = 0.0;
}

inline hipError_t predict_configure_kernels()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_sub),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_syrk_sub_sym),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)GEMM_LDS_BYTES);
#define PSOAP_SET_LDS(...)                                                                            \
    if (e == hipSuccess)                                                                              \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)GEMM_LDS_BYTES)
    PSOAP_SET_LDS(k_chol_dag<1, true, false>);
    PSOAP_SET_LDS(k_chol_dag<2, true, false>);
    PSOAP_SET_LDS(k_chol_dag<3, true, false>);
    PSOAP_SET_LDS(k_chol_dag<1, true, true>);
    PSOAP_SET_LDS(k_chol_dag<2, true, true>);
    PSOAP_SET_LDS(k_chol_dag<3, true, true>);
    PSOAP_SET_LDS(k_chol_dag<1, true, true, false, 1>);
    PSOAP_SET_LDS(k_chol_dag<2, true, true, false, 1>);
    PSOAP_SET_LDS(k_chol_dag<3, true, true, false, 1>);
#undef PSOAP_SET_LDS
    return e;
}

// Grow-only device (or pinned host) buffer of a predict workspace: repeated calls on one chunk handle
// allocate nothing (the retrieve loop calls predict once per chunk, psoap_retrieve_ST3.py:148).
template <class T, bool HOST = false>
struct Grow {
    T* p = nullptr;
    size_t cap = 0;
    Grow() = default;
    Grow(const Grow&) = delete;
    Grow& operator=(const Grow&) = delete;
    ~Grow() { release(); }
    void release()
    {
        if (p) (void)(HOST ? hipHostFree(p) : hipFree(p));
        p = nullptr;
        cap = 0;
    }
    hipError_t need(size_t count)
    {
        if (count <= cap && p) return hipSuccess;
        release();
        const hipError_t e = HOST ? hipHostMalloc(reinterpret_cast<void**>(&p), sizeof(T) * count)
                                  : hipMalloc(reinterpret_cast<void**>(&p), sizeof(T) * count);
        if (e == hipSuccess) cap = count;
        else p = nullptr;
        return e;
    }
    operator T*() const { return p; }
};

// timings of the last predict call (ms): everything on the device up to mu / Sigma complete, the Sigma
// download, and the whole call (host wall clock)
struct PredictTimes {
    double device_ms = 0.0, factor_ms = 0.0, sigma_ms = 0.0, download_ms = 0.0, total_ms = 0.0;
    double flops = 0.0;
};

struct PredictWs {
    Grow<double> K, W, R, Lwl, Pred, Fl, Sig, Gp, S, Mu, M0, Part, Colx, Ws;
    Grow<MatAcc> Acc;
    Grow<unsigned char> Dag;
    Grow<DagTask> Tasks;
    Grow<unsigned int> Order, Dep;   // ready-only hand-out (DagPool)
    Grow<unsigned long long> Tlog;   // per-task stamps of the launch (PSOAP_PREDICT_TLOG=<file>: tools/predict_timeline.py)
    Grow<DagMat> Mat;
    Grow<double, true> hSmall;   // pinned: colx / m0 staging, mu
    // task list cache
    int plan_P = -1, plan_Mt = -1, plan_workers = -1, plan_scheme = -2, plan_Ms = -1;
    DagPlan plan;
    int workers = 0, n_cus = 0;   // persistent workgroups the device admits; compute units
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = {};
    Grow<double> Var, Prior, Rowx, Diag;
    PredictTimes times;
    ~PredictWs()
    {
        if (stream) (void)hipStreamDestroy(stream);
        for (auto e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};

#define PR_TRY(expr)                                                 \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) {                                      \
            err = std::string(#expr) + ": " + hipGetErrorString(_e); \
            return 1;                                                \
        }                                                            \
    } while (0)

template <int C>
static void launch_region(hipStream_t st, double* out, size_t ld, int col0, int nrows, int ncols, const double* xrow,
                          size_t xrs, const double* xcol, size_t xcs, const GpHost& g, int sym, double nug)
{
    dim3 grid((ncols + NB - 1) / NB, (nrows + NB - 1) / NB);
    hipLaunchKernelGGL(k_fill_region<C>, grid, dim3(256), 0, st, out, ld, col0, nrows, ncols, xrow, xrs, xcol, xcs, g,
                       sym, nug);
}

static void launch_region_c(hipStream_t st, int C, double* out, size_t ld, int col0, int nrows, int ncols,
                            const double* xrow, size_t xrs, const double* xcol, size_t xcs, const GpHost& g, int sym,
                            double nug)
{
    if (C == 1) launch_region<1>(st, out, ld, col0, nrows, ncols, xrow, xrs, xcol, xcs, g, sym, nug);
    else if (C == 2) launch_region<2>(st, out, ld, col0, nrows, ncols, xrow, xrs, xcol, xcs, g, sym, nug);
    else launch_region<3>(st, out, ld, col0, nrows, ncols, xrow, xrs, xcol, xcs, g, sym, nug);
}

// The left-looking factorisation of an Npad x (Npad + Mt*128) augmented matrix [B | extra columns]:
// afterwards the extra columns hold U^-T (extra) and dR holds z = U^-T r.
inline void factor_augmented(hipStream_t st, double* dK, size_t ld, int P, int Mt, double* dW, double* dR, int Npad,
                             MatAcc* dAcc)
{
    for (int p = 0; p < P; ++p) {
        const int k0 = p * NB;
        const int ntile = P - p + Mt;
        if (p > 0)
            hipLaunchKernelGGL(k_panel_update, dim3(ntile, 1), dim3(GEMM_THREADS), GEMM_LDS_BYTES, st, dK, (size_t)0,
                               (int)ld, k0);
        hipLaunchKernelGGL(k_potrf_diag, dim3(1), dim3(512), 0, st, dK, (size_t)0, (int)ld, k0, dW, dR, Npad, dAcc);
        if (ntile > 1)
            hipLaunchKernelGGL(k_trsm_strip, dim3(ntile - 1, 1), dim3(GEMM_THREADS), GEMM_LDS_BYTES, st, dK, (size_t)0,
                               (int)ld, k0, dW, dR, Npad);
    }
}

// mode 0: components (predict_f_g / predict_f_g_h); 1: sum (predict_f_g_sum / _h_sum); 2: predict_f
// dFl_res / dSig_res: the data and noise vectors already resident on the device (a chunk handle's), or
// nullptr -> fl / sigma are uploaded into the workspace.
// Host side of the Sigma download.  The caller's array is pageable and, coming from numpy, freshly mapped; a plain D2H
// into it is staged by the runtime at 6-25 GB/s with a large spread (3-12 ms for the 75 MB of the retrieve shape).  The
// array is therefore page-locked IN PLACE (hipHostRegister: 2.6-3.1 ms, first-touch faults included) by a helper
// thread while the GPU is still factoring, the copy then runs at the link rate (1.4 ms) straight into it, and the
// registration is dropped afterwards (10 us).  (Round 3 first built a pinned bounce buffer emptied by 16 host threads:
// 1.7 ms on one box, 3.7 ms on another -- NUMA placement of buffer and threads.)  Small outputs skip all that.
// Prior variance of prediction column q: the squared amplitudes of the components that contribute to it (fill_V11_* put
// amp^2 on the diagonal), + the 1e-8 nugget of predict_f_g_sum.  ONE routine for the diagonal of the fused Sigma and for
// psoap_*_predict_var, with FMA contraction off: both round alike (and like the reference's a*a + b*b).
inline double predict_prior_variance(int mode, int c, int M, int q, const double* gp)
{
#pragma clang fp contract(off)
    if (mode == 0) return gp[2 * (q / M)] * gp[2 * (q / M)];
    double a2 = gp[0] * gp[0];
    for (int k = 1; k < c; ++k) a2 = a2 + gp[2 * k] * gp[2 * k];
    if (mode == 1 && c == 2) a2 = a2 + 1e-8;
    return a2;
}

struct SigmaPin {
    double* ptr = nullptr;
    size_t bytes = 0;
    int device = 0;
    std::thread th;
    std::atomic<int> ok{0};
    bool started = false;
    hipStream_t copy_stream = nullptr;      // an asynchronous copy INTO the array was queued on this stream
    bool copy_queued = false;
    static constexpr size_t MIN_BYTES = (size_t)4 << 20;

    void start(double* out, size_t n_bytes, int dev)
    {
        if (n_bytes < MIN_BYTES) return;
        ptr = out;
        bytes = n_bytes;
        device = dev;
        started = true;
        th = std::thread([this] {
            if (hipSetDevice(device) == hipSuccess && hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess)
                ok.store(1, std::memory_order_release);
        });
    }
    // -> true when the array is page-locked (the copy may then be asynchronous on any stream)
    bool wait()
    {
        if (started && th.joinable()) th.join();
        return ok.load(std::memory_order_acquire) != 0;
    }
    void queued_on(hipStream_t s)
    {
        copy_stream = s;
        copy_queued = true;
    }
    void release()
    {
        if (started && th.joinable()) th.join();
        // (an early return may come here with the DMA into the array still in flight: the registration outlives it)
        if (copy_queued) (void)hipStreamSynchronize(copy_stream);
        copy_queued = false;
        if (ok.load()) (void)hipHostUnregister(ptr);
        ok.store(0);
        started = false;
    }
    ~SigmaPin() { release(); }
};

// var_out (optional): diag(Sigma) alone -- R doubles instead of R^2, and no N R^2 product
inline int predict_run(PredictWs& ws, int mode, int c, int N, int M, const double* lwl, const double* fl,
                       const double* sigma, const double* dFl_res, const double* dSig_res, const double* lwl_pred,
                       const double* mu_c, const double* gp, double* mu_out, double* Sigma_out, int* status,
                       std::string& err, double* var_out = nullptr, bool force_staged = false)
{
    const auto t_begin = std::chrono::steady_clock::now();
    SigmaPin pin;         // declared first: its destructor joins the helper thread and drops the registration on every return path
    const int Npad = round_up(N, NB), P = Npad / NB;
    const bool transposed_mean = (mode == 1 && c == 3);  // covariance.py:294 uses V12.T in the mean
    if (transposed_mean && M != N) {
        err = "predict_f_g_h_sum is only defined for len(lwl_predict) == len(lwl) (covariance.py:294)";
        return 2;
    }
    const int Rq = (mode == 0) ? c * M : M;             // columns of W that feed Sigma
    const int Rq_pad = round_up(Rq, NB);
    const int Rx = transposed_mean ? N : 0;             // extra block holding U^-T V12 for the :294 mean
    const int Rx_pad = round_up(Rx, NB);
    const int Rtot_pad = Rq_pad + Rx_pad;
    const size_t ld = (size_t)Npad + Rtot_pad;
    const int Mt = Rtot_pad / NB;
    const double offset = (mode == 0 || (mode == 1 && c == 2)) ? 1.0 : mu_c[0];  // :140,:184,:248 vs :52,:294
    const int nslab = (Npad + 255) / 256;
    // The factorisation of [B | Cx^T] runs as ONE launch of the persistent dependency-graph kernel with
    // the appended columns as extra column tiles (k_chol_dag<C, true>: the cross-covariances are
    // evaluated on the fly like B itself).  The transposed-mean variant (covariance.py:294) needs a
    // block whose ROW abscissae are the prediction grid and keeps the staged three-kernel loop.
    // (force_staged: several processes share the device and the persistent launch is not to be used, or was disturbed --
    // psoap_gp.hip: predict_settled)
    const bool use_dag = !force_staged && !transposed_mean && (P + Mt) <= 255;
    // Sigma = A - W^T W is the Schur complement of the appended columns: with the persistent kernel its tiles are tasks
    // of the same launch (DAG_SCHUR), left-looking updates that need no critical path and fill the workgroups the
    // factorisation's row-to-row chain leaves idle.  PSOAP_PREDICT_FUSED=0: the separate product (k_syrk_sub_sym).
    const char* env_fused = getenv("PSOAP_PREDICT_FUSED");
    const bool fused_sigma = use_dag && Sigma_out && !(env_fused && atoi(env_fused) == 0);
    const int Ms = fused_sigma ? Rq_pad / NB : 0;
    GpHost gall;
    for (int k = 0; k < 6; ++k) gall.v[k] = (k < 2 * c) ? gp[k] : 0.0;

    if (!ws.stream) {
        PR_TRY(hipStreamCreateWithFlags(&ws.stream, hipStreamNonBlocking));
        for (auto& e : ws.ev) PR_TRY(hipEventCreate(&e));
    }
    if (Sigma_out) {
        int dev_now = 0;
        PR_TRY(hipGetDevice(&dev_now));
        pin.start(Sigma_out, sizeof(double) * (size_t)Rq * Rq, dev_now);
    }

    hipStream_t st = ws.stream;
    PR_TRY(ws.K.need((size_t)Npad * ld));
    PR_TRY(ws.W.need(WT_STRIDE));
    PR_TRY(ws.R.need(Npad));
    PR_TRY(ws.Acc.need(ACC_ROWS + 1));      // the records of the block rows + their total (k_acc_total)
    PR_TRY(ws.Lwl.need((size_t)c * N));
    PR_TRY(ws.Pred.need((size_t)c * M));
    PR_TRY(ws.Gp.need(6));
    PR_TRY(ws.Mu.need(Rq_pad));
    PR_TRY(ws.M0.need(Rq_pad));
    PR_TRY(ws.Part.need((size_t)nslab * Rtot_pad));
    const size_t n_small = 2 * (size_t)c * Rq_pad + 4 * (size_t)Rq_pad + 64;
    PR_TRY(ws.hSmall.need(n_small));
    double* h_colx = ws.hSmall;                      // (c, Rq_pad)
    double* h_m0 = h_colx + (size_t)c * Rq_pad;      // (Rq_pad)
    double* h_mu = h_m0 + Rq_pad;                    // (Rq_pad)
    MatAcc* h_acc = reinterpret_cast<MatAcc*>(h_mu + Rq_pad);
    double* h_rowx = h_mu + 2 * (size_t)Rq_pad + 8;   // (c, Rq_pad), behind h_acc and the variance staging
    double* h_diag = h_rowx + (size_t)c * Rq_pad;     // (Rq_pad)
    double *dK = ws.K, *dW = ws.W, *dR = ws.R, *dLwl = ws.Lwl, *dPred = ws.Pred, *dGp = ws.Gp, *dMu = ws.Mu,
           *dM0 = ws.M0, *dPart = ws.Part;
    MatAcc* dAcc = ws.Acc;
    const double* dFl = dFl_res;
    const double* dSig = dSig_res;
    const auto t_ev0 = std::chrono::steady_clock::now();
    PR_TRY(hipEventRecord(ws.ev[0], st));
    PR_TRY(hipMemcpyAsync(dLwl, lwl, sizeof(double) * (size_t)c * N, hipMemcpyHostToDevice, st));
    PR_TRY(hipMemcpyAsync(dPred, lwl_pred, sizeof(double) * (size_t)c * M, hipMemcpyHostToDevice, st));
    PR_TRY(hipMemcpyAsync(dGp, gp, sizeof(double) * 2 * c, hipMemcpyHostToDevice, st));
    if (!dFl) {
        PR_TRY(ws.Fl.need(N));
        PR_TRY(ws.Sig.need(N));
        PR_TRY(hipMemcpyAsync(ws.Fl, fl, sizeof(double) * N, hipMemcpyHostToDevice, st));
        PR_TRY(hipMemcpyAsync(ws.Sig, sigma, sizeof(double) * N, hipMemcpyHostToDevice, st));
        dFl = ws.Fl;
        dSig = ws.Sig;
    }
    // prior means of the prediction (added to W^T z)
    for (int q = 0; q < Rq; ++q) h_m0[q] = (mode == 0) ? mu_c[q / M] : mu_c[0];
    PR_TRY(hipMemcpyAsync(dM0, h_m0, sizeof(double) * Rq, hipMemcpyHostToDevice, st));

    if (use_dag) {
        // prediction abscissae per component for the appended columns; a component that does not
        // contribute to a column block (mode 0: C = vstack(V12_f, V12_g, ..), :136,:246) sits at 1e30
        for (int cc = 0; cc < c; ++cc) {
            double* row = h_colx + (size_t)cc * Rq_pad;
            for (int q = 0; q < Rq; ++q) {
                const int blk = (mode == 0) ? q / M : cc;
                row[q] = (mode == 0 && blk != cc) ? 1e30 : lwl_pred[(size_t)cc * M + (q % M)];
            }
            for (int q = Rq; q < Rq_pad; ++q) row[q] = 0.0;
        }
        const char* env_scheme = getenv("PSOAP_DAG_SCHEME");
        const int scheme = env_scheme ? atoi(env_scheme) : -1;
        const int workers = dag_pick_workers(dag_batch_flops(std::vector<int>(1, P), Mt), P, ws.n_cus > 0 ? ws.n_cus : ws.workers,
                                             ws.workers);
        if (ws.plan_P != P || ws.plan_Mt != Mt || ws.plan_workers != workers || ws.plan_scheme != scheme || ws.plan_Ms != Ms) {
            ws.plan = dag_build_tasks(1, P, workers, scheme, Mt, Ms);
            PR_TRY(hipStreamSynchronize(st));
            PR_TRY(ws.Tasks.need(ws.plan.tasks.size()));
            PR_TRY(hipMemcpy(ws.Tasks, ws.plan.tasks.data(), sizeof(DagTask) * ws.plan.tasks.size(),
                             hipMemcpyHostToDevice));
            PR_TRY(ws.Ws.need((size_t)NB * NB * ((size_t)ws.plan.n_slots + 1)));
            if (!ws.plan.order.empty()) {
                PR_TRY(ws.Order.need(ws.plan.order.size()));
                PR_TRY(ws.Dep.need(ws.plan.dep.size()));
                PR_TRY(hipMemcpy(ws.Order, ws.plan.order.data(), sizeof(unsigned int) * ws.plan.order.size(), hipMemcpyHostToDevice));
                PR_TRY(hipMemcpy(ws.Dep, ws.plan.dep.data(), sizeof(unsigned int) * ws.plan.dep.size(), hipMemcpyHostToDevice));
            }
            ws.plan_P = P;
            ws.plan_Mt = Mt;
            ws.plan_workers = workers;
            ws.plan_scheme = scheme;
            ws.plan_Ms = Ms;
        }
        const DagPlan& plan = ws.plan;
        const size_t arrive_off = sizeof(DagCtl) + sizeof(MatFlags);
        const size_t taken_off = arrive_off + sizeof(int) * ((size_t)plan.n_ctrs + 4);      // the ready-only hand-out's bitmap
        const size_t dag_bytes = taken_off + sizeof(unsigned int) * ((plan.tasks.size() + 31) / 32 + 1);
        PR_TRY(ws.Colx.need((size_t)c * Rq_pad));
        PR_TRY(ws.Dag.need(dag_bytes));
        PR_TRY(ws.Mat.need(1));
        PR_TRY(hipMemcpyAsync(ws.Colx, h_colx, sizeof(double) * (size_t)c * Rq_pad, hipMemcpyHostToDevice, st));
        PR_TRY(hipMemsetAsync(ws.Dag, 0, dag_bytes, st));
        PR_TRY(hipMemsetAsync(dW, 0, sizeof(double) * 2 * NB * NB, st));  // the strictly upper part of W stays zero
        PR_TRY(hipMemsetAsync(dW + WT_THIRD, 0, sizeof(double) * NB * NB, st));   // (scheme 0's third tile)
        hipLaunchKernelGGL(k_init_rhs, dim3((Npad + 255) / 256, 1), dim3(256), 0, st, dR, Npad, N, dFl, offset, dAcc);
        const int grid = (int)(plan.tasks.size() < (size_t)workers ? plan.tasks.size() : (size_t)workers);
        DagAug aug{P + Mt, Rq, Rq_pad, ws.Colx, nullptr, nullptr, nullptr, 0};
        if (fused_sigma) {
            // row abscissae of the Sigma tiles: the column ones with the opposite sentinel; prior variances for the diagonal
            for (int cc = 0; cc < c; ++cc)
                for (int q = 0; q < Rq_pad; ++q) {
                    const double x = h_colx[(size_t)cc * Rq_pad + q];
                    h_rowx[(size_t)cc * Rq_pad + q] = (x == 1e30) ? -1e30 : x;
                }
            for (int q = 0; q < Rq_pad; ++q) h_diag[q] = q < Rq ? predict_prior_variance(mode, c, M, q, gp) : 0.0;
            PR_TRY(ws.Rowx.need((size_t)c * Rq_pad));
            PR_TRY(ws.Diag.need(Rq_pad));
            PR_TRY(ws.S.need((size_t)Rq_pad * Rq_pad));
            PR_TRY(hipMemcpyAsync(ws.Rowx, h_rowx, sizeof(double) * (size_t)c * Rq_pad, hipMemcpyHostToDevice, st));
            PR_TRY(hipMemcpyAsync(ws.Diag, h_diag, sizeof(double) * Rq_pad, hipMemcpyHostToDevice, st));
            aug.rowx = ws.Rowx;
            aug.diag = ws.Diag;
            aug.S = ws.S;
            aug.lds = (size_t)Rq_pad;
        }
        MatFlags* fl_ = reinterpret_cast<MatFlags*>(ws.Dag.p + sizeof(DagCtl));
        DagCtl* ctl_ = reinterpret_cast<DagCtl*>(ws.Dag.p);
        DagMat hm{};
        hm.K = dK; hm.R = dR; hm.Wt = dW; hm.lw = dLwl; hm.gp = dGp; hm.sigma = dSig; hm.acc = dAcc;
        hm.N = N; hm.Npad = Npad; hm.P = P; hm.ld = (int)ld;
        // a kernel argument would do, but the records of the likelihood path live in memory: same code path
        PR_TRY(hipMemcpyAsync(ws.Mat, &hm, sizeof(DagMat), hipMemcpyHostToDevice, st));
        PR_TRY(hipStreamSynchronize(st));   // hm is a stack object; the staging copies are tiny
        PR_TRY(hipEventRecord(ws.ev[1], st));
#define PSOAP_LAUNCH_AUG(CC, LAT, WPE)                                                                            \
    hipLaunchKernelGGL((k_chol_dag<CC, true, LAT, false, WPE>), dim3(grid), dim3(GEMM_THREADS), GEMM_LDS_BYTES, st,  \
                       ws.Mat.p, ws.Tasks.p, plan.queues, fl_, reinterpret_cast<int*>(ws.Dag.p + arrive_off),        \
                       ws.Ws.p, ctl_, tlog_, aug, StreamArgs{}, pool_)
        // debug: per-task stamps of this launch, written with the task list to the file PSOAP_PREDICT_TLOG names
        unsigned long long* tlog_ = nullptr;
        if (getenv("PSOAP_PREDICT_TLOG")) {
            PR_TRY(ws.Tlog.need(plan.tasks.size() * 8));
            PR_TRY(hipMemsetAsync(ws.Tlog, 0, sizeof(unsigned long long) * plan.tasks.size() * 8, st));
            tlog_ = ws.Tlog.p;
        }
        DagPool pool_{};
        if (!plan.order.empty()) {
            pool_.order = ws.Order.p;
            pool_.dep = ws.Dep.p;
            pool_.taken = reinterpret_cast<unsigned int*>(ws.Dag.p + taken_off);
            memcpy(pool_.n_main, plan.n_main, sizeof pool_.n_main);
        }
        const bool lat = plan.scheme >= 1;
        // (at most one workgroup per compute unit: the kernels compiled for one wave per SIMD, as in psoap_gp.hip: eval_dag)
        const bool wide = lat && ws.n_cus > 0 && grid <= ws.n_cus && !(getenv("PSOAP_DAG_WIDE") && getenv("PSOAP_DAG_WIDE")[0] == '0');
        if (c == 1) { if (wide) PSOAP_LAUNCH_AUG(1, true, 1); else if (lat) PSOAP_LAUNCH_AUG(1, true, 2); else PSOAP_LAUNCH_AUG(1, false, 2); }
        else if (c == 2) { if (wide) PSOAP_LAUNCH_AUG(2, true, 1); else if (lat) PSOAP_LAUNCH_AUG(2, true, 2); else PSOAP_LAUNCH_AUG(2, false, 2); }
        else { if (wide) PSOAP_LAUNCH_AUG(3, true, 1); else if (lat) PSOAP_LAUNCH_AUG(3, true, 2); else PSOAP_LAUNCH_AUG(3, false, 2); }
#undef PSOAP_LAUNCH_AUG
        PR_TRY(hipGetLastError());
    } else {
        PR_TRY(hipEventRecord(ws.ev[1], st));
        // [B | Cx^T]: zero the appended columns (padding rows/cols must be exact zeros), fill B's upper tiles
        hipLaunchKernelGGL(k_zero, dim3(2048), dim3(256), 0, st, dK, (size_t)Npad * ld);
        {
            dim3 grid(P * (P + 1) / 2, 1);
            if (c == 1) hipLaunchKernelGGL(k_fill_sym<1>, grid, dim3(256), 0, st, dK, (size_t)0, (int)ld, N, P, dLwl, dGp, dSig, 1);
            else if (c == 2) hipLaunchKernelGGL(k_fill_sym<2>, grid, dim3(256), 0, st, dK, (size_t)0, (int)ld, N, P, dLwl, dGp, dSig, 1);
            else hipLaunchKernelGGL(k_fill_sym<3>, grid, dim3(256), 0, st, dK, (size_t)0, (int)ld, N, P, dLwl, dGp, dSig, 1);
        }
        PR_TRY(hipGetLastError());
        if (mode == 0) {
            // C = vstack(V12_f, V12_g, ..) (:136,:246): column block k of Cx^T is component k alone
            for (int k = 0; k < c; ++k) {
                GpHost g1;
                g1.v[0] = gp[2 * k];
                g1.v[1] = gp[2 * k + 1];
                launch_region<1>(st, dK, ld, Npad + k * M, N, M, dLwl + (size_t)k * N, 0, dPred + (size_t)k * M, 0, g1, 0, 0.0);
            }
        } else {
            // V12 = V12_f + V12_g (+ V12_h) (:171,:279); predict_f's single V12 (:39-40) is the C = 1 case
            launch_region_c(st, c, dK, ld, Npad, N, M, dLwl, (size_t)N, dPred, (size_t)M, gall, 0, 0.0);
            if (transposed_mean)  // rows indexed by the prediction grid, columns by the data grid (M == N)
                launch_region_c(st, c, dK, ld, Npad + Rq_pad, N, N, dPred, (size_t)M, dLwl, (size_t)N, gall, 0, 0.0);
        }
        PR_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_init_rhs, dim3((Npad + 255) / 256, 1), dim3(256), 0, st, dR, Npad, N, dFl, offset, dAcc);
        PR_TRY(hipGetLastError());
        factor_augmented(st, dK, ld, P, Mt, dW, dR, Npad, dAcc);
        PR_TRY(hipGetLastError());
    }
    PR_TRY(hipEventRecord(ws.ev[2], st));

    // mean: m0 + W^T z  (W is the block that matches the reference's orientation for this mode)
    {
        const double* Wmean = dK + Npad + (transposed_mean ? Rq_pad : 0);
        hipLaunchKernelGGL(k_gemv_t_partial, dim3((Rq + 127) / 128, nslab), dim3(256), 0, st, Wmean, ld, Npad, Rq, dR,
                           dPart);
        hipLaunchKernelGGL(k_gemv_finish, dim3((Rq + 255) / 256), dim3(256), 0, st, dPart, nslab, Rq, dM0, dMu);
        PR_TRY(hipGetLastError());
    }
    PR_TRY(hipMemcpyAsync(h_mu, dMu, sizeof(double) * Rq, hipMemcpyDeviceToHost, st));
    hipLaunchKernelGGL(k_acc_total, dim3(1), dim3(1), 0, st, dAcc, P, dAcc + ACC_ROWS);
    PR_TRY(hipGetLastError());
    PR_TRY(hipMemcpyAsync(h_acc, dAcc + ACC_ROWS, sizeof(MatAcc), hipMemcpyDeviceToHost, st));

    bool sigma_direct = false;
    if (Sigma_out) {
        const int St = Rq_pad / NB;
        PR_TRY(ws.S.need((size_t)Rq_pad * Rq_pad));
        double* dS = ws.S;
        if (!fused_sigma) {
            // prior covariance of the prediction, upper tiles only: k_syrk_sub mirrors the result
            if (mode == 0) {
                // A = blockdiag(V11_f_predict, V11_g_predict, ..)  (:124-125,:234-236)
                hipLaunchKernelGGL(k_zero, dim3(1024), dim3(256), 0, st, dS, (size_t)Rq_pad * Rq_pad);
                for (int k = 0; k < c; ++k) {
                    GpHost g1;
                    g1.v[0] = gp[2 * k];
                    g1.v[1] = gp[2 * k + 1];
                    launch_region<1>(st, dS + (size_t)k * M * Rq_pad, (size_t)Rq_pad, k * M, M, M, dPred + (size_t)k * M, 0,
                                     dPred + (size_t)k * M, 0, g1, 1, 0.0);
                }
            } else {
                // V11 = sum of the component priors; 1e-8 nugget only in the two-component sum (:165 vs :271)
                const double nug = (mode == 1 && c == 2) ? 1e-8 : 0.0;
                if (Rq_pad != Rq) hipLaunchKernelGGL(k_zero, dim3(1024), dim3(256), 0, st, dS, (size_t)Rq_pad * Rq_pad);
                launch_region_c(st, c, dS, (size_t)Rq_pad, 0, M, M, dPred, (size_t)M, dPred, (size_t)M, gall, 1, nug);
            }
            PR_TRY(hipGetLastError());
            // Sigma = A - W^T W: one launch -- every tile is one K = Npad loop and all of them fit the device at once, so they
            // all finish together (launching tile rows one after the other serialises them: measured 2.2 -> 10.5 ms)
            hipLaunchKernelGGL(k_syrk_sub_sym, dim3(St * (St + 1) / 2), dim3(GEMM_THREADS), GEMM_LDS_BYTES, st, dK + Npad, ld,
                               Npad, dS, (size_t)Rq_pad, St, 0);
            PR_TRY(hipGetLastError());
        }
        PR_TRY(hipEventRecord(ws.ev[3], st));           // the device work of the call ends here
        // the download, queued right behind the product on the same stream: into the page-locked caller's array when the
        // registration went through (it did its work under the factorisation), through the runtime's staging otherwise
        const bool locked = pin.wait();
        if (locked) {
            pin.queued_on(st);
            PR_TRY(hipMemcpy2DAsync(Sigma_out, sizeof(double) * Rq, dS, sizeof(double) * Rq_pad, sizeof(double) * Rq, (size_t)Rq,
                                    hipMemcpyDeviceToHost, st));
        }
        sigma_direct = locked;
    }
    if (var_out) {
        // diag(Sigma) = diag(A) - column norms of W: the prior variances are the squared amplitudes of the components
        // that contribute to the column (fill_V11_* put amp^2 on the diagonal), + the 1e-8 nugget of predict_f_g_sum
        PR_TRY(ws.Var.need(Rq_pad));
        PR_TRY(ws.Prior.need(Rq_pad));
        double* h_prior = h_mu + Rq_pad + 8;          // behind h_mu and h_acc in the pinned staging block
        for (int q = 0; q < Rq; ++q) h_prior[q] = predict_prior_variance(mode, c, M, q, gp);
        PR_TRY(hipMemcpyAsync(ws.Prior, h_prior, sizeof(double) * Rq, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_colnorm_partial, dim3((Rq + 127) / 128, nslab), dim3(256), 0, st, dK + Npad, ld, Npad, Rq, dPart);
        hipLaunchKernelGGL(k_var_finish, dim3((Rq + 255) / 256), dim3(256), 0, st, dPart, nslab, Rq, ws.Prior.p, ws.Var.p);
        PR_TRY(hipGetLastError());
        PR_TRY(hipMemcpyAsync(h_prior, ws.Var, sizeof(double) * Rq, hipMemcpyDeviceToHost, st));
    }
    if (!Sigma_out) PR_TRY(hipEventRecord(ws.ev[3], st));
    PR_TRY(hipStreamSynchronize(st));
    if (use_dag) {
        unsigned int dag_err[6] = {0, 0, 0, 0, 0, 0};      // DagCtl::error, pad[0..4]
        PR_TRY(hipMemcpy(dag_err, ws.Dag.p + offsetof(DagCtl, error), sizeof(dag_err), hipMemcpyDeviceToHost));
        if (dag_err[0] != 0) {
            err = "predict: dependency wait timed out inside the persistent kernel";
            return 1;
        }
        if (const char* tpath = getenv("PSOAP_PREDICT_TLOG")) {
            std::vector<unsigned long long> hl(ws.plan.tasks.size() * 8);
            PR_TRY(hipMemcpy(hl.data(), ws.Tlog.p, sizeof(unsigned long long) * hl.size(), hipMemcpyDeviceToHost));
            if (FILE* fh = fopen(tpath, "wb")) {
                const unsigned long long hdr[4] = {(unsigned long long)ws.plan.tasks.size(), (unsigned long long)P,
                                                   (unsigned long long)Mt, (unsigned long long)ws.plan.scheme};
                fwrite(hdr, sizeof hdr, 1, fh);
                fwrite(ws.plan.tasks.data(), sizeof(DagTask), ws.plan.tasks.size(), fh);
                fwrite(hl.data(), sizeof(unsigned long long), hl.size(), fh);
                fclose(fh);
            }
        }
        if (dag_err[4] != 0) {
            // a workgroup moved between compute units under one of the launch's tasks (dag_where): the outputs are not
            // handed out -- the caller runs the call again
            err = "predict: the persistent launch was disturbed by the device's scheduler";
            pin.release();
            return 3;
        }
    }
    if (Sigma_out && !sigma_direct)
        PR_TRY(hipMemcpy2D(Sigma_out, sizeof(double) * Rq, ws.S, sizeof(double) * Rq_pad, sizeof(double) * Rq, Rq,
                           hipMemcpyDeviceToHost));
    pin.release();
    *status = (h_acc->info != 0.0) ? 1 : 0;
    memcpy(mu_out, h_mu, sizeof(double) * Rq);
    if (var_out) memcpy(var_out, h_mu + Rq_pad + 8, sizeof(double) * Rq);
    const auto t_end = std::chrono::steady_clock::now();
    {
        float a = 0.f, b = 0.f, cms = 0.f;
        (void)hipEventElapsedTime(&a, ws.ev[0], ws.ev[3]);
        (void)hipEventElapsedTime(&b, ws.ev[1], ws.ev[2]);
        (void)hipEventElapsedTime(&cms, ws.ev[2], ws.ev[3]);
        ws.times.device_ms = a;
        ws.times.factor_ms = b;
        ws.times.sigma_ms = cms;
        ws.times.total_ms = std::chrono::duration<double, std::milli>(t_end - t_begin).count();
        // what the download adds to the call beyond the device work (the groups travel while later ones are computed)
        const double pre_ms = std::chrono::duration<double, std::milli>(t_ev0 - t_begin).count();
        ws.times.download_ms = ws.times.total_ms - pre_ms - (double)a > 0.0 ? ws.times.total_ms - pre_ms - (double)a : 0.0;
        const double n = Npad, r = Rq_pad;
        ws.times.flops = n * n * n / 3.0 + n * n * r + (Sigma_out ? n * r * r : (var_out ? 2.0 * n * r : 0.0)) + 2.0 * n * r;   // SURVEY 8(d) F_pred
    }
    return 0;
}

}  // namespace psoap
