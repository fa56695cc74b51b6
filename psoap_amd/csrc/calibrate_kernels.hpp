// calibrate_kernels.hpp -- Chebyshev flux re-calibration of one epoch against reference epochs
// (SURVEY.md 8(f), row f-4).
//
// Reference: psoap/covariance.py optimize_calibration :560-624 (explicit A, B, C),
// optimize_calibration_static :628-707 (fills them itself), and the per-epoch loop of
// scripts/psoap_process_calibration_ST3.py:147-183 (three-component fills).  The solve is
//
//     fl' = mu + C B^-1 (fl_fixed - mu)            conditional mean at the epoch's pixels      (:603)
//     C'  = A - C B^-1 C^T                          conditional covariance (A holds sigma_cal^2) (:604)
//     D   = fl_cal[:, None] * T_k(lwl_cal)          Chebyshev design matrix, k = 0..order        (:584-593)
//     X   = (D^T C'^-1 D)^-1 D^T C'^-1 fl'          generalised least squares                    (:607-616)
//     fl_cor = D X                                                                             (:619)
//
// Device plan: two passes of the augmented-column factorisation that predict_* uses.
//   pass 1: [B | C^T] -> W1 = U^-T C^T, z1;  fl' = mu + W1^T z1;  C' = A - W1^T W1  (written straight
//           into pass 2's matrix buffer)
//   pass 2: [C' | D fl'] -> W2 = U2^-T [D fl'];  G = W2^T W2 holds D^T C'^-1 D and D^T C'^-1 fl'
// Only the (order+1)^2 normal equations are solved on the host.
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "predict_kernels.hpp"

namespace psoap {

// rows / columns N..Npad-1 of a padded symmetric matrix: unit diagonal (the rest is already zero)
__global__ void k_pad_identity(double* __restrict__ K, size_t ld, int N, int Npad)
{
    const int i = N + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Npad) K[(size_t)i * ld + i] = 1.0;
}

// dst[j][col0 + i] = src[i][j]  (i < rows, j < cols): C (M x N) -> the C^T column block of [B | C^T]
__global__ void k_transpose_in(double* __restrict__ dst, size_t ldd, int col0, const double* __restrict__ src,
                               size_t lds, int rows, int cols)
{
    __shared__ double tile[32][33];
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8)
        tile[r][tx] = (i0 + r < rows && j0 + tx < cols) ? src[(size_t)(i0 + r) * lds + j0 + tx] : 0.0;
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (j0 + r < cols && i0 + tx < rows) dst[(size_t)(j0 + r) * ldd + col0 + i0 + tx] = tile[tx][r];
}

__global__ void k_add_diag_sq(double* __restrict__ S, size_t ld, int M, const double* __restrict__ sigma)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) {
#pragma clang fp contract(off)
        S[(size_t)i * ld + i] = S[(size_t)i * ld + i] + sigma[i] * sigma[i];
    }
}

// Chebyshev T_k on the domain-mapped abscissa, numpy.polynomial.Chebyshev(coef, domain=[a, b]) semantics:
// x' = off + scl * x with (off, scl) mapping [a, b] onto [-1, 1]
__host__ __device__ inline void cheb_row(double x, double off, double scl, int order, double* T)
{
    const double u = off + scl * x;
    T[0] = 1.0;
    if (order >= 1) T[1] = u;
    for (int k = 2; k <= order; ++k) T[k] = 2.0 * u * T[k - 1] - T[k - 2];
}

constexpr int CAL_MAX_ORDER = 15;

// columns col0 .. col0+order of K2: D[i][k] = fl_cal[i] T_k(lwl_cal[i]);  column col0+order+1: fl'
__global__ void k_calib_aug(double* __restrict__ K2, size_t ld2, int col0, int M, int order,
                            const double* __restrict__ lwl_cal, const double* __restrict__ fl_cal, double off,
                            double scl, const double* __restrict__ flp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    double T[CAL_MAX_ORDER + 1];
    cheb_row(lwl_cal[i], off, scl, order, T);
    double* row = K2 + (size_t)i * ld2 + col0;
    for (int k = 0; k <= order; ++k) row[k] = fl_cal[i] * T[k];
    row[order + 1] = flp[i];
}

struct CalibInputs {
    // common
    int M, N, order;
    double lwl0, lwl1, mu;
    const double* lwl_cal;    // (M) abscissa of the Chebyshev polynomials
    const double* fl_cal;     // (M)
    const double* fl_fixed;   // (N)
    // kernel form (c >= 1): matrices are evaluated on the device
    int c;
    const double* lwls_cal;     // (c, M) rest-frame grids of the epoch
    const double* sigma_cal;    // (M)
    const double* lwls_fixed;   // (c, N)
    const double* sigma_fixed;  // (N)
    const double* gp;           // (2c)
    // explicit form (c == 0): caller-filled host matrices, row-major
    const double* A;  // (M, M)
    const double* B;  // (N, N)
    const double* C;  // (M, N)
};

// every failure leaves through the one cleanup block at `done`
#define CAL_TRY(expr)                                                \
    do {                                                             \
        hipError_t _e = (expr);                                      \
        if (_e != hipSuccess) {                                      \
            err = std::string(#expr) + ": " + hipGetErrorString(_e); \
            rc = 1;                                                  \
            goto done;                                               \
        }                                                            \
    } while (0)

// status: 0 ok, 1 B not positive definite, 2 C' not positive definite, 3 normal equations not positive definite
inline int calibrate_run(const CalibInputs& in, double* fl_cor, double* X, int* status, std::string& err)
{
    int rc = 0;
    const int M = in.M, N = in.N, order = in.order, c = in.c;
    const int Npad = round_up(N, NB), P1 = Npad / NB;
    const int Mpad = round_up(M, NB), P2 = Mpad / NB;
    const size_t ld1 = (size_t)Npad + Mpad;
    const size_t ld2 = (size_t)Mpad + NB;
    const int nslab = (Npad + 255) / 256;
    const double scl = 2.0 / (in.lwl1 - in.lwl0);
    const double off = (in.lwl1 * -1.0 - in.lwl0 * 1.0) / (in.lwl1 - in.lwl0);   // numpy polyutils.mapparms
    *status = 0;

    double *dK1 = nullptr, *dK2 = nullptr, *dW = nullptr, *dR = nullptr, *dTmp = nullptr, *dCal = nullptr,
           *dFix = nullptr, *dVec = nullptr, *dGp = nullptr, *dMu = nullptr, *dM0 = nullptr, *dPart = nullptr,
           *dG = nullptr;
    MatAcc* dAcc = nullptr;
    MatAcc hacc;
    std::vector<double> G((size_t)NB * NB), m0(M, in.mu), L((size_t)(order + 1) * (order + 1)), rhs(order + 1);
    GpHost gall;
    for (int k = 0; k < 6; ++k) gall.v[k] = (k < 2 * c) ? in.gp[k] : 0.0;

    CAL_TRY(hipMalloc(&dK1, sizeof(double) * (size_t)Npad * ld1));
    CAL_TRY(hipMalloc(&dK2, sizeof(double) * (size_t)Mpad * ld2));
    CAL_TRY(hipMalloc(&dW, sizeof(double) * NB * NB));
    CAL_TRY(hipMalloc(&dR, sizeof(double) * (size_t)std::max(Npad, Mpad)));
    CAL_TRY(hipMalloc(&dAcc, sizeof(MatAcc) * (ACC_ROWS + 1)));      // block-row records + their total
    CAL_TRY(hipMalloc(&dVec, sizeof(double) * (size_t)(3 * M + 2 * N)));    // lwl_cal, fl_cal, sigma_cal, fl_fixed, sigma_fixed
    CAL_TRY(hipMalloc(&dMu, sizeof(double) * Mpad));
    CAL_TRY(hipMalloc(&dM0, sizeof(double) * Mpad));
    CAL_TRY(hipMalloc(&dPart, sizeof(double) * (size_t)nslab * Mpad));
    CAL_TRY(hipMalloc(&dG, sizeof(double) * NB * NB));
    {
        double* dLwlCal = dVec;
        double* dFlCal = dVec + M;
        double* dSigCal = dVec + 2 * M;
        double* dFlFix = dVec + 3 * M;
        double* dSigFix = dVec + 3 * M + N;
        CAL_TRY(hipMemcpy(dLwlCal, in.lwl_cal, sizeof(double) * M, hipMemcpyHostToDevice));
        CAL_TRY(hipMemcpy(dFlCal, in.fl_cal, sizeof(double) * M, hipMemcpyHostToDevice));
        CAL_TRY(hipMemcpy(dFlFix, in.fl_fixed, sizeof(double) * N, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_zero, dim3(2048), dim3(256), 0, 0, dK1, (size_t)Npad * ld1);
        hipLaunchKernelGGL(k_zero, dim3(1024), dim3(256), 0, 0, dK2, (size_t)Mpad * ld2);
        hipLaunchKernelGGL(k_zero, dim3(64), dim3(256), 0, 0, dG, (size_t)NB * NB);
        if (c > 0) {
            CAL_TRY(hipMalloc(&dCal, sizeof(double) * (size_t)c * M));
            CAL_TRY(hipMalloc(&dFix, sizeof(double) * (size_t)c * N));
            CAL_TRY(hipMalloc(&dGp, sizeof(double) * 6));
            CAL_TRY(hipMemcpy(dCal, in.lwls_cal, sizeof(double) * (size_t)c * M, hipMemcpyHostToDevice));
            CAL_TRY(hipMemcpy(dFix, in.lwls_fixed, sizeof(double) * (size_t)c * N, hipMemcpyHostToDevice));
            CAL_TRY(hipMemcpy(dSigCal, in.sigma_cal, sizeof(double) * M, hipMemcpyHostToDevice));
            CAL_TRY(hipMemcpy(dSigFix, in.sigma_fixed, sizeof(double) * N, hipMemcpyHostToDevice));
            CAL_TRY(hipMemcpy(dGp, in.gp, sizeof(double) * 2 * c, hipMemcpyHostToDevice));
            // B = sum_k K_k(fixed) + sigma_fixed^2 (script :170-176), upper tiles, identity padding
            dim3 grid(P1 * (P1 + 1) / 2, 1);
            if (c == 1) hipLaunchKernelGGL(k_fill_sym<1>, grid, dim3(256), 0, 0, dK1, (size_t)0, (int)ld1, N, P1, dFix, dGp, dSigFix, 1);
            else if (c == 2) hipLaunchKernelGGL(k_fill_sym<2>, grid, dim3(256), 0, 0, dK1, (size_t)0, (int)ld1, N, P1, dFix, dGp, dSigFix, 1);
            else hipLaunchKernelGGL(k_fill_sym<3>, grid, dim3(256), 0, 0, dK1, (size_t)0, (int)ld1, N, P1, dFix, dGp, dSigFix, 1);
            // C^T[j][i] = sum_k K_k(cal_i, fixed_j) (script :160-166): rows = fixed grid, columns = epoch grid
            launch_region_c((hipStream_t)0, c, dK1, ld1, Npad, N, M, dFix, (size_t)N, dCal, (size_t)M, gall, 0, 0.0);
            // A = sum_k K_k(cal) + sigma_cal^2 (script :150-158), written into pass 2's buffer
            launch_region_c((hipStream_t)0, c, dK2, ld2, 0, M, M, dCal, (size_t)M, dCal, (size_t)M, gall, 1, 0.0);
            hipLaunchKernelGGL(k_add_diag_sq, dim3((M + 255) / 256), dim3(256), 0, 0, dK2, ld2, M, dSigCal);
        } else {
            CAL_TRY(hipMemcpy2D(dK1, sizeof(double) * ld1, in.B, sizeof(double) * N, sizeof(double) * N, N, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_pad_identity, dim3((Npad - N + 255) / 256 + 1), dim3(256), 0, 0, dK1, ld1, N, Npad);
            CAL_TRY(hipMalloc(&dTmp, sizeof(double) * (size_t)M * N));
            CAL_TRY(hipMemcpy(dTmp, in.C, sizeof(double) * (size_t)M * N, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_transpose_in, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, 0, dK1, ld1, Npad, dTmp,
                               (size_t)N, M, N);
            CAL_TRY(hipMemcpy2D(dK2, sizeof(double) * ld2, in.A, sizeof(double) * M, sizeof(double) * M, M, hipMemcpyHostToDevice));
        }
        CAL_TRY(hipGetLastError());

        // ---- pass 1
        hipLaunchKernelGGL(k_init_rhs, dim3((Npad + 255) / 256, 1), dim3(256), 0, 0, dR, Npad, N, dFlFix, in.mu, dAcc);
        factor_augmented((hipStream_t)0, dK1, ld1, P1, Mpad / NB, dW, dR, Npad, dAcc);
        hipLaunchKernelGGL(k_acc_total, dim3(1), dim3(1), 0, 0, dAcc, P1, dAcc + ACC_ROWS);
        CAL_TRY(hipGetLastError());
        CAL_TRY(hipMemcpy(&hacc, dAcc + ACC_ROWS, sizeof(MatAcc), hipMemcpyDeviceToHost));
        if (hacc.info != 0.0) { *status = 1; goto done; }
        CAL_TRY(hipMemcpy(dM0, m0.data(), sizeof(double) * M, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_gemv_t_partial, dim3((M + 127) / 128, nslab), dim3(256), 0, 0, dK1 + Npad, ld1, Npad, M, dR, dPart);
        hipLaunchKernelGGL(k_gemv_finish, dim3((M + 255) / 256), dim3(256), 0, 0, dPart, nslab, M, dM0, dMu);     // fl'
        hipLaunchKernelGGL(k_syrk_sub, dim3(P2, P2), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, dK1 + Npad, ld1, Npad, dK2, ld2);  // C'
        hipLaunchKernelGGL(k_pad_identity, dim3((Mpad - M + 255) / 256 + 1), dim3(256), 0, 0, dK2, ld2, M, Mpad);
        hipLaunchKernelGGL(k_calib_aug, dim3((M + 255) / 256), dim3(256), 0, 0, dK2, ld2, Mpad, M, order, dLwlCal, dFlCal, off,
                           scl, dMu);
        CAL_TRY(hipGetLastError());

        // ---- pass 2
        hipLaunchKernelGGL(k_init_rhs, dim3((Mpad + 255) / 256, 1), dim3(256), 0, 0, dR, Mpad, M, dMu, 0.0, dAcc);
        factor_augmented((hipStream_t)0, dK2, ld2, P2, 1, dW, dR, Mpad, dAcc);
        hipLaunchKernelGGL(k_acc_total, dim3(1), dim3(1), 0, 0, dAcc, P2, dAcc + ACC_ROWS);
        CAL_TRY(hipGetLastError());
        CAL_TRY(hipMemcpy(&hacc, dAcc + ACC_ROWS, sizeof(MatAcc), hipMemcpyDeviceToHost));
        if (hacc.info != 0.0) { *status = 2; goto done; }
        hipLaunchKernelGGL(k_syrk_sub, dim3(1, 1), dim3(GEMM_THREADS), GEMM_LDS_BYTES, 0, dK2 + Mpad, ld2, Mpad, dG, (size_t)NB);
        CAL_TRY(hipGetLastError());
        CAL_TRY(hipMemcpy(G.data(), dG, sizeof(double) * NB * NB, hipMemcpyDeviceToHost));
    }
    {
        // normal equations (order+1 unknowns): left = D^T C'^-1 D, right = D^T C'^-1 fl'  (G holds their negatives)
        const int n = order + 1;
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) L[(size_t)i * n + j] = -G[(size_t)i * NB + j];
            rhs[i] = -G[(size_t)i * NB + n];
        }
        for (int j = 0; j < n; ++j) {              // in-place lower Cholesky
            double d = L[(size_t)j * n + j];
            for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
            if (!(d > 0.0)) { *status = 3; goto done; }
            d = std::sqrt(d);
            L[(size_t)j * n + j] = d;
            for (int i = j + 1; i < n; ++i) {
                double s = L[(size_t)i * n + j];
                for (int k = 0; k < j; ++k) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
                L[(size_t)i * n + j] = s / d;
            }
        }
        for (int i = 0; i < n; ++i) {
            double s = rhs[i];
            for (int k = 0; k < i; ++k) s -= L[(size_t)i * n + k] * X[k];
            X[i] = s / L[(size_t)i * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {
            double s = X[i];
            for (int k = i + 1; k < n; ++k) s -= L[(size_t)k * n + i] * X[k];
            X[i] = s / L[(size_t)i * n + i];
        }
        double T[CAL_MAX_ORDER + 1];
        for (int i = 0; i < M; ++i) {              // fl_cor = D X  (:619)
            cheb_row(in.lwl_cal[i], off, scl, order, T);
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += in.fl_cal[i] * T[k] * X[k];
            fl_cor[i] = s;
        }
    }
done:
    (void)hipDeviceSynchronize();
    (void)hipFree(dK1); (void)hipFree(dK2); (void)hipFree(dW); (void)hipFree(dR); (void)hipFree(dAcc); (void)hipFree(dTmp);
    (void)hipFree(dCal); (void)hipFree(dFix); (void)hipFree(dVec); (void)hipFree(dGp); (void)hipFree(dMu); (void)hipFree(dM0);
    (void)hipFree(dPart); (void)hipFree(dG);
    return rc;
}

}  // namespace psoap
