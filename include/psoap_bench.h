/* psoap_bench.h -- C ABI of libpsoap_bench.so: measurement kernels, NOT part of the product library.
 *
 * bench.py, tools/ and one GPU test load it to state measured ceilings of the device beside the spec peaks
 * (fp64 MFMA issue rate, streaming HBM bandwidth, the MFMA tile engine alone, one in-block factorisation alone)
 * and to check the batched exp() of the fused-fill epilogue against the device library's exp() bit for bit.
 * The reference (iancze/PSOAP) has no counterpart: these are roofline instruments.
 * Every function returns 0 on success; psoap_bench_last_error() describes the last failure of the calling thread. */
#ifndef PSOAP_BENCH_H
#define PSOAP_BENCH_H
#ifdef __cplusplus
extern "C" {
#endif

const char *psoap_bench_last_error(void);

/* fp64 MFMA / HBM micro-benchmarks used to state the measured peaks beside the
 * spec peaks in bench.py (results in TFLOP/s and GB/s). */
int psoap_microbench_mfma_f64(int device, double *tflops);
int psoap_microbench_hbm(int device, double *write_gbs, double *copy_gbs);
/* The MFMA tile engine alone (512 workgroups, K = 4096).  variant: 9 / 8 = the production engine
 * (LDS-DMA staging) with all tiles reading the same L2-resident strips / every tile streaming its own
 * B strip from HBM (the factorisation's pattern); 1 / 0 = the same two patterns with the earlier
 * register-staged engine (global_load -> ds_write), kept for comparison; 16 + abl = loop ablations
 * (abl bit 0: no staging traffic, bit 1: no workgroup barrier, bit 2: no LDS fragment reads). */
int psoap_microbench_tile_engine(int device, int variant, double *tflops);
/* Self-check of the batched exp() the fused-fill epilogue uses for non-positive arguments: counts the
 * x[i] (n a multiple of 4) whose result differs in any bit from the device library's exp(). */
int psoap_microbench_exp_check(int device, long long n, const double *x, long long *mismatches);
/* How the fp64 MFMAs of one wave and the fp64 vector arithmetic of another wave on the same SIMD share the
 * machine (the fused-fill epilogue of one workgroup runs beside the K-loop of its neighbour).  mode bit 0: waves
 * 0-3 of every workgroup issue iters_mfma x 4 MFMAs; bit 1: waves 4-7 evaluate iters_valu batches of 4 exp();
 * bit 2: plain FMA chains instead of exp.  out3: mean MFMA-wave time (us), mean vector-wave time (us), fraction
 * of workgroups in which wave w and wave w + 4 shared a SIMD. */
int psoap_microbench_mix(int device, int mode, int iters_mfma, int iters_valu, double *out3);
/* One workgroup factoring a 128 x 128 tile (potrf_blocked), microseconds per factorisation; ablate 0 =
 * the shipped routine, 1-3 = timing ablations (no in-wave 16 x 16 factorisation / no MFMA phases / no W output). */
int psoap_microbench_potrf(int device, int ablate, double *usec);

/* Coherence litmus tests between two XCDs of the device (round 6; psoap_amd/csrc/litmus_kernels.hpp, tools/litmus.py):
 * what a reader wave sees of a 256-byte unit a writer wave on another XCD (same_xcd = 0) or on its own (1) has just
 * rewritten and announced through returning atomics.  plant_mode: how the reader touched the unit before the write (0 not
 * at all, 1 sc1 load, 2 plain load, 3 acquire + plain load); writer_mode: 0 sc1 stores, 1 plain stores + release fence,
 * 2 sc0 sc1 stores; reader_mode: 0 sc1 load, 1 acquire + sc1 load, 2 acquire + plain load, 3 returning atomic, 4 plain
 * load, 5 acquire + LDS-DMA load, 6 sc0 sc1 load; background != 0: the rest of the launch streams 512 MiB through the
 * L2s.  out8: iterations, stale plants, stale reads, reads that never turned fresh, longest / summed 100 MHz ticks until
 * fresh, XCC id of the reader, of the writer. */
int psoap_litmus_l2(int device, int plant_mode, int writer_mode, int reader_mode, int same_xcd, int iters, int background,
                    unsigned long long *out8);
/* Does the write-back of a line one XCD has partly rewritten with plain stores overwrite what another XCD wrote through
 * to the line's other words meanwhile?  out8[2]: iterations in which it did, out8[1]: in which the plain-stored word was lost. */
int psoap_litmus_writeback(int device, int iters, unsigned long long *out8);

#ifdef __cplusplus
}
#endif
#endif /* PSOAP_BENCH_H */
