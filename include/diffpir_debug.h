/* diffpir_debug.h -- development-only entry points of libdiffpir_hip.so (not part of the drop-in boundary). */
#ifndef DIFFPIR_DEBUG_H
#define DIFFPIR_DEBUG_H
#include "diffpir_engine.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Average time (ms) of one convolution launch of the given shape on synthetic operands; `dbg` disables parts of the
 * kernel for ablation (1 no MFMA, 2 no prologue transform, 4 no global loads, 8 no LDS stores, 16 no epilogue stores);
 * mode 0 plain / 1 nearest-up source / 2 avg-pool source; with_prm: fused GroupNorm+SiLU prologue on/off. */
int dpir_debug_conv_bench(dpir_engine* e, int B, int Cin, int Cout, int H, int W, int ks, int mode, int with_prm,
                          int dbg, int iters, double* ms_out);
/* Concurrency probe: `blocks` workgroups of `threads` threads park a pattern in `lds_bytes` of LDS and in 32 registers per
 * thread, wait `spin_ticks` of the 100 MHz wall clock, verify; *bad_out = LDS mismatches (low 32 bits) + register mismatches << 32. */
int dpir_debug_victim(dpir_engine* e, int lds_bytes, int threads, int blocks, long long spin_ticks, int iters, unsigned long long* bad_out);
/* ALU probe: threads run `iters_in_kernel` dependent exact-integer fp32 operations (mode 0 v_add_f32, 1 v_pk_add_f32, 2 v_pk_fma_f32,
 * 3 v_pk_mul_f32) and check the closed-form result; *bad_out = number of threads whose result was wrong. */
int dpir_debug_victim_alu(dpir_engine* e, int mode, int blocks, int iters_in_kernel, int launches, unsigned long long* bad_out);
/* Register-FFT probe (csrc/dbg_fft.inc): same source built with (pk) and without (nopk) packed-fp32 instructions; *bad_out =
 * threads whose two identical computations disagreed. */
int dpir_debug_victim_fft_pk(dpir_engine* e, int blocks, int iters_in_kernel, int launches, unsigned long long* bad_out);
int dpir_debug_victim_fft_nopk(dpir_engine* e, int blocks, int iters_in_kernel, int launches, unsigned long long* bad_out);
/* conv7 (csrc/conv7.hip: 64 co x 128 px per wave, weights straight into registers; every launch class) against conv6 (csrc/conv6.hip,
 * still built for the 8 x 32 geometry, W >= 32) on the same split planes and weight pack, with the residual form res_mode (-1 none,
 * 0 same shape, 1 half resolution, 2 double resolution).  x1: f16x1 planes / products; split: allow split-K (the partial slabs are
 * compared instead of output + fused GroupNorm statistics; *ksplit_out = slabs); scaled: a device output scale (the dgrad route).
 * *mismatches_out = elements whose bits differ (expected 0: same MFMA order per accumulator), *maxdiff_out their largest absolute
 * difference, *ms6_out / *ms7_out the average launch times over `iters` back-to-back launches. */
int dpir_debug_conv7_check(dpir_engine* e, int B, int Cin, int Cout, int H, int W, int res_mode, int x1, int split, int scaled, int iters,
                           double* ms6_out, double* ms7_out, unsigned long long* mismatches_out, float* maxdiff_out, int* ksplit_out);
/* 1 when a 3x3 launch of this shape may take conv7's fused GroupNorm hop (Conv6Emit), 0 when the dispatch falls back to the unfused path;
 * *capacity_out = resident EMIT workgroups of the device (CUs x occupancy) the waiting-set limit is derived from. */
int dpir_debug_conv7_emit_supported(dpir_engine* e, int B, int Cout, int H, int W, int* capacity_out);
#ifdef __cplusplus
}
#endif
#endif
