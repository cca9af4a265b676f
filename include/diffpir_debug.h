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
#ifdef __cplusplus
}
#endif
#endif
