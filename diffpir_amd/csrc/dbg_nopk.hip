#define PROBE_KERNEL victim_fft_nopk_kernel
#define PROBE_API dpir_debug_victim_fft_nopk
#include "dbg_fft.inc"
