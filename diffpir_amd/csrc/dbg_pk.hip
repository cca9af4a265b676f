#define PROBE_KERNEL victim_fft_pk_kernel
#define PROBE_API dpir_debug_victim_fft_pk
#include "dbg_fft.inc"
