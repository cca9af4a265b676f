// Kernel parameter block shared by the two 3x3 f16 MFMA kernels (conv6.hip, conv7.hip); filled by launch_conv6.
#pragma once
#include "common.h"
namespace dpir {

// Fused hop conv1 -> GroupNorm + FiLM + SiLU -> conv2 inside a ResBlock (round 4, conv7 only): conv1's epilogue does not store its fp32
// output; it adds its per-group sums to per-image accumulators (fixed point: integer addition is associative, so the result does not
// depend on the arrival order -- replays stay bitwise reproducible), waits until every workgroup of its image has arrived, folds the
// GroupNorm parameters for its own 128 channels and writes conv2's operand planes (normalised, activated, split into f16 hi / lo) itself.
struct Conv6Emit {
    char* hi = nullptr; char* lo = nullptr; int C8 = 0;      // conv2's planes [n][C8][H][W][16 B]; lo == null: f16x1
    const float* gamma = nullptr; const float* beta = nullptr;
    const float* film = nullptr; int film_stride = 0, film_off = 0, frows = 0; const StepDev* fstep = nullptr;
    long long* acc = nullptr;          // [B][32 groups][2] {sum * 2^20, sum of squares * 2^12}, zero before the launch
    unsigned* cnt = nullptr;           // [B][n_co_blocks] arrival counters, zero before the launch
    unsigned long long* range_ctr = nullptr;   // f16 operand range guard; a barrier time-out adds 2^40 to it
    int spin_limit = 250000;           // polls (~4 us each at the default interval: about a second) before a waiting workgroup gives up and flags the guard word
    int sleep_sel = 5;                 // poll interval of the wait: s_sleep 2 / 8 / 16 / 32 / 64 / 127 (default) / 2 x 127 / 4 x 127, x 64 cycles (DPIR_FUSE_SLEEP=0..7, A/B knob).
                                       // Round 5, two boxes, interleaved (profiles/r05/fused_hop_poll_interval_ab.log): 16 (rounds 4's value) 18.81 / 19.40 ms per forward, 127: 18.72 /
                                       // 19.31, 254: 19.31, 508: 19.42 -- up to 255 waiting workgroups polling ONE word every ~0.5 us delay the arrivals' own atomics
    int expect_extra = 0;              // tests only (DPIR_FUSE_EXPECT_EXTRA): arrivals waited for beyond the image's workgroups -> forces the time-out
};

struct Conv6K {
    const char* xhi; const char* xlo;      // blocked split activations [n][C8][H][W][16 B]
    int C8;
    const char* w16; const float* bias; float* out; const float* res; int res_mode;
    int B, Cout, H, W;
    int n_chunks_total;
    int tiles_x, tiles_y, n_co_blocks;
    int ksplit, chunks_per_split;
    float* partial;
    const float* zeros;
    float out_scale;
    const float* out_scale_dev;            // optional device scalar multiplied into out_scale (dgrad: undoes the run-time scaling of dY)
    float2* stat; int stat_slots;          // per-(image, channel, slot) {sum, sum of squares} of the stored values, or null
    Conv6Emit em;                          // em.hi != null: fused emission of the next convolution's operand planes (no fp32 output)
};

// conv7.hip: every case of launch_conv6 (all three geometries, split-K slabs, f16x1, dgrad scale) with the workgroup tile cut as
// 64 co x 128 px per wave; blocks = pixel tiles x co-blocks x ksplit
Status launch_conv7(hipStream_t s, const Conv6K& k, int blocks, bool x1);
int conv7_emit_capacity();     // resident EMIT workgroups on this device (CUs x occupancy)

}  // namespace dpir
