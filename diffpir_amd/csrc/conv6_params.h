// Kernel parameter block shared by the two 3x3 f16 MFMA kernels (conv6.hip, conv7.hip); filled by launch_conv6.
#pragma once
#include "common.h"
namespace dpir {

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
};

// conv7.hip: every case of launch_conv6 (all three geometries, split-K slabs, f16x1, dgrad scale) with the workgroup tile cut as
// 64 co x 128 px per wave; blocks = pixel tiles x co-blocks x ksplit
Status launch_conv7(hipStream_t s, const Conv6K& k, int blocks, bool x1);

}  // namespace dpir
