// LDS-DMA idioms shared by the f16 MFMA convolution kernels (conv5.hip, conv6.hip): buffer-descriptor loads straight into LDS and the
// counted waits that order them.  Device pass only (the buffer-resource builtin type does not exist in hipcc's host pass).
#pragma once
#if defined(__HIP_DEVICE_COMPILE__)
namespace dpir {

// LDS-DMA through a buffer descriptor: `buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds`.  The descriptor (4 SGPRs) and
// the scalar offset carry everything wave-uniform, so a DMA instruction costs ONE live VGPR (the per-lane byte offset) and
// no vector address arithmetic; a per-lane offset >= num_records is out of range and the hardware writes ZEROS for that
// lane -- which is how the halo positions outside the image (and the padding of the last piece) are produced.
#define BLDS6(rsrc, dst, voff, soff) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(dst), 16, (voff), (soff), 0, 0)
constexpr unsigned kOutOfRange = 0xFFFFFFFFu;
// Descriptor from values that ARE wave-uniform but that the compiler cannot always prove so: without the readfirstlane it
// wraps every buffer operation in a "waterfall" loop (v_readfirstlane x4, compare, s_and_saveexec, op, loop).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_uniform(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

// Bit casts of SCALARS.  `__builtin_bit_cast(float, v.y)` on an element of an ext_vector_type value is miscompiled by this clang
// (every element reads element 0; seen as a b128 buffer load narrowed to one dword) -- elements go through these helpers.
__device__ __forceinline__ float as_f32(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ unsigned as_u32(float f) { return __builtin_bit_cast(unsigned, f); }

// s_waitcnt with only the vector-memory counter constrained (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] at [15:14])
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

// Workgroup barrier that waits for this wave's LDS operations only.  `__syncthreads()` carries a workgroup-scope fence, which the
// compiler lowers to `s_waitcnt vmcnt(0) lgkmcnt(0)`: at a chunk boundary that drains the wave's whole weight ring (the pieces of the
// next D taps, issued moments ago) although only the activation pieces -- covered by the counted wait in front -- have to be there.
__device__ __forceinline__ void barrier_lds_only() {
    __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14));      // lgkmcnt(0), vmcnt / expcnt unconstrained
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

}  // namespace dpir
#endif
