// Scheduling pieces shared by the two single-launch prox kernels (fft3.hip: fft2's bodies; fft5.hip: the wave-per-transform bodies): the layout of the
// scheduling words, the bounded dependency wait and the completion signal.  See fft3.hip for the design (tickets, one queue per XCD, self-cleaning state).
#pragma once
#include <hip/hip_runtime.h>

namespace dpir {

// scheduling words (unsigned): [0] next unclaimed plane, [2] workgroups that left, [16 + 16 x] ticket counter of XCD x,
// [256 + x * nr_max + r] base plane + 1 of round r of XCD x, then rowdone[P], coldone[P]
constexpr int PF_TICKET0 = 16, PF_ROUND0 = 256;

#ifndef DPIR_PROX_ACQ
#define DPIR_PROX_ACQ 0      // 1: the waiting workgroup drops its CU's L1 (agent-scope acquire) after the poll
#endif
#ifndef DPIR_PROX_REL
#define DPIR_PROX_REL 0      // 1: the producing workgroup writes the XCD's L2 back (agent-scope release) before it signals
#endif

// Polls read with SYSTEM scope (global_load ... sc0 sc1: served by memory, not by this XCD's L2).  An agent-scope relaxed load bypasses only the L1: the
// XCD's L2 may still hold the line from the PREVIOUS launch's polling (the last workgroup zeroes the words from whatever XCD it runs on, with write-through
// stores that do not touch other XCDs' L2s), so a waiting job could read last launch's final count and start before its dependencies -- measured: with few
// planes one inverse-row job in ~10 launches ran ahead of its column jobs; agent-scope fences on both sides did not help, this does.
__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned add_relaxed(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr unsigned PF_SPIN_LIMIT = 4u << 20;        // x >= 0.3 us per poll: seconds, never a hang
// all threads call; thread 0 polls `ctr` until it reaches `target`
struct DepWait {
    const unsigned* ctr; unsigned target; unsigned* err; int flags = 0;
    __device__ __forceinline__ void operator()() const {
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (ld_relaxed(ctr) < target) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > PF_SPIN_LIMIT) { atomicOr(err, 1u); break; }
            }
            if (DPIR_PROX_ACQ || (flags & 1)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
};
// all threads call after the job's last store
__device__ __forceinline__ void job_done(unsigned* ctr, int flags = 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its stores have reached the L2
    __syncthreads();
    if (threadIdx.x == 0) {
        if (DPIR_PROX_REL || (flags & 2)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        add_relaxed(ctr, 1u);
    }
}

// waves per SIMD the register allocator must leave room for (= resident 256-thread workgroups per CU): the column job holds 2 RJ values,
// RJ prefetched FBFy values and the transform's temporaries per thread

// Draws the next job of this workgroup's XCD queue.  Thread 0 only; returns (base plane of the round, index in the round); base >= P: nothing left.
__device__ __forceinline__ void draw_ticket(unsigned* sync, unsigned* err, unsigned xcc, int nr_max, unsigned per_round, int K, int P, unsigned* base_out, unsigned* idx_out) {
    unsigned* ticket = sync + PF_TICKET0 + 16 * xcc;
    unsigned* rbase = sync + PF_ROUND0 + xcc * (unsigned)nr_max;
    const unsigned tk = add_relaxed(ticket, 1u);
    const unsigned r = tk / per_round, idx = tk - r * per_round;
    unsigned base = (unsigned)P;
    if (r < (unsigned)nr_max) {
        if (idx == 0) {                       // first ticket of a round: claim K planes for this XCD
            base = add_relaxed(sync, (unsigned)K);
            __hip_atomic_store(rbase + r, base + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            unsigned spins = 0, v;
            while ((v = ld_relaxed(rbase + r)) == 0u) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > PF_SPIN_LIMIT) { atomicOr(err, 2u); v = (unsigned)P + 1u; break; }
            }
            base = v - 1u;
        }
    }
    *base_out = base; *idx_out = idx;
}
// The last workgroup to leave zeroes the scheduling words for the next launch (every other workgroup is past its last access).  All threads call;
// `flag`: one LDS word.
__device__ __forceinline__ void leave_and_clean(unsigned* sync, int nr_max, int P, volatile unsigned* flag) {
    if (threadIdx.x == 0) *flag = add_relaxed(sync + 2, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (*flag) {
        const unsigned words = (unsigned)(PF_ROUND0 + 8 * nr_max + 2 * P);
        for (unsigned i = threadIdx.x; i < words; i += blockDim.x) __hip_atomic_store(sync + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }          // HW_REG_XCC_ID[3:0]

}  // namespace dpir
