// fft5: the 256 x 256 FFT data-fidelity prox (utils/utils_sisr.py:65-75; in dpir_run_loop with the eps -> x0 prologue, gaussian_diffusion.py:297,328-333,
// and the re-noise epilogue, main_ddpir.py:448-456) as ONE persistent launch of the wave-per-transform bodies (fft4_body.h).  Scheduling as in fft3.hip:
// ticketed jobs, one queue per XCD (a workgroup reads HW_REG_XCC_ID and draws from its own XCD's queue; queues claim planes in rounds from one global
// counter), per-plane arrival counters, self-cleaning state, bounded spins.  What it buys over the three launches of fft4.hip:
//   * no launch boundaries (1.9 us each on this machine, tools/micro/stream_probe.hip) and no lock-step: planes are in different phases at the same time;
//   * every job of a plane runs on the XCD that claimed it, so the plane's 264 KB column-major half spectrum is written by the row jobs, transformed in place
//     by the column jobs and read by the inverse-row jobs through that XCD's own 4 MiB L2 (plain stores, `s_waitcnt vmcnt(0)`, counter; relaxed poll, loads);
//   * a waiting job has already requested everything that does not depend on what it waits for (the solve's three spectra; x_t / the blend base).
// Jobs of a plane (512-thread workgroups = 8 waves): 16 row jobs (8 row pairs each), ceil(ceil(NC / 4) / 2) column jobs (two 4-slot items each), 16 inverse-row jobs.
#include "fft4_body.h"
#include "prox_sched.h"
#include <algorithm>
#include <stdlib.h>

namespace dpir {
namespace {

constexpr int RJOBS5 = N4 / 2 / RW;       // 16 row jobs per plane

__host__ __device__ inline int col_jobs5(int NC) { return ((NC + WAVES - 1) / WAVES + 1) / 2; }
// float2 elements of dynamic LDS before the control words: the waves' tiles + max(rows' slot tile, columns' fold area)
__host__ __device__ inline size_t lds5_elems(int NC, int sf) {
    const size_t tile = (size_t)std::max(NC, N4 / 2 + 1) * TST * 2;                 // float4 entries = 2 float2
    const size_t fold = sf > 1 ? (size_t)RW * (N4 / sf) : 0;
    return (size_t)RW * WLDS + std::max(tile, fold);
}

// The three job bodies as REAL functions (not inlined): inlined into one loop they share a register allocation that holds every body's loop-invariant
// addresses at once (124 VGPRs against 40-52 for each body alone); as calls the kernel's count is the largest callee's plus the few values live across it.
// They read the launch arguments straight from the KERNARG segment (constant address space -> scalar loads at the point of use); handing them a reference
// to the by-value kernel parameter makes hipcc copy the 300-byte struct into every lane's scratch and read it back with vector loads (measured: 87 us).
typedef const __attribute__((address_space(4))) ProxFusedArgs* KArgs;
__device__ __attribute__((noinline)) void job_rows(float2* sm4, size_t wg, KArgs a) {
    rows4_body<false>(sm4, wg, a->x, a->pa, a->pb, a->pm, a->sp, a->hbuf, a->WP, a->tw, WaveTw{}, RowsFuse{a->fu.eps6, a->fu.out_ch}, a->slot_col);
}
template <int SF>
__device__ __attribute__((noinline)) void job_cols(float2* sm4, int plane, int cj, KArgs a, const unsigned* ctr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = wave >> 2;     // waves 0-3: item 2 cj, waves 4-7: item 2 cj + 1 (possibly past the last: dead)
    float2* fold = sm4 + RW * WLDS + half * WAVES * (N4 / SF);
    const SolveArgs so{a->solve.FB, a->solve.F2B, a->solve.FBFy, a->solve.alpha, a->solve.sf, a->solve.sp, a->solve.invW, a->solve.slot_col};
    const WaveTw w = wave_tw_load(a->tw, lane);
    cols4_item_body<SF == 1 ? 2 : 3, SF>(sm4 + wave * WLDS, fold, plane, 2 * cj + half, wave & 3, a->hbuf, so, a->WP, w, DepWait{ctr, RJOBS5, a->err, a->flags});
}
__device__ __attribute__((noinline)) void job_irows(float2* sm4, size_t wg, KArgs a, const unsigned* ctr, unsigned target) {
    const RenoiseFuse rn{a->rn.xt, a->rn.sp, a->rn.lp, a->rn.n1, a->rn.n2, a->rn.stride, a->rn.with_n1};
    irows4_body<false>(sm4, wg, a->hbuf, a->out, a->scale, a->oa, a->ob, a->blend_base, a->g, a->WP, a->tw, WaveTw{}, rn, a->col_slot, DepWait{ctr, target, a->err, a->flags});
}

template <int SF, int OCC>
__global__ __launch_bounds__(RTHREADS, 2 * OCC) void prox_wave_fused_kernel(const ProxFusedArgs a, int lds_elems) {
    extern __shared__ __attribute__((aligned(16))) float2 sm4[];
    volatile unsigned* ctl = reinterpret_cast<volatile unsigned*>(sm4 + lds_elems);      // [0] base plane of the round, [1] index in the round
    const KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();      // `a` is the first kernel parameter: offset 0
    const int P = a.P, K = a.K, NC = a.WP;
    const int cjobs = col_jobs5(NC);
    const unsigned JP = 2 * RJOBS5 + cjobs, per_round = JP * (unsigned)K;
    const unsigned xcc = xcc_id();
    unsigned* rowdone = a.sync + PF_ROUND0 + 8 * a.nr_max;
    unsigned* coldone = rowdone + P;
    for (;;) {
        if (threadIdx.x == 0) {
            unsigned base, idx;
            draw_ticket(a.sync, a.err, xcc, a.nr_max, per_round, K, P, &base, &idx);
            ctl[0] = base; ctl[1] = idx;
        }
        __syncthreads();
        const unsigned base = ctl[0], idx = ctl[1];
        __syncthreads();
        if (base >= (unsigned)P) break;
        const unsigned n_rows = (unsigned)K * RJOBS5, n_cols = (unsigned)K * cjobs;
        if (idx < n_rows) {
            const int plane = (int)(base + idx / RJOBS5), j = (int)(idx % RJOBS5);
            if (plane < P) {
                job_rows(sm4, (size_t)plane * RJOBS5 + j, ka);
                job_done(rowdone + plane, a.flags);
            }
        } else if (idx < n_rows + n_cols) {
            const unsigned i2 = idx - n_rows;
            const int plane = (int)(base + i2 / cjobs), cj = (int)(i2 % cjobs);
            if (plane < P) {
                job_cols<SF>(sm4, plane, cj, ka, rowdone + plane);
                job_done(coldone + plane, a.flags);
            }
        } else {
            const unsigned i3 = idx - n_rows - n_cols;
            const int plane = (int)(base + i3 / RJOBS5), j = (int)(i3 % RJOBS5);
            if (plane < P)
                job_irows(sm4, (size_t)plane * RJOBS5 + j, ka, coldone + plane, (unsigned)cjobs);
        }
        __syncthreads();          // the next job reuses the LDS areas
    }
    leave_and_clean(a.sync, a.nr_max, P, ctl);
}

template <int SF, int OCC>
Status launch_V(hipStream_t s, const ProxFusedArgs& a, int cus, int wgs_per_cu) {
    const int NC = a.WP;
    const size_t elems = lds5_elems(NC, SF);
    const size_t lds = elems * sizeof(float2) + 16;
    auto fn = prox_wave_fused_kernel<SF, OCC>;
    static LdsAttrOnce attr;
    DPIR_HIP(attr.set(reinterpret_cast<const void*>(fn), 160 * 1024));
    const int JP = 2 * RJOBS5 + col_jobs5(NC);
    // no more workgroups per XCD than one round has jobs (an XCD whose workgroups start first must not claim a second round before the others claimed their first)
    const long long want = 8ll * a.K * JP;
    const unsigned G = (unsigned)std::max(8ll, std::min((long long)cus * wgs_per_cu, want));
    hipLaunchKernelGGL(fn, dim3(G), dim3(RTHREADS), lds, s, a, (int)elems);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
template <int SF>
Status launch_T(hipStream_t s, const ProxFusedArgs& a, int cus) {
    // resident 512-thread workgroups per CU: 3 at <= 80 VGPRs (DPIR_FFT5_WGS: development switch 2 / 3 / 4)
    static const int wgs = getenv("DPIR_FFT5_WGS") ? std::max(1, std::min(4, atoi(getenv("DPIR_FFT5_WGS")))) : 3;
    switch (wgs) {
        case 4: return launch_V<SF, 4>(s, a, cus, 4);
        case 2: return launch_V<SF, 2>(s, a, cus, 2);
        case 1: return launch_V<SF, 1>(s, a, cus, 1);
        default: return launch_V<SF, 3>(s, a, cus, 3);
    }
}

}  // namespace

// scheduling words for P planes in rounds of K (zero before the first launch; the kernel cleans up after itself)
size_t prox_wave_fused_sync_words(int P, int K) { const int nr = (P + K - 1) / K + 12; return (size_t)PF_ROUND0 + 8 * (size_t)nr + 2 * (size_t)P; }

// a: as for launch_prox_fused, with hbuf / solve spectra COLUMN-major (fft4.hip) and WP = stored columns per plane
Status launch_prox_wave_fused(hipStream_t s, const ProxFusedArgs& a, int sf, int cus) {
    if (!a.sync || !a.err || a.K < 1 || a.nr_max != (a.P + a.K - 1) / a.K + 12) return invalid("prox_wave_fused: bad scheduling state");
    if (sf > 1 && (!a.solve.invW || !a.solve.slot_col || a.solve.sf != sf || a.WP % sf)) return invalid("prox_wave_fused: bad sf > 1 arguments");
    if (sf == 1) return launch_T<1>(s, a, cus);
    if (sf == 2) return launch_T<2>(s, a, cus);
    if (sf == 4) return launch_T<4>(s, a, cus);
    return invalid("prox_wave_fused: sf must be 1, 2 or 4");
}

}  // namespace dpir
