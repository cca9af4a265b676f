// fft3: the FFT data-fidelity prox (utils/utils_sisr.py:65-75 `data_solution`; inside dpir_run_loop also the eps -> x0 prologue,
// gaussian_diffusion.py:297,328-333, and the re-noise epilogue, main_ddpir.py:448-456) as ONE persistent launch instead of the three
// dependent launches of fft2.hip.  The three passes keep their bodies (fft2_body.h: same arithmetic, same bits); what changes is how
// they are scheduled and where the half-spectrum intermediate lives:
//
//   * jobs, not grids.  Per plane: RJOBS row jobs (16 row pairs each), `strips` column jobs (16 columns each: forward FFT -> closed-form
//     solve -> inverse FFT), RJOBS inverse-row jobs.  A column job of plane p waits for p's row jobs only, an inverse-row job for p's
//     column jobs only -- per-plane arrival counters, no grid barrier, no launch boundary; everything a job needs that does NOT depend
//     on another job (FBFy for the solve, x_t / the blend base for the epilogue) is requested BEFORE the wait.
//   * tickets.  A workgroup draws job after job from a queue; a job's dependencies always carry LOWER ticket numbers of the same queue,
//     so whoever holds them is already running (or done): progress never depends on how many workgroups are resident at once.
//   * one queue per XCD.  A workgroup reads its XCC id (s_getreg HW_REG_XCC_ID) and draws from that XCD's queue only; a queue claims
//     planes in rounds of K from one global counter (first ticket of a round).  Every job of a plane therefore runs on the XCD that
//     claimed it -- verified from the hardware register, not assumed from blockIdx -- and the plane's 295 KB half-spectrum intermediate
//     is written and re-read through that XCD's own 4 MiB L2: plain stores, `s_waitcnt vmcnt(0)` (the stores are in the L2), one
//     agent-scope counter increment; consumer: relaxed poll of the counter, then its loads.  No L2 write-back fence per hand-off and
//     no fabric round trip of the intermediate (it reaches HBM once, when the kernel ends).  An XCD without workgroups claims nothing;
//     an XCD with more of them claims more -- placement changes speed, never the result.
//   * self-cleaning state.  The last workgroup to leave zeroes the scheduling words, so a (graph-replayed) launch needs no memset node
//     in front of it; the words are zeroed once when the buffer is allocated.  Every spin is bounded; a time-out raises `err`.
#include "fft2_body.h"
#include "prox_sched.h"
#include <algorithm>

namespace dpir {

size_t prox_fused_sync_words(int P, int K) { const int nr = (P + K - 1) / K + 12; return (size_t)PF_ROUND0 + 8 * (size_t)nr + 2 * (size_t)P; }

namespace {

template <int R, int RJ, int SF, bool PF, int OCC>
__global__ __launch_bounds__(256, OCC) void prox_fused_kernel(const ProxFusedArgs a) {
    constexpr int N = R * RJ, THREADS = 256, SLOTS = THREADS / R, CS = THREADS / R;
    constexpr int RJOBS = N / 2 / SLOTS;                                  // row-pair jobs per plane
    constexpr size_t BODY = std::max(rows_lds_elems<R, RJ, THREADS>(), cols_lds_elems<R, RJ, THREADS, SF>());
    extern __shared__ __attribute__((aligned(16))) float2 sm2[];
    volatile unsigned* ctl = reinterpret_cast<volatile unsigned*>(sm2 + N + BODY);     // [0] base plane of the round, [1] index in the round
    for (int i = threadIdx.x; i < N; i += THREADS) sm2[i] = a.tw[i];
    const int P = a.P, K = a.K, WP = a.WP;
    const int strips = WP / CS;
    const unsigned JP = 2 * RJOBS + strips, per_round = JP * (unsigned)K;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;          // HW_REG_XCC_ID[3:0]
    unsigned* ticket = a.sync + PF_TICKET0 + 16 * xcc;
    unsigned* rbase = a.sync + PF_ROUND0 + xcc * (unsigned)a.nr_max;
    unsigned* rowdone = a.sync + PF_ROUND0 + 8 * a.nr_max;
    unsigned* coldone = rowdone + P;
    const size_t total_rows = (size_t)P * N;
    __syncthreads();
    for (;;) {
        if (threadIdx.x == 0) {
            const unsigned tk = add_relaxed(ticket, 1u);
            const unsigned r = tk / per_round, idx = tk - r * per_round;
            unsigned base = (unsigned)P;
            if (r < (unsigned)a.nr_max) {
                if (idx == 0) {                       // first ticket of a round: claim K planes for this XCD
                    base = add_relaxed(a.sync, (unsigned)K);
                    __hip_atomic_store(rbase + r, base + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    unsigned spins = 0, v;
                    while ((v = ld_relaxed(rbase + r)) == 0u) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > PF_SPIN_LIMIT) { atomicOr(a.err, 2u); v = (unsigned)P + 1u; break; }
                    }
                    base = v - 1u;
                }
            }
            ctl[0] = base; ctl[1] = idx;
        }
        __syncthreads();
        const unsigned base = ctl[0], idx = ctl[1];
        __syncthreads();
        if (base >= (unsigned)P) break;
        const unsigned n_rows = (unsigned)K * RJOBS, n_cols = (unsigned)K * strips;
        if (idx < n_rows) {
            const int plane = (int)(base + idx / RJOBS), j = (int)(idx % RJOBS);
            if (plane < P) {
                rfft_rows_body<R, RJ, THREADS, true>(sm2, (size_t)plane * RJOBS + j, a.x, a.pa, a.pb, a.pm, a.sp, a.hbuf, WP, total_rows, a.tw, a.fu, a.slot_col);
                job_done(rowdone + plane);
            }
        } else if (idx < n_rows + n_cols) {
            const unsigned i2 = idx - n_rows;
            const int plane = (int)(base + i2 / strips), strip = (int)(i2 % strips);
            if (plane < P) {
                cfft_cols_body<R, RJ, SF == 1 ? 2 : 3, THREADS, SF, true, DepWait, PF>(sm2, plane, strip, a.hbuf, a.solve, WP, a.tw, DepWait{rowdone + plane, RJOBS, a.err});
                job_done(coldone + plane);
            }
        } else {
            const unsigned i3 = idx - n_rows - n_cols;
            const int plane = (int)(base + i3 / RJOBS), j = (int)(i3 % RJOBS);
            if (plane < P)
                irfft_rows_body<R, RJ, THREADS, true>(sm2, (size_t)plane * RJOBS + j, a.hbuf, a.out, a.scale, a.oa, a.ob, a.blend_base, a.g, WP, total_rows, a.tw,
                                                      a.rn, a.col_slot, DepWait{coldone + plane, (unsigned)strips, a.err});
        }
        __syncthreads();          // the next job reuses the LDS areas
    }
    // the last workgroup to leave zeroes the scheduling words for the next launch (every other workgroup is past its last access)
    if (threadIdx.x == 0) ctl[0] = add_relaxed(a.sync + 2, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (ctl[0]) {
        const unsigned words = (unsigned)(PF_ROUND0 + 8 * a.nr_max + 2 * P);
        for (unsigned i = threadIdx.x; i < words; i += THREADS) __hip_atomic_store(a.sync + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int R, int RJ, int SF, bool PF, int OCC>
Status launch_V(hipStream_t s, const ProxFusedArgs& a, int cus) {
    constexpr int N = R * RJ, THREADS = 256, CS = THREADS / R, RJOBS = N / 2 / (THREADS / R);
    constexpr size_t BODY = std::max(rows_lds_elems<R, RJ, THREADS>(), cols_lds_elems<R, RJ, THREADS, SF>());
    const size_t lds = (N + BODY) * sizeof(float2) + 16;
    auto fn = prox_fused_kernel<R, RJ, SF, PF, OCC>;
    static LdsAttrOnce attr;
    DPIR_HIP(attr.set(reinterpret_cast<const void*>(fn), 160 * 1024));
    static int occ_cache[16] = {};
    int dev = 0;
    DPIR_HIP(hipGetDevice(&dev));
    int occ = (dev >= 0 && dev < 16) ? occ_cache[dev] : 0;
    if (occ == 0) {
        DPIR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, THREADS, lds));
        occ = std::max(1, std::min(occ, OCC));
        if (const char* ev = getenv("DPIR_PROX_OCC")) occ = std::max(1, std::min(8, atoi(ev)));
        if (dev >= 0 && dev < 16) occ_cache[dev] = occ;
    }
    const int strips = a.WP / CS, JP = 2 * RJOBS + strips;
    // no more workgroups per XCD than one round has jobs: an XCD whose workgroups start first must not claim a second round while the
    // others have not claimed their first
    const long long want = 8ll * a.K * JP;
    const unsigned G = (unsigned)std::max(8ll, std::min((long long)cus * occ, want));
    hipLaunchKernelGGL(fn, dim3(G), dim3(THREADS), lds, s, a);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

// development switch DPIR_PROX_VARIANT = <prefetch 0|1><waves per SIMD 2|3> (N = 256 only; default: see below)
template <int R, int RJ, int SF>
Status launch_T(hipStream_t s, const ProxFusedArgs& a, int cus) {
    if constexpr (RJ > 16) return launch_V<R, RJ, SF, true, 1>(s, a, cus);
    else {
        static const int variant = getenv("DPIR_PROX_VARIANT") ? atoi(getenv("DPIR_PROX_VARIANT")) : 12;
        switch (variant) {
            case 2: return launch_V<R, RJ, SF, false, 2>(s, a, cus);
            case 3: return launch_V<R, RJ, SF, false, 3>(s, a, cus);
            case 13: return launch_V<R, RJ, SF, true, 3>(s, a, cus);
            default: return launch_V<R, RJ, SF, true, 2>(s, a, cus);
        }
    }
}

}  // namespace

bool prox_fused_supported(int H, int W, int sf) { return H == W && (H == 256 || H == 512) && (sf == 1 || sf == 2 || sf == 4); }
int prox_fused_round(int P) { return std::max(1, std::min(6, (P + 7) / 8)); }

Status launch_prox_fused(hipStream_t s, const ProxFusedArgs& a, int N, int sf, int cus) {
    if (!prox_fused_supported(N, N, sf)) return invalid("prox_fused: unsupported size");
    if (!a.sync || !a.err || a.K < 1 || a.nr_max != (a.P + a.K - 1) / a.K + 12) return invalid("prox_fused: bad scheduling state");
    if (sf > 1 && (!a.solve.invW || !a.solve.slot_col || a.solve.sf != sf)) return invalid("prox_fused: bad sf > 1 arguments");
    if (N == 256) {
        if (sf == 1) return launch_T<16, 16, 1>(s, a, cus);
        return sf == 2 ? launch_T<16, 16, 2>(s, a, cus) : launch_T<16, 16, 4>(s, a, cus);
    }
    if (sf == 1) return launch_T<16, 32, 1>(s, a, cus);
    return sf == 2 ? launch_T<16, 32, 2>(s, a, cus) : launch_T<16, 32, 4>(s, a, cus);
}

}  // namespace dpir
