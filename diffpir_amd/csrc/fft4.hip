// fft4: the half-spectrum FFT prox at N = 256 (utils/utils_sisr.py:9-19, 65-95; the headline deblurring / SR configurations) with ONE WAVE PER
// TRANSFORM (fft4_wave.h) and a COLUMN-MAJOR half spectrum.  Same mathematics as fft2.hip (row-pair packing, half spectrum, closed-form solve between
// the column transforms), different mapping to the machine:
//   * rows    : a wave takes one pair of real rows: float4 loads, wave-private LDS re-distribution, 256-point FFT in the wave, Hermitian un-packing,
//               and writes {A[k], B[k]} of its two rows as ONE 16-byte store into the column-major spectrum [plane][slot][row] (rows r, r + 1 adjacent);
//   * columns : a wave takes one column (2 KB contiguous): forward FFT -> closed-form solve on registers whose operands (FBFy, FB, F2B, stored
//               column-major too) are 512-byte contiguous wave loads -> inverse FFT -> store.  No workgroup barrier, no strips, no padding columns
//               (129 stored columns instead of 144);
//   * rows inv: a wave gathers {A, B}[k] of its row pair (16-byte loads), Hermitian re-packing through the LDS tile, inverse FFT, float4 epilogue
//               (x*2-1, guidance blend, or the fused re-noise with Philox draws per float4 group exactly as fft2's).
// A batch-16 apply is 6 waves per SIMD in every pass (fft2: 1.5), each of them ~60 VGPRs.
// sf > 1 keeps fft2's alias-grouped SLOT order (slot sf q + b = alias b of fold group q): a workgroup's four waves are four consecutive slots, i.e.
// one fold group at sf = 4 and two at sf = 2; the row aliases u + a N/sf of a column live in ONE lane (registers j, since N/sf is a multiple of 64).
#include "common.h"
#include "elem.h"
#include "philox.h"
#include "fft4_wave.h"
#include <vector>

namespace dpir {
namespace {

constexpr int N4 = 256, WAVES = 4, THREADS4 = 64 * WAVES;
constexpr int WLDS = 320;        // float2 per wave: the 16 x 18 transpose tile; also 256 natural-order complex values or 2 x 256 staged floats
// row passes: a workgroup = RW waves = RW row pairs = 16 consecutive rows, so that the column-major spectrum is written / read in FULL 128-byte lines
// (16 rows x 8 bytes of one slot) through an LDS tile [slot][RW + 1] of 16-byte {A, B} entries (one wave = one entry per slot; + 1: bank spread)
constexpr int RW = 8, RTHREADS = 64 * RW, TST = RW + 1;

__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------ rows forward
// grid: pairs / 8 workgroups x 512 threads; wave = one row pair.  NC = stored columns (slots) per plane; slot_col (sf > 1): slot -> column | mirror << 16, -1 padding.
__global__ __launch_bounds__(RTHREADS) void rfft4_rows_kernel(const float* x, float pa, float pb, float pm, const StepDev* sp, float2* out, int NC,
                                                             const float2* tw, RowsFuse fu, const int* slot_col) {
    extern __shared__ __attribute__((aligned(16))) float2 sm4[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t pair = (size_t)blockIdx.x * RW + wave;
    float2* lds = sm4 + wave * WLDS;
    float4* tile = reinterpret_cast<float4*>(sm4 + RW * WLDS);         // [NC][TST]
    const WaveTw w = wave_tw_load(tw, lane);
    if (sp) pm = sp->tau;
    const size_t ra = 2 * pair;
    const size_t plane = ra / N4; const int r = (int)(ra - plane * N4);
    float4 qa = *reinterpret_cast<const float4*>(x + ra * N4 + lane * 4);
    float4 qb = *reinterpret_cast<const float4*>(x + (ra + 1) * N4 + lane * 4);
    if (fu.eps6) {
#pragma clang fp contract(off)
        const size_t n = plane / 3, c = plane - n * 3;
        const float* ep = fu.eps6 + ((n * fu.out_ch + c) * N4 + r) * N4 + lane * 4;
        const float4 ea = *reinterpret_cast<const float4*>(ep), eb = *reinterpret_cast<const float4*>(ep + N4);
        const float c1 = sp->c1, c2 = sp->c2;
        qa.x = fminf(fmaxf(c1 * qa.x - c2 * ea.x, -1.0f), 1.0f); qa.y = fminf(fmaxf(c1 * qa.y - c2 * ea.y, -1.0f), 1.0f);
        qa.z = fminf(fmaxf(c1 * qa.z - c2 * ea.z, -1.0f), 1.0f); qa.w = fminf(fmaxf(c1 * qa.w - c2 * ea.w, -1.0f), 1.0f);
        qb.x = fminf(fmaxf(c1 * qb.x - c2 * eb.x, -1.0f), 1.0f); qb.y = fminf(fmaxf(c1 * qb.y - c2 * eb.y, -1.0f), 1.0f);
        qb.z = fminf(fmaxf(c1 * qb.z - c2 * eb.z, -1.0f), 1.0f); qb.w = fminf(fmaxf(c1 * qb.w - c2 * eb.w, -1.0f), 1.0f);
    }
    float* st = reinterpret_cast<float*>(lds);                        // [2][256] floats
    *reinterpret_cast<float4*>(st + lane * 4) = qa;
    *reinterpret_cast<float4*>(st + N4 + lane * 4) = qb;
    wsync();
    float2 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float a = (st[lane + 64 * j] * pa + pb) * pm, b = (st[N4 + lane + 64 * j] * pa + pb) * pm;
        v[j] = make_float2(a, b);
    }
    wsync();
    wave_fft256<false>(v, w, lds, lane);
    wsync();
#pragma unroll
    for (int j = 0; j < 4; ++j) lds[lane + 64 * j] = v[j];
    wsync();
    // un-pack: A[k] = (Z[k] + conj(Z[N-k]))/2, B[k] = (Z[k] - conj(Z[N-k]))/(2i), k = 0..N/2, into this wave's entry of every slot's tile row
    if (!slot_col) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = lane + 64 * j;
            if (k > N4 / 2) break;
            float2 zk = v[j], zn = lds[(N4 - k) & (N4 - 1)];
            zn.y = -zn.y;
            const float2 d = csub(zk, zn);
            tile[k * TST + wave] = make_float4(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y), 0.5f * d.y, -0.5f * d.x);
        }
    } else {
        for (int s = lane; s < NC; s += 64) {
            const int cm = slot_col[s];
            float4 ab = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cm >= 0) {
                const int k = cm & 0xffff;
                float2 zk = lds[k], zn = lds[(N4 - k) & (N4 - 1)];
                zn.y = -zn.y;
                const float2 d = csub(zk, zn);
                ab = make_float4(0.5f * (zk.x + zn.x), 0.5f * (zk.y + zn.y), 0.5f * d.y, -0.5f * d.x);
            }
            tile[s * TST + wave] = ab;
        }
    }
    __syncthreads();
    // the workgroup's 16 rows of every slot: one full 128-byte line per slot, eight lanes per line
    const size_t ra0 = (size_t)blockIdx.x * RW * 2;
    const size_t plane0 = ra0 / N4; const int r0 = (int)(ra0 - plane0 * N4);
    float4* o4 = reinterpret_cast<float4*>(out + (plane0 * NC) * N4 + r0);
    for (int i = threadIdx.x; i < NC * RW; i += RTHREADS) {
        const int s = i >> 3, c = i & 7;
        o4[(size_t)s * (N4 / 2) + c] = tile[s * TST + c];
    }
}

// ------------------------------------------------------------------------------------------------ rows inverse
__global__ __launch_bounds__(RTHREADS) void irfft4_rows_kernel(const float2* in, float* out, float scale, float oa, float ob, const float* blend_base, float g,
                                                              int NC, const float2* tw, RenoiseFuse rn, const int* col_slot) {
    extern __shared__ __attribute__((aligned(16))) float2 sm4[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t pair = (size_t)blockIdx.x * RW + wave;
    float2* lds = sm4 + wave * WLDS;
    float4* tile = reinterpret_cast<float4*>(sm4 + RW * WLDS);         // [N/2 + 1 columns][TST]
    const size_t ra = 2 * pair;
    const size_t plane = ra / N4; const int r = (int)(ra - plane * N4);
    // what the epilogue combines with the transform (x_t for the fused re-noise, or the blend base) does not depend on it: requested first
    const float* pre_src = rn.xt ? rn.xt : blend_base;
    const size_t ga = ra * N4 + lane * 4, gb = ga + N4;
    float4 pre_a = make_float4(0.f, 0.f, 0.f, 0.f), pre_b = pre_a;
    if (pre_src) { pre_a = *reinterpret_cast<const float4*>(pre_src + ga); pre_b = *reinterpret_cast<const float4*>(pre_src + gb); }
    {   // the workgroup's 16 rows of columns 0..N/2: one full 128-byte line per column, eight lanes per line
        const float4* i4 = reinterpret_cast<const float4*>(in + (plane * NC) * N4 + (r - 2 * wave));
        for (int i = threadIdx.x; i < (N4 / 2 + 1) * RW; i += RTHREADS) {
            const int k = i >> 3, c = i & 7;
            const int s = col_slot ? col_slot[k] : k;
            tile[k * TST + c] = i4[(size_t)s * (N4 / 2) + c];
        }
    }
    const WaveTw w = wave_tw_load(tw, lane);
    __syncthreads();
    float4 ab[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = lane + 64 * j;
        ab[j] = k <= N4 / 2 ? tile[k * TST + wave] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // Hermitian re-packing: Z[k] = A[k] + i B[k], Z[N-k] = conj(A[k]) + i conj(B[k])
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = lane + 64 * j;
        if (k > N4 / 2) break;
        const float4 q = ab[j];
        lds[k] = make_float2(q.x - q.w, q.y + q.z);
        if (k > 0 && k < N4 / 2) lds[N4 - k] = make_float2(q.x + q.w, -q.y + q.z);
    }
    wsync();
    float2 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = lds[lane + 64 * j];
    wsync();
    wave_fft256<true>(v, w, lds, lane);
    wsync();
    float* st = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        st[lane + 64 * j] = (v[j].x * scale) * oa + ob;
        st[N4 + lane + 64 * j] = (v[j].y * scale) * oa + ob;
    }
    wsync();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float4 q = *reinterpret_cast<const float4*>(st + h * N4 + lane * 4);
        const size_t gi = h ? gb : ga;
        const float4 pre = h ? pre_b : pre_a;
        if (blend_base) {
            const float4 b0 = rn.xt ? *reinterpret_cast<const float4*>(blend_base + gi) : pre;
            q.x = b0.x + g * (q.x - b0.x); q.y = b0.y + g * (q.y - b0.y); q.z = b0.z + g * (q.z - b0.z); q.w = b0.w + g * (q.w - b0.w);
        }
        if (rn.xt) {
#pragma clang fp contract(off)
            const StepDev sd = *rn.sp;
            const size_t per_image = (size_t)3 * N4 * N4;
            const size_t n = gi / per_image, e = gi - n * per_image;
            float z1[4] = {0.f, 0.f, 0.f, 0.f}, z2[4];
            if (rn.n2) {                         // host-fed noise: this batch's tensors, step i
                const float4 t2 = *reinterpret_cast<const float4*>((rn.lp ? rn.lp->n2 : rn.n2) + (size_t)sd.i * rn.stride + gi);
                z2[0] = t2.x; z2[1] = t2.y; z2[2] = t2.z; z2[3] = t2.w;
                if (rn.with_n1) {
                    const float4 t1 = *reinterpret_cast<const float4*>((rn.lp ? rn.lp->n1 : rn.n1) + (size_t)sd.i * rn.stride + gi);
                    z1[0] = t1.x; z1[1] = t1.y; z1[2] = t1.z; z1[3] = t1.w;
                }
            } else {
                const uint64_t img = (uint64_t)(rn.lp->image_offset + (long long)n);
                philox_normal4(rn.lp->seed, 2 + 4 * (uint64_t)sd.i, img, e >> 2, z2);
                if (rn.with_n1) philox_normal4(rn.lp->seed, 1 + 4 * (uint64_t)sd.i, img, e >> 2, z1);
            }
            const float xv[4] = {pre.x, pre.y, pre.z, pre.w}, av[4] = {q.x, q.y, q.z, q.w};
            float rv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a = av[u];
                const float eps = (xv[u] - sd.sa_t * a) / sd.s1m_t;
                float inner = sd.q * eps;
                if (rn.with_n1) inner = inner + sd.es * z1[u];
                float vv = sd.sa_p * a + sd.k1 * inner;
                vv = vv + sd.k2 * z2[u];
                rv[u] = vv;
            }
            *reinterpret_cast<float4*>(rn.xt + gi) = make_float4(rv[0], rv[1], rv[2], rv[3]);
            continue;
        }
        *reinterpret_cast<float4*>(out + gi) = q;
    }
}

// ------------------------------------------------------------------------------------------------ columns
// wave = one stored column (slot) of one plane.  MODE 0: forward only; MODE 2 (sf = 1): forward -> FX = (FR - conj(FB) (FB FR)/(F2B + alpha)) / alpha,
// FR = FBFy + F(alpha x) -> inverse; MODE 3 (sf > 1): the same with FB FR and F2B averaged over the sf x sf aliases (utils_sisr.py:65-75 `splits` + mean).
// grid: P * ceil(NC / 4) workgroups: the four waves of a workgroup are four consecutive slots of ONE plane.
template <int MODE, int SF>
__global__ __launch_bounds__(THREADS4) void cfft4_cols_kernel(float2* buf, SolveArgs a, int NC, const float2* tw) {
    extern __shared__ __attribute__((aligned(16))) float2 sm4[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = (NC + WAVES - 1) / WAVES;
    const int plane = blockIdx.x / groups;
    const int s = (blockIdx.x - plane * groups) * WAVES + wave;
    const bool live = s < NC;                                          // MODE 3 has a workgroup barrier: dead waves stay until it
    if (MODE != 3 && !live) return;
    float2* lds = sm4 + wave * WLDS;
    float2* base = buf + ((size_t)plane * NC + (live ? s : 0)) * N4;
    float2 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = live ? base[lane + 64 * j] : make_float2(0.f, 0.f);
    const WaveTw w = wave_tw_load(tw, lane);
    wave_fft256<false>(v, w, lds, lane);
    if (MODE == 2) {
        const float alpha = a.sp ? a.sp->tau : a.alpha;
        const int n_img = plane / 3;
        const float2* FB = a.FB + ((size_t)n_img * NC + s) * N4;
        const float* F2B = a.F2B + ((size_t)n_img * NC + s) * N4;
        const float2* FBFy = a.FBFy + ((size_t)plane * NC + s) * N4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = lane + 64 * j;
            const float2 fr = cadd(FBFy[u], v[j]);
            const float2 fb = FB[u];
            const float2 x1 = cmul2(fb, fr);
            const float den = F2B[u] + alpha;
            const float2 q = make_float2(x1.x / den, x1.y / den);
            const float2 tq = cmulc2(q, fb);                          // conj(FB) * q
            v[j] = make_float2((fr.x - tq.x) / alpha, (fr.y - tq.y) / alpha);
        }
        wsync();
        wave_fft256<true>(v, w, lds, lane);
    }
    if (MODE == 3) {
        // slot s = sf q + b: alias b of fold group q.  Row aliases u + a Hs (Hs = N / sf, a multiple of 64) are registers of ONE lane.
        const float alpha = a.sp ? a.sp->tau : a.alpha;
        constexpr int Hs = N4 / SF, KH = 4 / SF;                        // KH folded values per lane (sf 4: 1, sf 2: 2)
        const int QW = N4 / SF / 2 + 1;
        const int n_img = plane / 3;
        float2* fold = sm4 + WAVES * WLDS;                              // [WAVES][Hs] row-folded FB * FR of each slot of the workgroup
        const float2* FB = a.FB + ((size_t)n_img * NC + (live ? s : 0)) * N4;
        const float2* FBFy = a.FBFy + ((size_t)plane * NC + (live ? s : 0)) * N4;
        float2 fb[4], sacc[KH];
#pragma unroll
        for (int i = 0; i < KH; ++i) sacc[i] = make_float2(0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int u = lane + 64 * j;
            fb[j] = live ? FB[u] : make_float2(0.f, 0.f);
            v[j] = cadd(live ? FBFy[u] : make_float2(0.f, 0.f), v[j]);
            sacc[j % KH] = cadd(sacc[j % KH], cmul2(fb[j], v[j]));
        }
#pragma unroll
        for (int i = 0; i < KH; ++i) fold[wave * Hs + lane + 64 * i] = sacc[i];
        __syncthreads();
        // R[p] of this slot's fold group: sum over the group's sf slots (mirrored aliases: conj of the mirrored row), / (sf^2 (invW + alpha))
        const int cmine = live ? a.slot_col[s] : -1;
        const bool mir = cmine >= 0 && (cmine >> 16);
        const int q = s / SF, w0 = (wave / SF) * SF;                    // first wave (slot) of my fold group inside the workgroup
        const float inv_n = 1.0f / (float)(SF * SF);
        float2 R[KH];
#pragma unroll
        for (int i = 0; i < KH; ++i) {
            const int p0 = lane + 64 * i;                               // the row (mod Hs) this lane needs R for ...
            const int p = mir ? (Hs - p0) % Hs : p0;                    // ... which for a mirrored slot is R[(Hs - p) % Hs] conjugated
            const int pm = (Hs - p) % Hs;
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int b = 0; b < SF; ++b) {
                const int sb = q * SF + b;
                const int cm = sb < NC ? a.slot_col[sb] : -1;
                if (cm < 0) continue;
                if (cm >> 16) { const float2 z = fold[(w0 + b) * Hs + pm]; acc.x += z.x; acc.y -= z.y; }
                else acc = cadd(acc, fold[(w0 + b) * Hs + p]);
            }
            float2 rr = make_float2(0.f, 0.f);
            if (live && q < QW) {
                const float den = a.invW[((size_t)n_img * Hs + p) * QW + q] + alpha;
                rr = make_float2(acc.x * inv_n / den, acc.y * inv_n / den);
            }
            if (mir) rr.y = -rr.y;
            R[i] = rr;
        }
        if (!live) return;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 tq = cmulc2(R[j % KH], fb[j]);                 // conj(FB) * R~
            v[j] = make_float2((v[j].x - tq.x) / alpha, (v[j].y - tq.y) / alpha);
        }
        wsync();
        wave_fft256<true>(v, w, lds, lane);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) base[lane + 64 * j] = v[j];
}

// invW[n, p, q] = mean over the sf x sf aliases of F2B (utils_sisr.py:71), column-major slots
__global__ void fold_f2b4_kernel(const float* F2B, const int* slot_col, int NC, int sf, float* invW, size_t total) {
    const int Hs = N4 / sf, QW = N4 / sf / 2 + 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int q = (int)(i % QW);
        const int p = (int)((i / QW) % Hs);
        const size_t n = i / ((size_t)QW * Hs);
        const float* pl = F2B + n * (size_t)NC * N4;
        float acc = 0.f;
        for (int b = 0; b < sf; ++b) {
            const int slot = sf * q + b;
            const int cm = slot < NC ? slot_col[slot] : -1;
            if (cm < 0) continue;
            const int base_row = (cm >> 16) ? (Hs - p) % Hs : p;          // |FB|^2 is real: the mirrored alias is just the mirrored row
            for (int a = 0; a < sf; ++a) acc += pl[(size_t)slot * N4 + base_row + a * Hs];
        }
        invW[i] = acc / (float)(sf * sf);
    }
}

}  // namespace

bool fft4_supported(int H, int W, int sf) { return H == 256 && W == 256 && (sf == 1 || sf == 2 || sf == 4); }
// stored columns (slots) per plane: W/2 + 1 for sf = 1, sf * (W/sf/2 + 1) alias-grouped slots otherwise
int fft4_columns(int W, int sf) { return sf == 1 ? W / 2 + 1 : sf * (W / sf / 2 + 1); }
// slot -> (column | mirrored << 16) or -1, column -> canonical slot: fft2's alias grouping without strip padding
void fft4_build_map(int N, int sf, std::vector<int>& slot_col, std::vector<int>& col_slot) {
    const int NC = fft4_columns(N, sf), Ws = N / sf;
    slot_col.assign(NC, -1);
    col_slot.assign(N / 2 + 1, -1);
    for (int q = 0; q <= Ws / 2; ++q)
        for (int b = 0; b < sf; ++b) {
            const int c = q + b * Ws;
            if (c >= N) continue;
            const int col = c <= N / 2 ? c : N - c, mir = c <= N / 2 ? 0 : 1;
            slot_col[sf * q + b] = col | (mir << 16);
            if (col_slot[col] < 0 || (!mir && (slot_col[col_slot[col]] >> 16))) col_slot[col] = sf * q + b;      // prefer the direct copy
        }
}

static size_t lds4(bool fold, int sf) { return ((size_t)WAVES * WLDS + (fold ? (size_t)WAVES * (N4 / sf) : 0)) * sizeof(float2); }
static size_t lds4_rows(int NC) { return (size_t)RW * WLDS * sizeof(float2) + (size_t)NC * TST * sizeof(float4); }

Status launch_rfft4_rows(hipStream_t s, const float2* tw, const float* x, float pa, float pb, float pm, const StepDev* sp, float2* out, int P, int NC,
                         const float* eps6, int out_ch, const int* slot_col) {
    if (eps6 && !sp) return invalid("rfft4_rows: the fused x0 prologue reads its coefficients from the device step block");
    const size_t pairs = (size_t)P * N4 / 2;            // a multiple of RW: no partial workgroup
    hipLaunchKernelGGL(rfft4_rows_kernel, dim3((unsigned)(pairs / RW)), dim3(RTHREADS), lds4_rows(NC), s, x, pa, pb, pm, sp, out, NC, tw,
                       RowsFuse{eps6, out_ch}, slot_col);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_irfft4_rows(hipStream_t s, const float2* tw, const float2* in, float* out, float scale, float oa, float ob, const float* blend, float g, int P,
                          int NC, const RenoiseArgs* ra, const int* col_slot) {
    RenoiseFuse rn{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (ra) rn = RenoiseFuse{ra->xt, ra->sp, ra->lp, ra->n1, ra->n2, ra->stride, ra->with_n1};
    const size_t pairs = (size_t)P * N4 / 2;
    hipLaunchKernelGGL(irfft4_rows_kernel, dim3((unsigned)(pairs / RW)), dim3(RTHREADS), lds4_rows(N4 / 2 + 1), s, in, out, scale, oa, ob, blend, g, NC,
                       tw, rn, col_slot);
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_cfft4_cols(hipStream_t s, const float2* tw, float2* buf, const SolveArgs& a, bool solve, int P, int NC) {
    const unsigned grid = (unsigned)(P * ((NC + WAVES - 1) / WAVES));
    if (!solve) hipLaunchKernelGGL((cfft4_cols_kernel<0, 1>), dim3(grid), dim3(THREADS4), lds4(false, 1), s, buf, a, NC, tw);
    else if (a.sf == 1) hipLaunchKernelGGL((cfft4_cols_kernel<2, 1>), dim3(grid), dim3(THREADS4), lds4(false, 1), s, buf, a, NC, tw);
    else {
        if ((a.sf != 2 && a.sf != 4) || !a.invW || !a.slot_col || NC % a.sf) return invalid("cfft4_cols: bad sf > 1 arguments");
        if (a.sf == 2) hipLaunchKernelGGL((cfft4_cols_kernel<3, 2>), dim3(grid), dim3(THREADS4), lds4(true, 2), s, buf, a, NC, tw);
        else hipLaunchKernelGGL((cfft4_cols_kernel<3, 4>), dim3(grid), dim3(THREADS4), lds4(true, 4), s, buf, a, NC, tw);
    }
    DPIR_HIP(hipGetLastError());
    return Status{};
}
Status launch_fold_f2b4(hipStream_t s, const float* F2B, const int* slot_col, int NC, int sf, float* invW, int B) {
    const size_t total = (size_t)B * (N4 / sf) * (N4 / sf / 2 + 1);
    hipLaunchKernelGGL(fold_f2b4_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, F2B, slot_col, NC, sf, invW, total);
    DPIR_HIP(hipGetLastError());
    return Status{};
}

}  // namespace dpir
