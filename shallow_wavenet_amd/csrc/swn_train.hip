// Backward pass of the teacher-forced stack (fp32) - the GPU side of the reference's
// loss.backward() through CSWNV.forward / DSWNV.forward (train_cswnv...py:724-874).
//
// Gradients are produced in the PACKED parameter layout (csrc/swn_geom.hpp) into a zeroed buffer;
// the host unfolds them onto the reference's nn.Parameters (nets/_autograd.py).  Two generic fp32
// kernels do all contractions:
//   time_gemm_kernel    Y[b][m][t] (op)= sum_{tap,c} A(m,tap,c) * X[b][c][t + sgn*(tap-center)*dil]
//                       (conv forward / data-gradient of a dilated conv / plain 1x1, zero outside [0,T))
//   reduce_gemm_kernel  G[m][tap][c] += sum_{b,t} P[b][m][t] * Q[b][c][t + sgn*(tap-center)*dil]
//                       (weight gradients; split over time blocks, float atomics)
// plus element-wise kernels for the gate, the rank-1 upsampler / hoisted in_x, the input layer and
// the Laplace head.  Hidden states h_0..h_L saved by the forward are reused; the gate pre-activations
// are recomputed (one extra conv GEMM per layer) instead of being stored.
//
// Mixed-precision mode (precision = SWN_PRECISION_BF16): the same chain with bf16 operands on the matrix cores -
//   time_gemm_bf16t_kernel / reduce_gemm_bf16s_kernel   128 x 128 tiles, fp32 operands rounded on the way into LDS
//   time_gemm_b16_kernel, reduce_gemm_bf16s_kernel<true, true>   the layer GEMMs of the GEMM-stack geometries with BOTH
//       operands read as bf16: gate_bwd_kernel leaves da and the layer input as bf16 rows (two copies each, the second moved
//       right by one position so that every tap shift lands on an aligned element), wd_t16_kernel the transposed matrices;
//       what the smaller operands buy is a third workgroup per CU
// and the pre-activations are read back where the forward of the same mode kept them (swn_backward_keep).  The BL6 class takes
// the fused per-layer backward of csrc/swn_bwd_bl6.hip instead.
#include <hip/hip_runtime.h>
#include <atomic>
#include "swn_geom.hpp"
#include "swn_mma.hpp"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

constexpr int SWN_WUP_COPIES = 16;      // see wup_fold_kernel

struct TimeGemm {
    const float* A; long a_sm, a_stap, a_sc;        // A(m, tap, c)
    const float* X; long x_sb, x_sc, x_st;          // X[b][c][t]
    float* Y; long y_sb, y_sm;                      // Y[b][m][t], t contiguous
    const float* mask; long k_sb, k_sm;             // optional relu mask (same indexing as Y): keep where mask > 0
    int M, taps, KC, T, sgn, center, dil, accumulate;
    // optional (appended, zero = off): multiplicative masks for dropout, and a separate valid length of X
    const float* xmul; long xm_sb, xm_sc;           // X[b][c][t] is read as X * xmul[b][c][t]
    const float* ymul; long ym_sb, ym_sm;           // the result is scaled by ymul[b][m][t] before it is stored / added
    int XT;                                         // X is valid on [0, XT) (0: same as T)
    int nb;                                         // batch size (set by the launcher of the XCD-ordered kernel)
    const float* bias; int relu;                    // optional: v = acc + bias[m], then max(v, 0)  (forward 1x1 layers)
    int ksplit;                                     // > 1 (bf16 64x64 kernel only): the k range is cut into ksplit parts, one per
                                                    // workgroup, added atomically into a zeroed Y (frame-rate GEMMs: few tiles, long k)
    int y_g4;                                       // time_gemm_b16_kernel only: Y (B, M, T) is stored in the G4 layout (swn_geom.hpp), plain store
    // optional bf16 operands (time_gemm_b16_kernel: data gradients of the mixed-precision chain)
    const unsigned short* A16; long a16_sm, a16_stap;   // A(m, tap, c) = A16[tap * a16_stap + m * a16_sm + c]  (c contiguous)
    const unsigned short* X16; long x16_sb, x16_sc, x16_odd;   // X[b][c][t] = X16[b * x16_sb + c * x16_sc + t], and x16_odd elements on a
                                                    // second copy moved right by one position (copy[u] = X[u - 1]): a pair
                                                    // (t, t + 1) shifted by any tap is then one aligned dword of one of the two
};

// Epilogue of the time contractions: relu mask, dropout multiplier, accumulate, store - for NV results of one thread.
// The reads of one kind are issued together through a buffer resource (invalid rows = the out-of-range offset) BEFORE
// anything is stored: written as a per-element read-modify-write the compiler must keep every load behind the previous
// store (Y may alias), i.e. one global round trip per element.
template <int NV>
__device__ __forceinline__ void tg_epilogue(const TimeGemm& g, const int b, const int (&m)[NV], const int t, float (&v)[NV]) {
    bool ok[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) ok[i] = m[i] < g.M && t < g.T;
    if (g.bias) {
#pragma unroll
        for (int i = 0; i < NV; ++i) { v[i] += g.bias[ok[i] ? m[i] : 0]; if (g.relu) v[i] = fmaxf(v[i], 0.f); }
    }
    if (g.mask) {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(g.mask + (size_t)b * g.k_sb);
        float k[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) k[i] = bld1(r, ok[i] ? (unsigned)((m[i] * g.k_sm + t) * 4) : SWN_OOB);
#pragma unroll
        for (int i = 0; i < NV; ++i) if (!(k[i] > 0.f)) v[i] = 0.f;
    }
    if (g.ymul) {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(g.ymul + (size_t)b * g.ym_sb);
        float k[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) k[i] = bld1(r, ok[i] ? (unsigned)((m[i] * g.ym_sm + t) * 4) : SWN_OOB);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] *= k[i];
    }
    if (g.accumulate) {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(g.Y + (size_t)b * g.y_sb);
        float k[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) k[i] = bld1(r, ok[i] ? (unsigned)((m[i] * g.y_sm + t) * 4) : SWN_OOB);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] += k[i];
    }
    float* Y = g.Y + (size_t)b * g.y_sb + t;
#pragma unroll
    for (int i = 0; i < NV; ++i) if (ok[i]) Y[(size_t)m[i] * g.y_sm] = v[i];
}

// 64(m) x 64(t) tile, BK = 16 over k = (tap, c); product on the matrix cores (swn_mma.hpp); NST k-tiles of operand
// loads in flight per thread.
template <bool XMUL>
__global__ __launch_bounds__(256) void time_gemm_kernel(const TimeGemm g) {
    __shared__ float As[16][SWN_MMA_PITCH];
    __shared__ float Bs[16][SWN_MMA_PITCH];
    // XCD-aware order, as in time_gemm_bf16t_kernel: 1-D grid, each XCD walks a contiguous range of time tiles
    const int ntt = (g.T + 63) / 64, mtl = (g.M + 63) / 64;
    const int chunk = (ntt * g.nb + 7) / 8;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int ttl = idx / mtl, mtile = idx - ttl * mtl;
    const int gt = xcd * chunk + ttl;
    if (ttl >= chunk || gt >= ntt * g.nb) return;
    const int b = gt / ntt, t0 = (gt - b * ntt) * 64, m0 = mtile * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const __amdgpu_buffer_rsrc_t rA = rsrc_of(g.A), rX = rsrc_of(g.X + (size_t)b * g.x_sb);
    const __amdgpu_buffer_rsrc_t rM = rsrc_of(XMUL ? g.xmul + (size_t)b * g.xm_sb : g.A);
    const int Kd = g.taps * g.KC;
    swn_f32x4 acc[4] = {};
    // (tap, c) of the k index each thread loads, advanced by 16 per k-tile without divisions:
    // A tile: thread owns k-row kk = tid & 15 for rows (tid >> 4) + 16 i ; B tile: k-rows (tid >> 6) + 4 i, column tid & 63
    const int kka = tid & 15, tt = tid & 63;
    int tapA = kka / g.KC, cA = kka - tapA * g.KC;
    int tapB[4], cB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int kk = (tid >> 6) + 4 * i; tapB[i] = kk / g.KC; cB[i] = kk - tapB[i] * g.KC; }
    const int XT = g.XT ? g.XT : g.T;
    const bool tok = t0 + tt < g.T;
    constexpr int NST = 3;
    float ra[NST][4], rb[NST][4], rm[NST][4];
    auto fetch = [&](int k0, float (&qa)[4], float (&qb)[4], float (&qm)[4]) {
        const bool ka = k0 + kka < Kd;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + (tid >> 4) + 16 * i;
            qa[i] = bld1(rA, (ka && m < g.M) ? (unsigned)((m * g.a_sm + tapA * g.a_stap + cA * g.a_sc) * 4) : SWN_OOB);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ts = t0 + tt + g.sgn * (tapB[i] - g.center) * g.dil;
            const bool ok = k0 + (tid >> 6) + 4 * i < Kd && ts >= 0 && ts < XT && tok;
            qb[i] = bld1(rX, ok ? (unsigned)((cB[i] * g.x_sc + ts * g.x_st) * 4) : SWN_OOB);
            if (XMUL) qm[i] = bld1(rM, ok ? (unsigned)(((size_t)cB[i] * g.xm_sc + ts) * 4) : SWN_OOB);
        }
        cA += 16; while (cA >= g.KC) { cA -= g.KC; ++tapA; }
#pragma unroll
        for (int i = 0; i < 4; ++i) { cB[i] += 16; while (cB[i] >= g.KC) { cB[i] -= g.KC; ++tapB[i]; } }
    };
#pragma unroll
    for (int u = 0; u < NST; ++u) fetch(16 * u, ra[u], rb[u], rm[u]);
    for (int k0 = 0; k0 < Kd; k0 += 16 * NST) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {                 // tiles past Kd hold zeros: no branch inside the loop
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                As[kka][(tid >> 4) + 16 * i] = ra[u][i];
                Bs[(tid >> 6) + 4 * i][tt] = XMUL ? rb[u][i] * rm[u][i] : rb[u][i];
            }
            __syncthreads();
            fetch(k0 + 16 * (u + NST), ra[u], rb[u], rm[u]);
            swn_mma_64x64x16(As, Bs, acc, lane, w);
            __syncthreads();
        }
    }
    const int t = t0 + swn_mma_col(lane, w);
    int mrow[16]; float v[16];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) { mrow[4 * mt + i] = m0 + swn_mma_row(lane, mt, i); v[4 * mt + i] = acc[mt][i]; }
    tg_epilogue<16>(g, b, mrow, t, v);
}

// bf16-operand twin of time_gemm_kernel (precision = SWN_PRECISION_BF16): same operands in HBM (fp32), rounded to bf16 on
// their way into LDS, k-tiles of 32, fp32 accumulators and the same epilogue.  A thread stages k-PAIRS (one packed
// LDS word each); the (tap, c) of every k index advances without divisions.
template <bool XMUL>
__global__ __launch_bounds__(256) void time_gemm_bf16_kernel(const TimeGemm g) {
    __shared__ __attribute__((aligned(16))) unsigned As[64][SWN_MMB_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned Bs[64][SWN_MMB_PITCH];
    const int nsp = g.ksplit > 1 ? g.ksplit : 1;
    const int b = blockIdx.z / nsp, ksp = blockIdx.z - b * nsp, t0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const __amdgpu_buffer_rsrc_t rA = rsrc_of(g.A), rX = rsrc_of(g.X + (size_t)b * g.x_sb);
    const __amdgpu_buffer_rsrc_t rM = rsrc_of(XMUL ? g.xmul + (size_t)b * g.xm_sb : g.A);
    const int Kall = g.taps * g.KC;
    const int per = ((Kall + 31) / 32 + nsp - 1) / nsp * 32;      // k per part, whole k-tiles
    const int kbeg = ksp * per, Kd = kbeg + per < Kall ? kbeg + per : Kall;
    swn_f32x4 acc[4] = {};
    const int kp = tid & 15, tt = tid & 63, kq = tid >> 6;
    const int dtap = 32 / g.KC, dc = 32 - dtap * g.KC;          // a k-tile step in (tap, c) coordinates
    int tapA[2], cA[2], tapB[4][2], cB[4][2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int kk = kbeg + 2 * kp + e; tapA[e] = kk / g.KC; cA[e] = kk - tapA[e] * g.KC;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int kb = kbeg + 2 * (kq + 4 * i) + e; tapB[i][e] = kb / g.KC; cB[i][e] = kb - tapB[i][e] * g.KC; }
    }
    unsigned arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int m = m0 + (tid >> 4) + 16 * i; arow[i] = m < g.M ? (unsigned)(m * g.a_sm * 4) : SWN_OOB; }
    const int XT = g.XT ? g.XT : g.T;
    const bool tok = t0 + tt < g.T;
    constexpr int NST = 3;
    float ra[NST][8], rb[NST][8], rm[NST][8];
    auto fetch = [&](int k0, float (&qa)[8], float (&qb)[8], float (&qm)[8]) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool ka = k0 + 2 * kp + e < Kd;
            const unsigned ko = (unsigned)((tapA[e] * g.a_stap + cA[e] * g.a_sc) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) qa[2 * i + e] = bld1(rA, (ka && arow[i] != SWN_OOB) ? arow[i] + ko : SWN_OOB);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ts = t0 + tt + g.sgn * (tapB[i][e] - g.center) * g.dil;
                const bool ok = k0 + 2 * (kq + 4 * i) + e < Kd && ts >= 0 && ts < XT && tok;
                qb[2 * i + e] = bld1(rX, ok ? (unsigned)((cB[i][e] * g.x_sc + ts * g.x_st) * 4) : SWN_OOB);
                if (XMUL) qm[2 * i + e] = bld1(rM, ok ? (unsigned)(((size_t)cB[i][e] * g.xm_sc + ts) * 4) : SWN_OOB);
            }
            tapA[e] += dtap; cA[e] += dc; if (cA[e] >= g.KC) { cA[e] -= g.KC; ++tapA[e]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) { tapB[i][e] += dtap; cB[i][e] += dc; if (cB[i][e] >= g.KC) { cB[i][e] -= g.KC; ++tapB[i][e]; } }
        }
    };
#pragma unroll
    for (int u = 0; u < NST; ++u) fetch(kbeg + 32 * u, ra[u], rb[u], rm[u]);
    for (int k0 = kbeg; k0 < Kd; k0 += 32 * NST) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {                 // tiles past Kd hold zeros: no branch inside the loop
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                As[(tid >> 4) + 16 * i][kp] = swn_pack_bf16(ra[u][2 * i], ra[u][2 * i + 1]);
                Bs[tt][kq + 4 * i] = XMUL ? swn_pack_bf16(rb[u][2 * i] * rm[u][2 * i], rb[u][2 * i + 1] * rm[u][2 * i + 1])
                                          : swn_pack_bf16(rb[u][2 * i], rb[u][2 * i + 1]);
            }
            __syncthreads();
            fetch(k0 + 32 * (u + NST), ra[u], rb[u], rm[u]);
            swn_mmb_64x64x32(As, Bs, acc, lane, w);
            __syncthreads();
        }
    }
    const int t = t0 + swn_mma_col(lane, w);
    int mrow[16]; float v[16];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) { mrow[4 * mt + i] = m0 + swn_mma_row(lane, mt, i); v[4 * mt + i] = acc[mt][i]; }
    if (nsp > 1) {                                  // partial sums of one k part (no bias / mask / scale in this form)
        float* Y = g.Y + (size_t)b * g.y_sb + t;
#pragma unroll
        for (int i = 0; i < 16; ++i) if (mrow[i] < g.M && t < g.T) atomicAdd(Y + (size_t)mrow[i] * g.y_sm, v[i]);
        return;
    }
    tg_epilogue<16>(g, b, mrow, t, v);
}

// Third form of the time contraction (bf16 mode, KC % 32 == 0, time-contiguous X, no mask on X): workgroup tile
// 128 x 128 (2 x 2 waves of 64 x 64), and address arithmetic cut to one add per load: a k-tile of 32 never straddles
// a tap, so the tap and the channel base are workgroup-uniform scalars, the time shift is applied once per tile, and
// "reads as zero" is an out-of-range base that survives the additions (the weight view is given a 1 GiB range so
// that an invalid row and an invalid tile may both be markers without wrapping back into range).
// Staging: X as (column, 4 consecutive k) -> one 8-byte LDS store; A the same, by 16-byte loads when k is its
// contiguous axis (AKC) and by rows when m is (data gradients).
constexpr unsigned SWN_OOB_A = 0x40000000u;
template <bool AKC>
__global__ __launch_bounds__(256) void time_gemm_bf16t_kernel(const TimeGemm g) {
    // LDS double-buffered: one barrier per k-tile (a single buffer needed two: 388 -> 378 us per data gradient at REF6)
    __shared__ __attribute__((aligned(16))) unsigned As2[2][128][SWN_MMB_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned Bs2[2][128][SWN_MMB_PITCH];
    int buf = 0;
#define As As2[buf]
#define Bs Bs2[buf]
    // XCD-aware order (1-D grid; workgroup id i runs on XCD i % 8): each XCD walks a contiguous range of time tiles,
    // the row tiles of one time tile back to back - they read the same X window, and neighbouring time tiles share
    // their tap halos (up to (K-1)*dil positions, more than the tile itself at dil 49) in that XCD's L2.
    const int ntt = (g.T + 127) / 128, mtl = (g.M + 127) / 128;
    const int chunk = (ntt * g.nb + 7) / 8;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int ttl = idx / mtl, mtile = idx - ttl * mtl;
    const int gt = xcd * chunk + ttl;
    if (ttl >= chunk || gt >= ntt * g.nb) return;
    const int b = gt / ntt, t0 = (gt - b * ntt) * 128, m0 = mtile * 128;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.A), 0, 0x40000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = rsrc_of(g.X + (size_t)b * g.x_sb);
    const int ntiles = g.taps * (g.KC / 32);
    swn_f32x4 acc[4][4] = {};
    const int tb = tid & 127, qh = tid >> 7;                  // X: column tb, k-quads qh + 2 i
    const unsigned xs4 = (unsigned)(g.x_sc * 4);
    unsigned xk[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xk[i] = (unsigned)(4 * (qh + 2 * i)) * xs4;
    const int XT = g.XT ? g.XT : g.T;
    const bool tok = t0 + tb < g.T;
    // A: AKC: k-quad tid & 7 of rows (tid >> 3) + 32 i ; else: row tid & 127, k-quads qh + 2 i
    unsigned arow[4];
    const unsigned as4 = (unsigned)(g.a_sc * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (AKC) { const int m = m0 + (tid >> 3) + 32 * i; arow[i] = m < g.M ? (unsigned)((m * g.a_sm + 4 * (tid & 7)) * 4) : SWN_OOB_A; }
        else { const int m = m0 + tb; arow[i] = m < g.M ? (unsigned)(m * g.a_sm * 4) + (unsigned)(4 * (qh + 2 * i)) * as4 : SWN_OOB_A; }
    }
    int ftap = 0, fc0 = 0;                                     // (tap, channel base) of the next tile to fetch: scalars
    float ra[2][16], rb[2][16];
    auto fetch = [&](float (&qa)[16], float (&qb)[16]) {
        const bool live = fc0 < g.KC;
        const int ts = t0 + tb + g.sgn * (ftap - g.center) * g.dil;
        const unsigned xb = (live && tok && ts >= 0 && ts < XT) ? (unsigned)(ts * 4) + (unsigned)fc0 * xs4 : SWN_OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) qb[4 * i + e] = bld1(rX, xb + xk[i] + (unsigned)e * xs4);
        const unsigned sA = live ? (unsigned)((ftap * g.a_stap + fc0 * g.a_sc) * 4) : SWN_OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (AKC) {
                const swn_fl4 v = bld4(rA, arow[i] + sA);
                qa[4 * i] = v.x; qa[4 * i + 1] = v.y; qa[4 * i + 2] = v.z; qa[4 * i + 3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) qa[4 * i + e] = bld1(rA, arow[i] + sA + (unsigned)e * as4);
            }
        }
        // taps innermost: the K windows of one channel chunk overlap (shift dil < tile width) and are read back to back,
        // and neighbouring time tiles walk the chunks in step, so a chunk's halo is shared in L2 while it is hot
        // (taps outermost put 12 k-tiles between two reads of the same bytes: 1.29 GB fetched per data-gradient launch at the
        //  run.sh geometry against 0.38 GB in this order, FETCH_SIZE; the launch itself gained 1 % - the Infinity Cache had
        //  been absorbing the re-reads)
        if (++ftap >= g.taps) { ftap = 0; fc0 += 32; }
    };
    auto stage = [&](const float (&qa)[16], const float (&qb)[16]) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u2 vb = {swn_pack_bf16(qb[4 * i], qb[4 * i + 1]), swn_pack_bf16(qb[4 * i + 2], qb[4 * i + 3])};
            *reinterpret_cast<u2*>(&Bs[tb][2 * (qh + 2 * i)]) = vb;
            const u2 va = {swn_pack_bf16(qa[4 * i], qa[4 * i + 1]), swn_pack_bf16(qa[4 * i + 2], qa[4 * i + 3])};
            if (AKC) *reinterpret_cast<u2*>(&As[(tid >> 3) + 32 * i][2 * (tid & 7)]) = va;
            else *reinterpret_cast<u2*>(&As[tb][2 * (qh + 2 * i)]) = va;
        }
    };
    auto mma = [&]() {
        const int kq = lane >> 4, rc = lane & 15;
        swn_bf16x8 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i] = *reinterpret_cast<const swn_bf16x8*>(&As[64 * wm + 16 * i + rc][4 * kq]);
            fb[i] = *reinterpret_cast<const swn_bf16x8*>(&Bs[64 * wn + 16 * i + rc][4 * kq]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    fetch(ra[0], rb[0]);
    fetch(ra[1], rb[1]);
    for (int j = 0; j < ntiles; j += 2) {                      // a tile past the end is all zeros
        stage(ra[0], rb[0]);
        __syncthreads();                                       // every wave is past the MFMAs that read this buffer two tiles ago
        fetch(ra[0], rb[0]);
        mma();
        buf ^= 1;
        stage(ra[1], rb[1]);
        __syncthreads();
        fetch(ra[1], rb[1]);
        mma();
        buf ^= 1;
    }
#undef As
#undef Bs
    const int kq = lane >> 4, rc = lane & 15;
    int mrow[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) mrow[4 * i + e] = m0 + 64 * wm + 16 * i + 4 * kq + e;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * i + e] = acc[i][j][e];
        tg_epilogue<16>(g, b, mrow, t0 + 64 * wn + 16 * j + rc, v);
    }
}


// Time contraction with BOTH operands in bf16 (mixed-precision chain at the GEMM-stack geometries: the layer data gradients, and
// the in_x products of the dropout chain): A from a bf16 matrix whose k axis is contiguous (16-byte loads of eight k straight into
// LDS), X from bf16 rows [channel][t] - t contiguous, i.e. the k axis of the product is the STRIDED one.  A k-tile of X (32 channel
// rows x 128 positions) is staged exactly as it lies in memory - two 16-byte loads per thread, eight lanes = one 128-byte row
// segment, 16-byte LDS stores - and the MFMA B fragments (eight consecutive k of one position) are gathered by the transposing
// LDS read ds_read_b64_tr_b16 (two per fragment).  The first form of this kernel gave a thread a column pair of eight k rows:
// eight dword loads per k-tile and two byte-permutes per LDS store - 32 load instructions per workgroup and k-tile where this
// form issues 8.  An odd tap shift reads the second copy of X (moved right by one position: x16_odd), so that every 16-byte
// piece starts on a dword.  LDS image: rows of 256 bytes, the 32-byte chunk pairs of row r XOR-swizzled by (r & 3) | ((r >> 3) & 1) << 2:
// the eight rows a transposed read touches per 32-lane half land on eight different bank groups.
// The buffer view of X starts 16 bytes before the operand (a piece that straddles position 0 has a negative offset): X16 must
// not be the first bytes of an allocation (it never is: the copies live inside the work buffers).
// 16 operand registers per thread and stage; three workgroups per CU (launch bound) - these loops wait on memory round trips.
// MT = 16-row accumulator tiles per wave: the workgroup tile is 32 MT rows x 128 positions.  M = 192 (the run.sh Laplace net) in
// 128-row tiles left the second row tile half empty - a quarter of the MFMAs and of the A traffic on zeros; 96-row tiles fit.
typedef short swn_s16x4 __attribute__((ext_vector_type(4)));
typedef short swn_s16x8 __attribute__((ext_vector_type(8)));
// (The 128-row form needs 183 registers and spills 15 under the three-workgroup bound; bound to two workgroups it measured slower on
// the long-k data gradient of in_x - 530 against 475 us - and the same on the output-bound forward product.)
template <int MT>
__global__ __launch_bounds__(256, 3) void time_gemm_b16_kernel(const TimeGemm g) {
    constexpr int RM = 32 * MT;
    __shared__ __attribute__((aligned(16))) unsigned As2[2][128][SWN_MMB_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char Xs2[2][32 * 256];
    const int ntt = (g.T + 127) / 128, mtl = (g.M + RM - 1) / RM;
    const int chunk = (ntt * g.nb + 7) / 8;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int ttl = idx / mtl, mtile = idx - ttl * mtl;
    const int gt = xcd * chunk + ttl;
    if (ttl >= chunk || gt >= ntt * g.nb) return;
    const int b = gt / ntt, t0 = (gt - b * ntt) * 128, m0 = mtile * RM;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(g.A16), 0, 0x40000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = rsrc_of(g.X16 + (size_t)b * g.x16_sb - 8);      // every X offset below carries + 8 elements
    const int ntiles = g.taps * (g.KC / 32);
    swn_f32x4 acc[MT][4] = {};
    const int XT = g.XT ? g.XT : g.T;
    const int a8 = tid & 3, ar0 = tid >> 2;                    // A: k-octet a8 of rows ar0, ar0 + 64
    unsigned arow[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int m = m0 + ar0 + 64 * i; arow[i] = (m < g.M && ar0 + 64 * i < RM) ? (unsigned)((m * g.a16_sm + 8 * a8) * 2) : SWN_OOB_A; }
    const int xr = tid >> 3, xc = tid & 7;                     // X: k row xr of the tile, 16-byte pieces xc and xc + 8 of its 128 positions
    const unsigned rs2 = (unsigned)(g.x16_sc * 2);
    unsigned xdst[2];
    {
        const int sw = (xr & 3) | (((xr >> 3) & 1) << 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int c = xc + 8 * i; xdst[i] = (unsigned)(xr * 256 + (((((c >> 1) ^ sw) << 1) | (c & 1)) << 4)); }
    }
    int ftap = 0, fc0 = 0;
    swn_fl4 ra[2][2], rb[2][2]; int rsh[2];
    auto fetch = [&](swn_fl4 (&qa)[2], swn_fl4 (&qb)[2], int& qsh) {
        const bool live = fc0 < g.KC;
        const unsigned sA = live ? (unsigned)((ftap * g.a16_stap + fc0) * 2) : SWN_OOB;
#pragma unroll
        for (int i = 0; i < 2; ++i) qa[i] = bld4(rA, arow[i] + sA);
        const int sh = g.sgn * (ftap - g.center) * g.dil, par = sh & 1;
        qsh = sh;
        // even shift: positions ts .. ts + 7 of the plain copy; odd: elements ts + 1 .. of the copy moved right by one (a dword start)
        const unsigned rowb = (unsigned)(fc0 + xr) * rs2 + (unsigned)(((par ? g.x16_odd + 1 : 0) + 8) * 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ts = t0 + 8 * (xc + 8 * i) + sh;
            qb[i] = bld4(rX, (live && ts + 7 >= 0 && ts < XT) ? rowb + (unsigned)(ts * 2) : SWN_OOB);
        }
        if (++ftap >= g.taps) { ftap = 0; fc0 += 32; }         // taps innermost, as in time_gemm_bf16t_kernel
    };
    auto stage = [&](int buf, const swn_fl4 (&qa)[2], swn_fl4 (&qb)[2], const int sh) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<swn_fl4*>(&As2[buf][ar0 + 64 * i][4 * a8]) = qa[i];
            const int ts = t0 + 8 * (xc + 8 * i) + sh;
            if (ts < 0 || ts + 8 > XT) {                       // rare: the piece straddles an end of the sequence
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int e0 = ts + 2 * d, e1 = e0 + 1;
                    const unsigned mk = ((e0 >= 0 && e0 < XT) ? 0x0000ffffu : 0u) | ((e1 >= 0 && e1 < XT) ? 0xffff0000u : 0u);
                    qb[i][d] = __uint_as_float(__float_as_uint(qb[i][d]) & mk);
                }
            }
            *reinterpret_cast<swn_fl4*>(&Xs2[buf][xdst[i]]) = qb[i];
        }
    };
    // transposed-read addresses of this lane: lane 4q + p of the 16-lane group kq supplies row 8 kq + 4 h + q, bytes 8 p .. 8 p + 7 of
    // the 32-byte chunk pair of n-tile J (swizzled by the row)
    const int kq = lane >> 4, rc = lane & 15;
    const unsigned trb = (unsigned)((8 * kq + (rc >> 2)) * 256 + ((rc & 3) << 3));
    const int trs = (rc >> 2) | ((kq & 1) << 2);
    auto mma = [&](int buf) {
        swn_bf16x8 fa[MT], fb[4];
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const swn_bf16x8*>(&As2[buf][16 * MT * wm + 16 * i + rc][4 * kq]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned char* src = &Xs2[buf][trb + (unsigned)(((4 * wn + j) ^ trs) << 5)];
            const swn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) swn_s16x4*)(src));
            const swn_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) swn_s16x4*)(src + 4 * 256));
            const swn_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            fb[j] = __builtin_bit_cast(swn_bf16x8, v);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    fetch(ra[0], rb[0], rsh[0]);
    fetch(ra[1], rb[1], rsh[1]);
    for (int j = 0; j < ntiles; j += 2) {                      // a tile past the end is all zeros
        stage(0, ra[0], rb[0], rsh[0]);
        __syncthreads();                                       // every wave is past the MFMAs that read this buffer two tiles ago
        fetch(ra[0], rb[0], rsh[0]);
        mma(0);
        stage(1, ra[1], rb[1], rsh[1]);
        __syncthreads();
        fetch(ra[1], rb[1], rsh[1]);
        mma(1);
    }
    int mrow[4 * MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) mrow[4 * i + e] = m0 + 16 * MT * wm + 16 * i + 4 * kq + e;
    if (g.y_g4) {                                              // a lane's four rows of a position: one 16-byte piece (M % 4 == 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = t0 + 64 * wn + 16 * j + rc;
#pragma unroll
            for (int i = 0; i < MT; ++i)
                if (t < g.T && mrow[4 * i] < g.M) *reinterpret_cast<swn_f32x4*>(g.Y + swn_g4(g.M, g.T, b, mrow[4 * i], t)) = acc[i][j];
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v[4 * MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * i + e] = acc[i][j][e];
        tg_epilogue<4 * MT>(g, b, mrow, t0 + 64 * wn + 16 * j + rc, v);
    }
}

// transposed bf16 copy of the dilated-conv matrices for time_gemm_b16_kernel: dst[l][tap][i][o2] = Wd[l][o2][tap][i]
__global__ __launch_bounds__(256) void wd_t16_kernel(const float* __restrict__ wd, unsigned short* __restrict__ dst, const int L,
                                                     const int K, const int H, const int Hp) {
    const int H2 = 2 * H;
    const size_t n = (size_t)L * K * H * H2;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int o2 = (int)(e % H2); size_t r = e / H2;
        const int i = (int)(r % H); r /= H;
        const int tap = (int)(r % K); const int l = (int)(r / K);
        dst[e] = (unsigned short)(swn_pack_bf16(wd[(((size_t)l * H2 + o2) * K + tap) * Hp + i], 0.f) & 0xffffu);
    }
}

struct ReduceGemm {
    const float* P; long p_sb, p_sm, p_st;          // P[b][m][t]
    const float* Q; long q_sb, q_sc, q_st;          // Q[b][c][t]
    float* G; long g_sm, g_stap, g_sc;              // G(m, tap, c)  += ...
    float* gb;                                      // optional: gb[m] += sum_{b,t} P
    int M, taps, KC, T, sgn, center, dil, TS;       // TS: time positions per block
    const float* qmul; long qm_sb, qm_sc;           // optional: Q[b][c][t] is read as Q * qmul[b][c][t]
    int QT;                                         // Q is valid on [0, QT) (0: same as T)
    int nsegtot;                                    // batch * time segments (set by the launcher)
    const unsigned short* P16; long p16_sb, p16_sm; // optional bf16 copy of P (rows 16-byte aligned, readable up to the next multiple of 32)
    const unsigned short* Q16; long q16_sb, q16_sc, q16_odd;   // optional (with P16): two bf16 copies of Q, the second q16_odd elements on and
                                                    // moved right by one position (copy[u] = Q[u - 1]): an octet shifted by any tap
                                                    // starts at an even element of one of them
};

__global__ __launch_bounds__(256) void reduce_gemm_kernel(const ReduceGemm g) {
    __shared__ float Ps[16][SWN_MMA_PITCH];
    __shared__ float Qs[16][SWN_MMA_PITCH];
    // XCD-aware order, as in reduce_gemm_bf16s_kernel: 1-D grid, the tiles of one time segment back to back on one XCD
    const int nseg = (g.T + g.TS - 1) / g.TS;
    const int Nc = g.taps * g.KC;
    const int mtl = (g.M + 63) / 64, ntl = (Nc + 63) / 64, per_seg = mtl * ntl;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int sg = (idx / per_seg) * 8 + xcd, tile = idx - (idx / per_seg) * per_seg;
    if (sg >= g.nsegtot) return;
    const int b = sg / nseg, ts0 = (sg - b * nseg) * g.TS;
    const int by = tile / mtl, bx = tile - by * mtl;
    const int m0 = bx * 64, n0 = by * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* Pb = g.P + (size_t)b * g.p_sb;
    const float* Qb = g.Q + (size_t)b * g.q_sb;
    swn_f32x4 acc[4] = {};
    float rs[4] = {0.f, 0.f, 0.f, 0.f};        // partial row sums of P: thread (tid&15 = time, tid>>4 + 16 i = row)
    const int tend = ts0 + g.TS < g.T ? ts0 + g.TS : g.T;
    // the four Q columns this thread stages: (tap, c) and the time shift, fixed for the whole time loop
    int qc[4], qshift[4]; bool qok[4];
    const int QT = g.QT ? g.QT : g.T;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int nc = n0 + (tid >> 4) + 16 * i;
        const int tap = nc / g.KC;
        qok[i] = nc < Nc; qc[i] = nc - tap * g.KC; qshift[i] = g.sgn * (tap - g.center) * g.dil;
    }
    for (int t0 = ts0; t0 < tend; t0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int kk = e & 15, mm = e >> 4;        // time fastest: coalesced when p_st == 1
            const int t = t0 + kk, m = m0 + mm;
            const float v = (t < tend && m < g.M) ? Pb[m * g.p_sm + t * g.p_st] : 0.f;
            Ps[kk][mm] = v;
            rs[i] += v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = tid & 15, nn = (tid >> 4) + 16 * i;
            const int t = t0 + kk;
            float v = 0.f;
            if (t < tend && qok[i]) {
                const int tsrc = t + qshift[i];
                if (tsrc >= 0 && tsrc < QT) {
                    v = Qb[qc[i] * g.q_sc + tsrc * g.q_st];
                    if (g.qmul) v *= g.qmul[(size_t)b * g.qm_sb + (size_t)qc[i] * g.qm_sc + tsrc];
                }
            }
            Qs[kk][nn] = v;
        }
        __syncthreads();
        swn_mma_64x64x16(Ps, Qs, acc, lane, w);
        __syncthreads();
    }
    if (g.gb && by == 0) {            // bias gradient: rows summed over the 16 time lanes of each group
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = rs[i];
            v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
            const int m = m0 + (tid >> 4) + 16 * i;
            if ((tid & 15) == 0 && m < g.M) atomicAdd(g.gb + m, v);
        }
    }
    const int nc = n0 + swn_mma_col(lane, w);
    if (nc >= Nc) return;
    const int tap = nc / g.KC, c = nc - tap * g.KC;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + swn_mma_row(lane, mt, i);
            if (m < g.M) atomicAdd(g.G + m * g.g_sm + tap * g.g_stap + c * g.g_sc, acc[mt][i]);
        }
}

// bf16-operand twin of reduce_gemm_kernel: time tiles of 32, operands fetched through buffer resources two tiles
// ahead (no branch around a load), rounded to bf16 into LDS; row sums for the bias gradient stay exact fp32.
template <bool QMUL>
__global__ __launch_bounds__(256) void reduce_gemm_bf16_kernel(const ReduceGemm g) {
    __shared__ __attribute__((aligned(16))) unsigned Ps[64][SWN_MMB_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned Qs[64][SWN_MMB_PITCH];
    const int nseg = (g.T + g.TS - 1) / g.TS;
    const int b = blockIdx.z / nseg, ts0 = (blockIdx.z - b * nseg) * g.TS;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int Nc = g.taps * g.KC;
    const __amdgpu_buffer_rsrc_t rP = rsrc_of(g.P + (size_t)b * g.p_sb), rQ = rsrc_of(g.Q + (size_t)b * g.q_sb);
    const __amdgpu_buffer_rsrc_t rM = rsrc_of(QMUL ? g.qmul + (size_t)b * g.qm_sb : g.P);
    swn_f32x4 acc[4] = {};
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    const int tend = ts0 + g.TS < g.T ? ts0 + g.TS : g.T;
    const int QT = g.QT ? g.QT : g.T;
    const int kp = tid & 15;                    // time pair (2kp, 2kp+1) of the tile; rows / columns (tid >> 4) + 16 i
    int qc[4], qshift[4]; bool qok[4], mok[4]; int pm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int nc = n0 + (tid >> 4) + 16 * i;
        const int tap = nc / g.KC;
        qok[i] = nc < Nc; qc[i] = nc - tap * g.KC; qshift[i] = g.sgn * (tap - g.center) * g.dil;
        pm[i] = m0 + (tid >> 4) + 16 * i; mok[i] = pm[i] < g.M;
    }
    constexpr int NST = 2;
    float rp[NST][8], rq[NST][8], rm[NST][8];
    auto fetch = [&](int t0, float (&qp)[8], float (&qq)[8], float (&qm)[8]) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int t = t0 + 2 * kp + e;
            const bool okt = t < tend;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                qp[2 * i + e] = bld1(rP, (okt && mok[i]) ? (unsigned)((pm[i] * g.p_sm + t * g.p_st) * 4) : SWN_OOB);
                const int tsrc = t + qshift[i];
                const bool ok = okt && qok[i] && tsrc >= 0 && tsrc < QT;
                qq[2 * i + e] = bld1(rQ, ok ? (unsigned)((qc[i] * g.q_sc + tsrc * g.q_st) * 4) : SWN_OOB);
                if (QMUL) qm[2 * i + e] = bld1(rM, ok ? (unsigned)(((size_t)qc[i] * g.qm_sc + tsrc) * 4) : SWN_OOB);
            }
        }
    };
#pragma unroll
    for (int u = 0; u < NST; ++u) fetch(ts0 + 32 * u, rp[u], rq[u], rm[u]);
    for (int t0 = ts0; t0 < tend; t0 += 32 * NST) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {                 // tiles past tend hold zeros
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rs[i] += rp[u][2 * i] + rp[u][2 * i + 1];
                Ps[(tid >> 4) + 16 * i][kp] = swn_pack_bf16(rp[u][2 * i], rp[u][2 * i + 1]);
                Qs[(tid >> 4) + 16 * i][kp] = QMUL ? swn_pack_bf16(rq[u][2 * i] * rm[u][2 * i], rq[u][2 * i + 1] * rm[u][2 * i + 1])
                                                    : swn_pack_bf16(rq[u][2 * i], rq[u][2 * i + 1]);
            }
            __syncthreads();
            fetch(t0 + 32 * (u + NST), rp[u], rq[u], rm[u]);
            swn_mmb_64x64x32(Ps, Qs, acc, lane, w);
            __syncthreads();
        }
    }
    if (g.gb && blockIdx.y == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = rs[i];
            v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
            if (kp == 0 && mok[i]) atomicAdd(g.gb + pm[i], v);
        }
    }
    const int nc = n0 + swn_mma_col(lane, w);
    if (nc >= Nc) return;
    const int tap = nc / g.KC, c = nc - tap * g.KC;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + swn_mma_row(lane, mt, i);
            if (m < g.M) atomicAdd(g.G + m * g.g_sm + tap * g.g_stap + c * g.g_sc, acc[mt][i]);
        }
}

// bf16 form of the weight-gradient contraction for time-contiguous operands (p_st == q_st == 1, no mask): the
// reduction index IS the contiguous axis of both operands.  Workgroup tile 128 x 128 shared through LDS (a first cut
// loaded MFMA fragments straight from HBM into registers, no LDS at all - bound by the L1 path, every wave fetching its
// own copy: 64 KB per step against 256 MFMA cycles).  A thread fetches 16-byte time-quads (8 lanes = one 128-byte row segment), rounds them to bf16 and stores
// 8 bytes into LDS; double-buffered, one barrier per step of 32 positions.
// P16: the P operand comes from its bf16 copy (g.P16: eight positions per 16-byte load, no conversion, half the bytes in
// flight per step - the kernel waits on memory round trips, not on the matrix cores); the row sums of the bias gradient are
// then sums of the rounded values.
// Q16 (with P16): Q too comes from bf16 copies - two of them, the second moved right by one position, so that an octet shifted
// by any tap starts at an even element of one of the two (gate_bwd writes both for the layer's input): 341 -> 277 us per layer
// at REF6 with three workgroups per CU (a launch bound of four - 128 registers, LDS exactly full - measured the same).
template <bool P16, bool Q16 = false>
__global__ __launch_bounds__(256) void reduce_gemm_bf16s_kernel(const ReduceGemm g) {
    __shared__ __attribute__((aligned(16))) unsigned Ps[2][128][SWN_MMB_PITCH];
    __shared__ __attribute__((aligned(16))) unsigned Qs[2][128][SWN_MMB_PITCH];
    // XCD-aware order (1-D grid; workgroup id i runs on XCD i % 8): the mt*nt tiles that read the same time segment are
    // dispatched back to back on ONE XCD and walk the segment together, so its rows come out of HBM once per segment
    // and are shared in that XCD's L2 instead of being fetched by up to eight L2s.
    const int nseg = (g.T + g.TS - 1) / g.TS;
    const int Nc = g.taps * g.KC;
    const int mtl = (g.M + 127) / 128, ntl = (Nc + 127) / 128, per_seg = mtl * ntl;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int sg = (idx / per_seg) * 8 + xcd, tile = idx - (idx / per_seg) * per_seg;
    if (sg >= g.nsegtot) return;
    const int b = sg / nseg, ts0 = (sg - b * nseg) * g.TS;
    const int by = tile / mtl, bx = tile - by * mtl;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int m0 = bx * 128, n0 = by * 128;
    const __amdgpu_buffer_rsrc_t rP = rsrc_of(g.P + (size_t)b * g.p_sb), rQ = rsrc_of(g.Q + (size_t)b * g.q_sb);
    const int tend = ts0 + g.TS < g.T ? ts0 + g.TS : g.T;
    const int QT = g.QT ? g.QT : g.T;
    const int smax = (g.taps - 1) * g.dil;
    const int q4 = tid & 7, r0 = tid >> 3;                  // staging: time-quad q4 of rows / columns r0 + 32 i
    const __amdgpu_buffer_rsrc_t rP16 = rsrc_of(P16 ? g.P16 + (size_t)b * g.p16_sb : nullptr);
    const int p8 = tid & 3, pr0 = tid >> 2;                 // P16 staging: time-octet p8 of rows pr0 + 64 i
    unsigned prow16[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int m = m0 + pr0 + 64 * i; prow16[i] = m < g.M ? (unsigned)(m * g.p16_sm * 2) : SWN_OOB; }
    const bool want_rs = g.gb && by == 0;
    unsigned prow[4], qrow[4]; int qsh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + 32 * i;
        prow[i] = m < g.M ? (unsigned)(m * g.p_sm * 4) : SWN_OOB;
        const int n = n0 + r0 + 32 * i;
        const int tap = n / g.KC, c = n - tap * g.KC;
        qsh[i] = g.sgn * (tap - g.center) * g.dil;
        qrow[i] = n < Nc ? (unsigned)((c * g.q_sc + qsh[i]) * 4) : SWN_OOB;     // modular: may be "negative"
    }
    // Q16: octet p8 of columns pr0 + 64 i; the row base holds the tap shift (even part) and the copy its parity selects
    const __amdgpu_buffer_rsrc_t rQ16 = rsrc_of(Q16 ? g.Q16 + (size_t)b * g.q16_sb : nullptr);
    unsigned qrow16[2]; int qsh16[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int n = n0 + pr0 + 64 * i;
        const int tap = n / g.KC, c = n - tap * g.KC;
        const int sft = g.sgn * (tap - g.center) * g.dil, par = sft & 1;
        qsh16[i] = sft;
        qrow16[i] = n < Nc ? (unsigned)((c * g.q16_sc + sft + par + (par ? g.q16_odd : 0)) * 2) : SWN_OOB;     // modular
    }
    swn_f32x4 acc[4][4] = {};
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    swn_fl4 ra[2][4], rb[2][4];              // two steps of operands in flight (HBM latency is ~10 steps of MFMA work)
                                             // (P16: ra[.][0..1] hold the two 16-byte octets of bf16 bits)
    auto load_fast = [&](int t, swn_fl4 (&ra)[4], swn_fl4 (&rb)[4]) {
        const unsigned to = (unsigned)((t + 4 * q4) * 4);
        if (P16) {
#pragma unroll
            for (int i = 0; i < 2; ++i) ra[i] = bld4(rP16, prow16[i] + (unsigned)((t + 8 * p8) * 2));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = bld4(rP, prow[i] + to);
        }
        if (Q16) {
#pragma unroll
            for (int i = 0; i < 2; ++i) rb[i] = bld4(rQ16, qrow16[i] + (unsigned)((t + 8 * p8) * 2));
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) rb[i] = bld4(rQ, qrow[i] + to);
        }
    };
    auto load_edge = [&](int t, swn_fl4 (&ra)[4], swn_fl4 (&rb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int te = t + 4 * q4 + e;
                const bool okp = te < tend;
                const int tsrc = te + qsh[i];
                const bool okq = okp && tsrc >= 0 && tsrc < QT && qrow[i] != SWN_OOB;
                if (!P16) ra[i][e] = bld1(rP, okp ? prow[i] + (unsigned)(te * 4) : SWN_OOB);
                if (!Q16) rb[i][e] = bld1(rQ, okq ? qrow[i] + (unsigned)(te * 4) : SWN_OOB);
            }
        if (Q16) {      // pair by pair: a dword of the selected copy holds the shifted positions (te, te + 1); invalid halves masked off
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int te = t + 8 * p8 + 2 * d, ts0 = te + qsh16[i];
                    const bool k0 = te < tend && ts0 >= 0 && ts0 < QT && qrow16[i] != SWN_OOB;
                    const bool k1 = te + 1 < tend && ts0 + 1 >= 0 && ts0 + 1 < QT && qrow16[i] != SWN_OOB;
                    const unsigned mk = (k0 ? 0x0000ffffu : 0u) | (k1 ? 0xffff0000u : 0u);
                    rb[i][d] = __uint_as_float(__float_as_uint(bld1(rQ16, (k0 || k1) ? qrow16[i] + (unsigned)(te * 2) : SWN_OOB)) & mk);
                }
        }
        if (P16) {      // whole octets (the copy's rows are readable up to the next multiple of 32), positions >= tend masked off
            const int nv = tend - (t + 8 * p8);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const swn_fl4 v = bld4(rP16, nv > 0 ? prow16[i] + (unsigned)((t + 8 * p8) * 2) : SWN_OOB);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const unsigned mk = 2 * d + 1 < nv ? 0xffffffffu : (2 * d < nv ? 0x0000ffffu : 0u);
                    ra[i][d] = __uint_as_float(__float_as_uint(v[d]) & mk);
                }
            }
        }
    };
    auto stage = [&](int buf, const swn_fl4 (&ra)[4], const swn_fl4 (&rb)[4]) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        if (P16) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (want_rs) {
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const unsigned u = __float_as_uint(ra[i][d]);
                        rs[i] += __uint_as_float(u << 16) + __uint_as_float(u & 0xffff0000u);
                    }
                }
                *reinterpret_cast<swn_fl4*>(&Ps[buf][pr0 + 64 * i][4 * p8]) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (!P16) {
                rs[i] += (ra[i].x + ra[i].y) + (ra[i].z + ra[i].w);
                const u2 vp = {swn_pack_bf16(ra[i].x, ra[i].y), swn_pack_bf16(ra[i].z, ra[i].w)};
                *reinterpret_cast<u2*>(&Ps[buf][r0 + 32 * i][2 * q4]) = vp;
            }
            if (!Q16) {
                const u2 vq = {swn_pack_bf16(rb[i].x, rb[i].y), swn_pack_bf16(rb[i].z, rb[i].w)};
                *reinterpret_cast<u2*>(&Qs[buf][r0 + 32 * i][2 * q4]) = vq;
            }
        }
        if (Q16) {
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<swn_fl4*>(&Qs[buf][pr0 + 64 * i][4 * p8]) = rb[i];
        }
    };
    auto mma = [&](int buf) {
        const int kq = lane >> 4, rc = lane & 15;
        swn_bf16x8 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i] = *reinterpret_cast<const swn_bf16x8*>(&Ps[buf][64 * wm + 16 * i + rc][4 * kq]);
            fb[i] = *reinterpret_cast<const swn_bf16x8*>(&Qs[buf][64 * wn + 16 * i + rc][4 * kq]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    };
    // steps [sa, sb) are interior: every shifted 32-window inside [0, QT) and inside the segment
    const int nsteps = (tend - ts0 + 31) / 32;
    int sa = ts0 >= smax ? 0 : (smax - ts0 + 31) / 32;
    const int lim = QT - smax < tend ? QT - smax : tend;          // t + 32 <= lim
    int sb = lim - ts0 >= 32 ? (lim - ts0) / 32 : 0;
    if (sa > nsteps) sa = nsteps;
    if (sb > nsteps) sb = nsteps;
    if (sb < sa) sb = sa;
    for (int sidx = 0; sidx < sa; ++sidx) { load_edge(ts0 + 32 * sidx, ra[0], rb[0]); stage(0, ra[0], rb[0]); __syncthreads(); mma(0); __syncthreads(); }
    if ((sb - sa) & 1) { load_fast(ts0 + 32 * sa, ra[0], rb[0]); stage(0, ra[0], rb[0]); __syncthreads(); mma(0); __syncthreads(); ++sa; }
    if (sb > sa) {
        const int last = sb - 1;                            // an even number of interior steps from here on
        load_fast(ts0 + 32 * sa, ra[0], rb[0]);
        load_fast(ts0 + 32 * (sa + 1), ra[1], rb[1]);
        for (int sidx = sa; sidx < sb; sidx += 2) {         // loads past the end re-read the last step: harmless, branch-free
            stage(0, ra[0], rb[0]);
            __syncthreads();
            load_fast(ts0 + 32 * (sidx + 2 < sb ? sidx + 2 : last), ra[0], rb[0]);
            mma(0);
            stage(1, ra[1], rb[1]);
            __syncthreads();
            load_fast(ts0 + 32 * (sidx + 3 < sb ? sidx + 3 : last), ra[1], rb[1]);
            mma(1);
        }
        __syncthreads();
    }
    for (int sidx = sb; sidx < nsteps; ++sidx) { load_edge(ts0 + 32 * sidx, ra[0], rb[0]); stage(0, ra[0], rb[0]); __syncthreads(); mma(0); __syncthreads(); }

    if (P16 && want_rs) {           // a row's four time-octets sit in four neighbouring lanes
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v = rs[i];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2);
            const int m = m0 + pr0 + 64 * i;
            if (p8 == 0 && m < g.M) atomicAdd(g.gb + m, v);
        }
    }
    if (!P16 && g.gb && by == 0) {  // bias gradient: a row's eight time-quads sit in eight neighbouring lanes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = rs[i];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
            const int m = m0 + r0 + 32 * i;
            if (q4 == 0 && m < g.M) atomicAdd(g.gb + m, v);
        }
    }
    const int kq = lane >> 4, rc = lane & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + 64 * wn + 16 * j + rc;
        if (n >= Nc) continue;
        const int tap = n / g.KC, c = n - tap * g.KC;
        float* Gn = g.G + tap * g.g_stap + c * g.g_sc;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + 64 * wm + 16 * i + 4 * kq + e;
                if (m < g.M) atomicAdd(Gn + m * g.g_sm, acc[i][j][e]);
            }
    }
}

// ---- gate backward: recomputed pre-activations a (B,2H,T) -> da, dgx (in place over a / second buffer),
//      and the highway carry  dh_prev += dh * z.      (cswnv_shift1.py:276-278)
struct GateBwd {
    SwnGeom g; SwnLayout y;
    const float* P; const float* cond; const void* audio;
    const float* hs; float* dhs; float* a_da; float* dgx;
    long dgx_sb;           // elements between two utterances' (2H, Tp) blocks of dgx
    int B, Tf, Tp, coff, l;
    // dropout mode (appended, null = off)
    const float* gx;       // (B, L, 2H, Tp) sample-rate in_x products (no bias) of the masked conditioning
    const float* in_mul;   // (B, H, Tp) mask on this layer's INPUT h_{l-1}: it was the dropped output of layer l-1
    float* gwxa;           // softmax audio_in: gradient of the one-hot columns of in_x, [L][Q][2H] (packed wxa section)
    const float* a_in;     // gate pre-activations to read (null: a_da, where the recompute GEMM just left them)
    unsigned short* da16; long da16_pitch;     // optional bf16 copies of da, rows of da16_pitch elements: the plain one (weight-
    long da16_odd; int skip_da32;              // gradient P operand) and, da16_odd elements on, one moved right by a position
                                               // (with the plain one: the data gradient's X operand, TimeGemm::X16);
                                               // skip_da32: both GEMMs read the copies, the fp32 da is not stored
    unsigned short* h16; long h16_odd;         // optional: the layer's (masked) input as two bf16 copies like da16's (rows of
                                               // da16_pitch elements): the weight gradient's Q operand (ReduceGemm::Q16)
    int g4;                                    // a_in and gx are in the G4 layout (swn_geom.hpp): the GEMM stack wrote / read them
    unsigned short* dgx16; long dgx16_sb;      // optional (swn_drop_inx16): d gx as bf16 rows of da16_pitch elements INSTEAD of the
                                               // fp32 dgx - the operand of the two merged in_x contractions behind the layer loop
};

// One (position, channel) of the gate backward.  G4: the pre-activations (and, dropout mode, the in_x rows) arrive as values - the
// caller read its four channels' 16-byte pieces of the G4 layout (swn_geom.hpp) - instead of being read from (B, R, Tp) tensors.
struct GbIn { float im, hprev, dh, dhl; };     // per (position, channel): input mask, (masked) layer input, d h_{l+1}, d h_l so far
__device__ __forceinline__ GbIn gate_bwd_in(const GateBwd& a, const int b, const int t, const int o) {
    const int H = a.g.H;
    const size_t hb = ((size_t)b * (a.g.L + 1)) * H * a.Tp;
    GbIn r;
    r.im = a.in_mul ? a.in_mul[((size_t)b * H + o) * a.Tp + t] : 1.f;
    r.hprev = a.hs[hb + ((size_t)a.l * H + o) * a.Tp + t] * r.im;
    r.dh = a.dhs[hb + ((size_t)(a.l + 1) * H + o) * a.Tp + t];
    r.dhl = a.dhs[hb + ((size_t)a.l * H + o) * a.Tp + t];
    return r;
}
template <int KIND, bool G4>
__device__ __forceinline__ void gate_bwd_one(const GateBwd& a, const int b, const int t, const int o, const GbIn in, const float pz,
                                             const float pc, const float xz, const float xc) {
    const SwnGeom& g = a.g;
    const int H = g.H, H2 = 2 * g.H, l = a.l, seg = g.seg;
    const float* P = a.P;
    const size_t hb = ((size_t)b * (g.L + 1)) * H * a.Tp;
    const float im = in.im, hprev = in.hprev, dh = in.dh;
    float* az = a.a_da + ((size_t)b * H2 + o) * a.Tp + t;
    float* ac = a.a_da + ((size_t)b * H2 + H + o) * a.Tp + t;
    float gz, gc;
    if (G4 && a.gx) {
        gz = xz + P[a.y.bxr + (size_t)l * H2 + o];
        gc = xc + P[a.y.bxr + (size_t)l * H2 + H + o];
    } else if (a.gx) {
        const float* gr = a.gx + (((size_t)b * g.L + l) * H2) * a.Tp + t;
        gz = gr[(size_t)o * a.Tp] + P[a.y.bxr + (size_t)l * H2 + o];
        gc = gr[(size_t)(H + o) * a.Tp] + P[a.y.bxr + (size_t)l * H2 + H + o];
    } else {
        gz = P[a.y.bx + (size_t)l * H2 + o]; gc = P[a.y.bx + (size_t)l * H2 + H + o];
        const float* condb = a.cond + (size_t)b * a.Tf * g.N;
        for (int s = 0; s < seg; ++s) {
            const int tt = t + s + a.coff;
            int f = tt / g.U; const int jj = tt - f * g.U;
            f = f < a.Tf ? f : a.Tf - 1;
            const float w = P[a.y.wup + jj];
            const float* cr = condb + (size_t)f * g.N + (size_t)(l * seg + s) * H2;
            gz = fmaf(w, cr[o], gz); gc = fmaf(w, cr[H + o], gc);
        }
    }
    int idx = 0;
    if (KIND == SWN_KIND_SOFTMAX && g.audio_in) {
        idx = reinterpret_cast<const int*>(a.audio)[(size_t)b * a.Tp + t] % g.Q; idx = idx < 0 ? idx + g.Q : idx;
        const float* wa = P + a.y.wxa + ((size_t)l * g.Q + idx) * H2;
        gz += wa[o]; gc += wa[H + o];
    }
    const float* ain = a.a_in ? a.a_in : a.a_da;
    const float sz = (G4 ? pz : ain[((size_t)b * H2 + o) * a.Tp + t]) + P[a.y.bd + (size_t)l * H2 + o];
    const float sc = (G4 ? pc : ain[((size_t)b * H2 + H + o) * a.Tp + t]) + P[a.y.bd + (size_t)l * H2 + H + o];
    const float z = sigm(gz * sz), c = tanhf(gc * sc);
    const float dz = dh * (hprev - c) * z * (1.f - z);      // d/d(gz*sz)
    const float dc = dh * (1.f - z) * (1.f - c * c);        // d/d(gc*sc)
    if (!a.skip_da32) { *az = dz * gz; *ac = dc * gc; }     // da
    if (a.h16) {
        const unsigned short hu = (unsigned short)(swn_pack_bf16(hprev, 0.f) & 0xffffu);
        unsigned short* rh = a.h16 + ((size_t)b * H + o) * a.da16_pitch + t;
        rh[0] = hu; rh[a.h16_odd + 1] = hu;
    }
    if (a.da16) {
        const unsigned pk = swn_pack_bf16(dz * gz, dc * gc);
        unsigned short* rz = a.da16 + ((size_t)b * H2 + o) * a.da16_pitch + t;
        unsigned short* rc = a.da16 + ((size_t)b * H2 + H + o) * a.da16_pitch + t;
        rz[0] = (unsigned short)(pk & 0xffffu); rz[a.da16_odd + 1] = (unsigned short)(pk & 0xffffu);
        rc[0] = (unsigned short)(pk >> 16); rc[a.da16_odd + 1] = (unsigned short)(pk >> 16);
    }
    if (a.dgx16) {
        const unsigned pk = swn_pack_bf16(dz * sz, dc * sc);
        a.dgx16[(size_t)b * a.dgx16_sb + (size_t)o * a.da16_pitch + t] = (unsigned short)(pk & 0xffffu);
        a.dgx16[(size_t)b * a.dgx16_sb + (size_t)(H + o) * a.da16_pitch + t] = (unsigned short)(pk >> 16);
    } else {
        a.dgx[(size_t)b * a.dgx_sb + (size_t)o * a.Tp + t] = dz * sz;
        a.dgx[(size_t)b * a.dgx_sb + (size_t)(H + o) * a.Tp + t] = dc * sc;
    }
    if (KIND == SWN_KIND_SOFTMAX && g.audio_in && a.gwxa) {      // one-hot input column idx: d in_x.W[o][A0+idx] += dgx
        atomicAdd(a.gwxa + ((size_t)l * g.Q + idx) * H2 + o, dz * sz);
        atomicAdd(a.gwxa + ((size_t)l * g.Q + idx) * H2 + H + o, dc * sc);
    }
    a.dhs[hb + ((size_t)l * H + o) * a.Tp + t] = in.dhl + dh * z * im;   // highway path (through the input's dropout mask)
}

template <int KIND>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const GateBwd a) {
    const int t = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
    if (t >= a.Tp) return;
    gate_bwd_one<KIND, false>(a, b, t, o, gate_bwd_in(a, b, t, o), 0.f, 0.f, 0.f, 0.f);
}

// the same with the kept pre-activations (and the in_x rows of the dropout mode) in the G4 layout: a thread takes FOUR channels of
// its position, whose pre-activations are one 16-byte piece (read channel by channel they were strided 4-byte loads: 216 -> 278 us
// per layer at the run.sh geometry); everything else is read and written per channel, lanes along t, as above.  H % 4 == 0.
template <int KIND>
__global__ __launch_bounds__(256) void gate_bwd_g4_kernel(const GateBwd a) {
    const int t = blockIdx.x * 256 + threadIdx.x, o0 = 4 * blockIdx.y, b = blockIdx.z;
    if (t >= a.Tp) return;
    const int H = a.g.H, H2 = 2 * a.g.H;
    const swn_f32x4 pz = *reinterpret_cast<const swn_f32x4*>(a.a_in + swn_g4(H2, a.Tp, b, o0, t));
    const swn_f32x4 pc = *reinterpret_cast<const swn_f32x4*>(a.a_in + swn_g4(H2, a.Tp, b, H + o0, t));
    swn_f32x4 xz = {0.f, 0.f, 0.f, 0.f}, xc = {0.f, 0.f, 0.f, 0.f};
    if (a.gx) {
        xz = *reinterpret_cast<const swn_f32x4*>(a.gx + swn_g4(a.g.L * H2, a.Tp, b, a.l * H2 + o0, t));
        xc = *reinterpret_cast<const swn_f32x4*>(a.gx + swn_g4(a.g.L * H2, a.Tp, b, a.l * H2 + H + o0, t));
    }
    GbIn in[4];                                                // every load of the four channels ahead of the first store (they may alias)
#pragma unroll
    for (int r = 0; r < 4; ++r) in[r] = gate_bwd_in(a, b, t, o0 + r);
#pragma unroll
    for (int r = 0; r < 4; ++r) gate_bwd_one<KIND, true>(a, b, t, o0 + r, in[r], pz[r], pc[r], xz[r], xc[r]);
}

// zeroes `n` floats at the start of each of gridDim.y rows `stride` floats apart
__global__ __launch_bounds__(256) void zero_rows_kernel(float* __restrict__ p, const size_t stride, const size_t n) {
    float* row = p + (size_t)blockIdx.y * stride;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) row[i] = 0.f;
}

// cond_bwd_kernel adds its upsampler-tap sums into one of SWN_WUP_COPIES zeroed copies (chosen by workgroup index) instead of
// straight into g w_up: ~800 workgroups per launch finish together, and 800 same-address float atomics per tap (~25 ns each
// at the L2) were a 20 us serial tail of a 66 us kernel.  wup_fold_kernel adds the copies into the packed gradient once, after
// the last layer.
__global__ __launch_bounds__(256) void wup_fold_kernel(const float* __restrict__ part, float* __restrict__ gwup, const int U) {
    const int jj = threadIdx.x;
    if (jj >= U) return;
    float v = 0.f;
#pragma unroll
    for (int c = 0; c < SWN_WUP_COPIES; ++c) v += part[c * 256 + jj];
    atomicAdd(gwup + jj, v);
}

// ---- hoisted conditioning backward, two orientations of the same (frame x tap) product:
//      threads 0..63    thread = in_x row o2:  dcond[b][f][(l*seg+s)*2H+o2] = sum_jj w_up[jj] * dgx[o2][t] ; gbx
//      threads 64..255  thread = upsampler tap jj:  gwup[jj] += sum_{s,o2} dgx[o2][t] * cond[b][f][(l*seg+s)*2H+o2]
//      with t = f*U + jj - s - coff  (the positions whose conditioning comes from frame f, tap jj)
template <int NCH>
__global__ __launch_bounds__(256) void cond_bwd_kernel(const GateBwd a, float* __restrict__ dcond, float* __restrict__ gbx,
                                                       float* __restrict__ gwup, const int FR) {
    // workgroup = (64 in_x rows, FR consecutive frames, utterance b).  Per frame the U + seg - 1 positions it touches are
    // staged through LDS with coalesced loads (lanes along t; it used to be one strided stream per thread, thrashing L1),
    // then one thread per row does its U x seg multiply-adds out of LDS (row pitch odd: conflict-free).
    // The sums over frames (g b_inx per row, g w_up per tap) stay in registers across the FR frames and leave as ONE
    // atomic each: with one frame per workgroup every w_up tap took B x Tf x 2H/64 = 7 200 same-address atomics per layer
    // at REF6, which - not the 200 MB of dgx - was what the 182 us of this kernel were.
    extern __shared__ float tile_mem[];
    const int pitch = (a.g.U + a.g.seg - 1) | 1;
    auto tile = [&](int r, int c) -> float& { return tile_mem[r * pitch + c]; };
    __shared__ __attribute__((aligned(16))) float wus[256];
    __shared__ __attribute__((aligned(16))) float cs[10][64];                 // seg <= 10 (swn_make_geom)
    const SwnGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.z;
    const int H2 = 2 * g.H, l = a.l, seg = g.seg, U = g.U;
    const int ncol = U + seg - 1;
    wus[tid] = tid < U ? a.P[a.y.wup + tid] : 0.f;
    // (mixed-precision chain: gate_bwd left d gx as bf16 rows of da16_pitch elements - half the bytes of this kernel's one big read)
    const bool d16 = a.dgx16 != nullptr;
    const __amdgpu_buffer_rsrc_t rD = d16 ? rsrc_of(a.dgx16 + (size_t)b * a.dgx16_sb) : rsrc_of(a.dgx + (size_t)b * a.dgx_sb);
    const int o2 = blockIdx.x * 64 + tid;
    float bsum = 0.f, wacc[2] = {0.f, 0.f};      // tid < 64: row sum ; tid >= 64: taps tid - 64 and tid + 128
    const int f1 = (blockIdx.y + 1) * FR < a.Tf ? (blockIdx.y + 1) * FR : a.Tf;
    // One frame ahead in registers: the loads of frame f + 1 (its dgx tile: NCH chunks of 64 columns x 16 rows per wave, and its
    // in_x products) are issued right after the barrier that publishes frame f, so their round trip runs under frame f's
    // multiply-adds instead of in front of frame f + 1's (a workgroup used to spend ~9 us per frame, three dependent round trips).
    const __amdgpu_buffer_rsrc_t rC = rsrc_of(a.cond + (size_t)b * a.Tf * g.N);
    float v[NCH][16], cpre[3];
    auto fetch = [&](const int f) {
        const int tbeg = f * U - (seg - 1) - a.coff;                              // tile column c <-> position tbeg + c
#pragma unroll
        for (int k = 0; k < 3; ++k) {                                             // seg <= 10: at most 640 products
            const int e = tid + 256 * k, sx = e >> 6, r2 = blockIdx.x * 64 + (e & 63);
            cpre[k] = bld1(rC, (e < seg * 64 && r2 < H2) ? (unsigned)(((size_t)f * g.N + (size_t)(l * seg + sx) * H2 + r2) * 4) : SWN_OOB);
        }
        if (d16) {
            // bf16 rows: a lane takes the dword (positions te, te + 1) with te = (tbeg & ~1) + 2 lane + 128 k - aligned whatever the
            // parity of the frame's first position; (NCH + 1) / 2 chunks of 128 columns cover the tile (2-byte loads: 58 -> 231 us)
            const int tal = tbeg & ~1;
#pragma unroll
            for (int k = 0; k < (NCH + 1) / 2; ++k) {
                const int te = tal + 2 * lane + 128 * k;
                const bool tok = te - tbeg < ncol && te + 1 >= 0 && te < a.Tp;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int r2 = blockIdx.x * 64 + w + 4 * i;
                    v[k][i] = bld1(rD, (tok && te >= 0 && r2 < H2) ? ((unsigned)r2 * (unsigned)a.da16_pitch + (unsigned)te) * 2u : SWN_OOB);
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = lane + 64 * k, t = tbeg + c;
            const bool tok = c < ncol && t >= 0 && t < a.Tp;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int r2 = blockIdx.x * 64 + w + 4 * i;
                v[k][i] = bld1(rD, (tok && r2 < H2) ? ((unsigned)r2 * (unsigned)a.Tp + (unsigned)t) * 4u : SWN_OOB);
            }
        }
    };
    fetch(blockIdx.y * FR);
    for (int f = blockIdx.y * FR; f < f1; ++f) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int e = tid + 256 * k; if (e < seg * 64) cs[e >> 6][e & 63] = cpre[k]; }
        if (d16) {
            const int tbeg = f * U - (seg - 1) - a.coff, tal = tbeg & ~1;
#pragma unroll
            for (int k = 0; k < (NCH + 1) / 2; ++k) {
                const int te = tal + 2 * lane + 128 * k, c0 = te - tbeg;         // columns c0 (may be -1) and c0 + 1
                const bool k0 = c0 >= 0 && c0 < ncol && te >= 0 && te < a.Tp, k1 = c0 + 1 < ncol && te + 1 >= 0 && te + 1 < a.Tp;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const unsigned u = __float_as_uint(v[k][i]);
                    if (c0 >= 0 && c0 < ncol) tile(w + 4 * i, c0) = k0 ? __uint_as_float(u << 16) : 0.f;
                    if (c0 + 1 < ncol) tile(w + 4 * i, c0 + 1) = k1 ? __uint_as_float(u & 0xffff0000u) : 0.f;
                }
            }
        } else {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int c = lane + 64 * k;
            if (c < ncol) {
#pragma unroll
                for (int i = 0; i < 16; ++i) tile(w + 4 * i, c) = v[k][i];
            }
        }
        }
        __syncthreads();
        if (f + 1 < f1) fetch(f + 1);
        __builtin_amdgcn_sched_barrier(0);                                        // keep the requests above the LDS loops
        // Both orientations read LDS in batches of eight independent loads: written as plain accumulation loops the compiler
        // kept one read in flight (read, wait, fma), and the 110 dependent LDS round trips of wave 0 - ~6 us per frame - were
        // the kernel.
        if (tid >= 64) {
            // the other orientation of the same tile (was a second pass over dgx, wup_bwd_kernel): thread = upsampler tap jj,
            // gwup[jj] += sum_{s, rows} dgx[row][t(jj, s)] * cond[f][(l*seg+s)*2H + row]   (columns: conflict-free, pitch odd)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int jj = tid - 64 + 192 * q;
                if (jj < U) {
                    float acc = 0.f;
                    for (int sx = 0; sx < seg; ++sx) {
                        const float* col = tile_mem + jj + (seg - 1) - sx;
                        for (int r = 0; r < 64; r += 8) {
                            float d[8];
#pragma unroll
                            for (int e = 0; e < 8; ++e) d[e] = col[(r + e) * pitch];
                            const swn_fl4 c0 = *reinterpret_cast<const swn_fl4*>(&cs[sx][r]);
                            const swn_fl4 c1 = *reinterpret_cast<const swn_fl4*>(&cs[sx][r + 4]);
                            acc = fmaf(d[0], c0.x, acc); acc = fmaf(d[1], c0.y, acc); acc = fmaf(d[2], c0.z, acc); acc = fmaf(d[3], c0.w, acc);
                            acc = fmaf(d[4], c1.x, acc); acc = fmaf(d[5], c1.y, acc); acc = fmaf(d[6], c1.z, acc); acc = fmaf(d[7], c1.w, acc);
                        }
                    }
                    wacc[q] += acc;
                }
            }
        } else if (o2 < H2) {
            for (int s = 0; s < seg; ++s) {
                const float* row = tile_mem + tid * pitch + (seg - 1) - s;       // t = f*U + jj - s - coff
                float dsum = 0.f, psum = 0.f;
                int jj = 0;
                for (; jj + 8 <= U; jj += 8) {
                    float d[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[e] = row[jj + e];
                    const swn_fl4 w0 = *reinterpret_cast<const swn_fl4*>(&wus[jj]);
                    const swn_fl4 w1 = *reinterpret_cast<const swn_fl4*>(&wus[jj + 4]);
                    dsum = fmaf(w0.x, d[0], dsum); dsum = fmaf(w0.y, d[1], dsum); dsum = fmaf(w0.z, d[2], dsum); dsum = fmaf(w0.w, d[3], dsum);
                    dsum = fmaf(w1.x, d[4], dsum); dsum = fmaf(w1.y, d[5], dsum); dsum = fmaf(w1.z, d[6], dsum); dsum = fmaf(w1.w, d[7], dsum);
#pragma unroll
                    for (int e = 0; e < 8; ++e) psum += d[e];
                }
                for (; jj < U; ++jj) { const float d = row[jj]; dsum = fmaf(wus[jj], d, dsum); psum += d; }
                if (s == 0) bsum += psum;                                        // every position belongs to exactly one (f, jj) at s = 0
                dcond[((size_t)b * a.Tf + f) * g.N + (size_t)(l * seg + s) * H2 + o2] = dsum;
            }
        }
        __syncthreads();                                                   // the tile is free for the next frame
    }
    if (tid >= 64) {
        float* part = gwup + 256 * ((blockIdx.x + 3 * blockIdx.y + 5 * blockIdx.z) & (SWN_WUP_COPIES - 1));   // gwup: the zeroed copies
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int jj = tid - 64 + 192 * q; if (jj < U) atomicAdd(part + jj, wacc[q]); }
    } else if (o2 < H2) {
        atomicAdd(gbx + (size_t)l * H2 + o2, bsum);
    }
}


// ---- input layer backward: dh0 -> gcb, gcv, gcc (laplace) | gct (softmax)
template <int KIND>
__global__ __launch_bounds__(256) void input_bwd_kernel(const GateBwd a, float* __restrict__ gP) {
    const SwnGeom& g = a.g;
    const int o = blockIdx.x, b = blockIdx.y, K = g.K, H = g.H;
    const float* P = a.P;
    const float* dh0 = a.dhs + ((size_t)b * (g.L + 1)) * H * a.Tp + (size_t)o * a.Tp;
    float scb = 0.f, sv[8] = {0.f}, sc[8] = {0.f};
    // the channel's constants stay in registers (they were re-read from memory for every position: 433 us at REF6)
    float kv[8], kc[8];
    const float kb = P[a.y.cb + o];
#pragma unroll
    for (int k = 0; k < 8; ++k) { kv[k] = (KIND == SWN_KIND_LAPLACE && k < K) ? P[a.y.cv + (size_t)k * H + o] : 0.f;
                                  kc[k] = (KIND == SWN_KIND_LAPLACE && k < K) ? P[a.y.cc + (size_t)k * H + o] : 0.f; }
    if (KIND == SWN_KIND_LAPLACE) {
        // four positions per thread and trip, all their loads requested before the first is used (one position per trip was a
        // chain of 65 dependent HBM round trips per thread: 84 us for 101 MB); out-of-range = the zero of a buffer load
        const __amdgpu_buffer_rsrc_t rA = rsrc_of(reinterpret_cast<const float*>(a.audio) + (size_t)b * (a.Tp + g.seg - 1));
        const __amdgpu_buffer_rsrc_t rD = rsrc_of(dh0);
        for (int t0 = threadIdx.x; t0 < a.Tp; t0 += 1024) {
            float dh[4], x[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 256 * u, ai = t + g.seg - 1;
                const bool ok = t < a.Tp;
                dh[u] = bld1(rD, ok ? (unsigned)t * 4u : SWN_OOB);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const int r = ai - (K - 1 - k); x[u][k] = bld1(rA, (ok && k < K && r >= 0) ? (unsigned)r * 4u : SWN_OOB); }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ai = t0 + 256 * u + g.seg - 1;
                float pre = kb;
#pragma unroll
                for (int k = 0; k < 8; ++k) { const int r = ai - (K - 1 - k); if (k < K && r >= 0) pre += fmaf(kv[k], x[u][k], kc[k]); }
                const float d = dh[u] / ((1.f + fabsf(pre)) * (1.f + fabsf(pre)));      // dh = 0 past the end
                scb += d;
#pragma unroll
                for (int k = 0; k < 8; ++k) { const int r = ai - (K - 1 - k); if (k < K && r >= 0) { sv[k] = fmaf(d, x[u][k], sv[k]); sc[k] += d; } }
            }
        }
    }
    for (int t = threadIdx.x; KIND != SWN_KIND_LAPLACE && t < a.Tp; t += 256) {
        float pre = kb;
        {
            const int* au = reinterpret_cast<const int*>(a.audio) + (size_t)b * a.Tp;
            int idxs[8];
            for (int k = 0; k < K && k < 8; ++k) {
                const int r = t - (K - 1 - k);
                idxs[k] = -1;
                if (r >= 0) { int idx = au[r] % g.Q; idx = idx < 0 ? idx + g.Q : idx; idxs[k] = idx; pre += P[a.y.ct + ((size_t)k * g.Q + idx) * H + o]; }
            }
            const float d = dh0[t] / ((1.f + fabsf(pre)) * (1.f + fabsf(pre)));
            scb += d;
            for (int k = 0; k < K && k < 8; ++k) if (idxs[k] >= 0) atomicAdd(gP + a.y.ct + ((size_t)k * g.Q + idxs[k]) * H + o, d);
        }
    }
    // block reduction, then one atomic per value and block (every thread used to add its own partial: 256 x H x B
    // atomics onto H addresses per section)
    __shared__ float red[4][17];
    auto wave_sum = [](float v) {
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_down(v, sft);
        return v;
    };
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    scb = wave_sum(scb);
    if (lane == 0) red[wv][0] = scb;
    if (KIND == SWN_KIND_LAPLACE) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float a1 = wave_sum(sv[k]), a2 = wave_sum(sc[k]);
            if (lane == 0) { red[wv][1 + k] = a1; red[wv][9 + k] = a2; }
        }
    }
    __syncthreads();
    if (threadIdx.x < 17) {
        const int i = threadIdx.x;
        const float v = red[0][i] + red[1][i] + red[2][i] + red[3][i];
        if (i == 0) atomicAdd(gP + a.y.cb + o, v);
        else if (KIND == SWN_KIND_LAPLACE) {
            const int k = (i - 1) & 7;
            if (k < K) atomicAdd(gP + (i <= 8 ? a.y.cv : a.y.cc) + (size_t)k * H + o, v);
        }
    }
}

// ---- Laplace head backward: grads wrt (mu, b, logb, a) time-major -> grad wrt raw (B, NO, Tp)
__global__ __launch_bounds__(256) void laplace_head_bwd_kernel(const float* __restrict__ raw, int Tp, int seg, int lpc,
                                                               const float* gmu, const float* gb, const float* glogb,
                                                               const float* ga, const float* gbc, const float* glc,
                                                               float* __restrict__ graw) {
    const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (t >= Tp) return;
    const int NO = 2 * seg + lpc;
    for (int j = 0; j < seg; ++j) {
        const size_t o = ((size_t)b * Tp + t) * seg + j;
        graw[((size_t)b * NO + j) * Tp + t] = gmu ? gmu[o] : 0.f;
        const float y = raw[((size_t)b * NO + seg + j) * Tp + t];
        const float s = sigm(y), bb = s;                    // b = exp(logsigmoid(y)) = sigmoid(y)
        float dlog = (glogb ? glogb[o] : 0.f) + (gb ? gb[o] * bb : 0.f);           // d/dlogb
        // clipped copies (logb floored at -14.1621): gradient flows only where the floor is inactive
        const float lb = fminf(y, 0.f) - log1pf(expf(-fabsf(y)));
        if (lb >= -14.162084148244246758816564788835f) dlog += (glc ? glc[o] : 0.f) + (gbc ? gbc[o] * bb : 0.f);
        graw[((size_t)b * NO + seg + j) * Tp + t] = dlog * (1.f - s);            // dlogsigmoid/dy = 1 - sigmoid(y)
    }
    for (int k = 0; k < lpc; ++k)
        graw[((size_t)b * NO + 2 * seg + k) * Tp + t] = ga ? ga[((size_t)b * Tp + t) * lpc + k] : 0.f;
}

size_t r64(size_t x) { return (x + 63) & ~(size_t)63; }

// arithmetic mode of the training call in progress on this thread (SwnModeScope, csrc/swn_geom.hpp)
thread_local int t_call_mode = SWN_PRECISION_FP32;
inline bool mode_bf16() { return t_call_mode == SWN_PRECISION_BF16; }

void launch_time(const TimeGemm& g, int B, hipStream_t st) {
    const dim3 grid((g.T + 63) / 64, (g.M + 63) / 64, B);
    if (mode_bf16()) {
        if (!g.xmul && g.x_st == 1 && g.KC % 32 == 0) {
            TimeGemm h = g;
            h.nb = B;
            const int chunk = (((g.T + 127) / 128) * B + 7) / 8;
            const dim3 big((unsigned)(8 * chunk * ((g.M + 127) / 128)));
            if (g.A16 && g.X16 && !g.mask) {
                // 96-row tiles where they cover M with fewer padded rows than 128-row tiles (M = 192: 2 x 96 against 128 + 64 of 128)
                const int m96 = (g.M + 95) / 96, m128 = (g.M + 127) / 128;
                if (m96 * 96 < m128 * 128) hipLaunchKernelGGL(time_gemm_b16_kernel<3>, dim3((unsigned)(8 * chunk * m96)), dim3(256), 0, st, h);
                else hipLaunchKernelGGL(time_gemm_b16_kernel<4>, big, dim3(256), 0, st, h);
            }
            else if (g.a_sc == 1) hipLaunchKernelGGL(time_gemm_bf16t_kernel<true>, big, dim3(256), 0, st, h);
            else hipLaunchKernelGGL(time_gemm_bf16t_kernel<false>, big, dim3(256), 0, st, h);
            return;
        }
        // frame-rate GEMMs (a few hundred columns, k in the hundreds to thousands): too few tiles to fill the chip and a long
        // latency-bound k loop per tile -> cut k over several workgroups
        const int wgs = (int)(grid.x * grid.y) * B, nk = (g.taps * g.KC + 31) / 32;
        if (!g.xmul && !g.mask && !g.ymul && !g.bias && !g.accumulate && wgs < 512 && nk >= 8 && g.y_sm == g.T &&
            g.y_sb == (long)g.M * g.T) {
            int ks = 1024 / wgs; if (ks > nk / 3) ks = nk / 3; if (ks > 8) ks = 8;
            if (ks > 1 && hipMemsetAsync(g.Y, 0, (size_t)B * g.M * g.T * sizeof(float), st) == hipSuccess) {
                TimeGemm h = g; h.ksplit = ks;
                hipLaunchKernelGGL(time_gemm_bf16_kernel<false>, dim3(grid.x, grid.y, B * ks), dim3(256), 0, st, h);
                return;
            }
        }
        if (g.xmul) hipLaunchKernelGGL(time_gemm_bf16_kernel<true>, grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL(time_gemm_bf16_kernel<false>, grid, dim3(256), 0, st, g);
        return;
    }
    TimeGemm h = g;
    h.nb = B;
    const dim3 lin((unsigned)(8 * ((grid.x * B + 7) / 8) * grid.y));
    if (g.xmul) { hipLaunchKernelGGL(time_gemm_kernel<true>, lin, dim3(256), 0, st, h); return; }
    hipLaunchKernelGGL(time_gemm_kernel<false>, lin, dim3(256), 0, st, h);
}
void launch_reduce(ReduceGemm g, int B, hipStream_t st) {
    g.TS = 512;
    const int nseg = (g.T + g.TS - 1) / g.TS;
    const dim3 grid((g.M + 63) / 64, (g.taps * g.KC + 63) / 64, B * nseg);
    if (mode_bf16()) {
        if (!g.qmul && g.p_st == 1 && g.q_st == 1 && g.T >= 256) {
            // 128 x 128 tiles; time segments sized for a target number of workgroups
            const int mt = (g.M + 127) / 128, nt = (g.taps * g.KC + 127) / 128;
            // (a single-tile gradient pays 16 K float atomics per workgroup for very little MFMA work: one workgroup per CU
            //  there; with many tiles per segment the atomics amortise and more workgroups hide the loads - measured both ways)
            const int target = mt * nt >= 4 ? 1024 : 256;
            int want = (target + mt * nt * B - 1) / (mt * nt * B);
            // a time segment is at least 256 positions, 1 024 where the gradient has many tiles: at the training drivers' own
            // chunk (one utterance, 8 800 positions) 31 segments of 288 positions meant 31 atomics per gradient element and nine
            // k-steps of work per workgroup (75 us per call, atomics-bound)
            const int seg_min = mt * nt >= 16 ? 1024 : 256;
            const int most = (g.T + seg_min - 1) / seg_min;
            if (want > most) want = most;
            if (want < 1) want = 1;
            g.TS = (((g.T + want - 1) / want) + 31) & ~31;
            const int ns = (g.T + g.TS - 1) / g.TS;
            {
                g.nsegtot = B * ns;
                const dim3 rgrid((unsigned)(mt * nt * ((g.nsegtot + 7) / 8) * 8));
                if (g.P16 && g.Q16) hipLaunchKernelGGL((reduce_gemm_bf16s_kernel<true, true>), rgrid, dim3(256), 0, st, g);
                else if (g.P16) hipLaunchKernelGGL(reduce_gemm_bf16s_kernel<true>, rgrid, dim3(256), 0, st, g);
                else hipLaunchKernelGGL(reduce_gemm_bf16s_kernel<false>, rgrid, dim3(256), 0, st, g);
            }
            return;
        }
        if (g.qmul) hipLaunchKernelGGL(reduce_gemm_bf16_kernel<true>, grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL(reduce_gemm_bf16_kernel<false>, grid, dim3(256), 0, st, g);
        return;
    }
    g.nsegtot = B * nseg;
    hipLaunchKernelGGL(reduce_gemm_kernel, dim3((unsigned)(grid.x * grid.y * ((g.nsegtot + 7) / 8) * 8)), dim3(256), 0, st, g);
}

}  // namespace

int  swn_call_mode() { return t_call_mode; }
void swn_call_mode_set(int mode) { t_call_mode = mode; }

static size_t chain_floats(const SwnGeom& g, int batch, int n_frames);
extern "C" size_t swn_backward_work_floats(const swn_net_desc* d, int batch, int n_frames) {
    SwnGeom g; if (swn_make_geom(d, &g) < 0 || batch < 1 || n_frames < 1) return 0;
    const long T = (long)n_frames * g.U;
    const long Tp = g.kind == SWN_KIND_SOFTMAX ? T - 1 : T - 2 * g.seg + 1;
    if (Tp < 1) return 0;
    return chain_floats(g, batch, n_frames);
}
static size_t chain_floats(const SwnGeom& g, int batch, int n_frames) {
    const long T = (long)n_frames * g.U;
    const long Tp = g.kind == SWN_KIND_SOFTMAX ? T - 1 : T - 2 * g.seg + 1;
    size_t fw = (size_t)g.n_aux; for (int i = 0; i < g.auxl; ++i) fw += g.aux_cout[i];
    return r64((size_t)batch * g.O1 * Tp) + r64((size_t)batch * g.S * Tp) + r64((size_t)batch * (g.L + 1) * g.H * Tp) +
           2 * r64((size_t)batch * 2 * g.H * Tp) + r64((size_t)batch * n_frames * g.N) + r64(fw * batch * n_frames) +
           (size_t)SWN_WUP_COPIES * 256 +     // partial upsampler-tap gradients of cond_bwd_kernel
           r64((size_t)batch * 2 * g.H * ((Tp + 2 + 31) & ~31L)) +     // two bf16 copies of a layer's da (mixed-precision GEMM operands)
           r64((size_t)g.L * g.K * g.H * 2 * g.H / 2) +               // transposed bf16 copy of the layer matrices (data gradients)
           r64((size_t)batch * g.H * ((Tp + 2 + 31) & ~31L));          // two bf16 copies of a layer's input (weight gradients' Q operand)
}
// floats of the dropout chain's scratch (swn_backward_drop without the kept d gx): the chain's + d xm + one masked layer input
static size_t drop_chain_floats(const SwnGeom& g, int batch, int n_frames) {
    const long T = (long)n_frames * g.U;
    const long Tp = g.kind == SWN_KIND_SOFTMAX ? T - 1 : T - 2 * g.seg + 1;
    return chain_floats(g, batch, n_frames) + r64((size_t)batch * g.A0 * (T - (g.kind == SWN_KIND_SOFTMAX ? 1 : g.seg))) + r64((size_t)batch * g.H * Tp);
}

namespace {

// dropout mode, conditioning side: dx = dxm * drop_x ; dC[b][c][f] = sum_j dx[c][f*U+j-coff] w_up[j] ;
// g w_up[j] += sum dx * C ; g b_up += sum dx          (backward of xm_fwd_kernel, csrc/swn_stack.hip)
// Workgroup = (FR frames, 32 channels, utterance); a wave takes channels w, w+4, ..., its lanes the taps j = lane + 64 q.
// The sums over channels and frames (g w_up per tap, g b_up) stay in registers and leave as one atomic per tap and
// workgroup.  (One workgroup per (frame, channel, utterance) with one atomic per thread: 32 M atomics onto 110 addresses
// at the run.sh geometry - 14 ms of a 47 ms step.)
// dropout mode: the masked input of a layer, once, as a plain tensor - the contraction kernels then run their fast forms
// (the masked-operand forms of the 64 x 64 kernels cost 2.6 + 1.7 ms per masked layer at the run.sh geometry, the fast forms 0.3 each)
__global__ __launch_bounds__(256) void mask_mul_kernel(const float* __restrict__ x, long x_sb, const float* __restrict__ m,
                                                       float* __restrict__ out, long n_per_b) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    const int b = blockIdx.y;
    if (i >= n_per_b) return;
    const float* xp = x + (size_t)b * x_sb + i;
    const float* mp = m + (size_t)b * n_per_b + i;
    float* op = out + (size_t)b * n_per_b + i;
#pragma unroll
    for (int e = 0; e < 4; ++e) if (i + e < n_per_b) op[e] = xp[e] * mp[e];
}

constexpr int XM_CC = 32;
__global__ __launch_bounds__(256) void xm_bwd_kernel(const float* __restrict__ dxm, const float* __restrict__ drop_x,
                                                     const float* __restrict__ C, const float* __restrict__ P, size_t wup,
                                                     float* __restrict__ dC, float* __restrict__ gwup, float* __restrict__ gbup,
                                                     int A0, int Tf, int U, int coff, int Tx, int FR) {
    __shared__ float red[4][256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.z;
    const int c0 = blockIdx.y * XM_CC;
    float wu[4], gw[4] = {0.f, 0.f, 0.f, 0.f}, gb = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) wu[q] = lane + 64 * q < U ? P[wup + lane + 64 * q] : 0.f;
    const __amdgpu_buffer_rsrc_t rX = rsrc_of(dxm + (size_t)b * A0 * Tx), rM = rsrc_of(drop_x + (size_t)b * A0 * Tx);
    const int f1 = (blockIdx.x + 1) * FR < Tf ? (blockIdx.x + 1) * FR : Tf;
    for (int f = blockIdx.x * FR; f < f1; ++f) {
        unsigned off[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lane + 64 * q, u = f * U + j - coff;
            off[q] = (j < U && u >= 0 && u < Tx) ? (unsigned)(u * 4) : SWN_OOB;
        }
        for (int ci = w; ci < XM_CC; ci += 8) {                  // two channels per pass: 16 loads in flight
            float x[2][4], m[2][4], cv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = c0 + ci + 4 * h;
                const unsigned ro = c < A0 ? (unsigned)((size_t)c * Tx * 4) : SWN_OOB;
                cv[h] = c < A0 ? C[((size_t)b * A0 + c) * Tf + f] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned o = ((off[q] | ro) & SWN_OOB) ? SWN_OOB : off[q] + ro;
                    x[h][q] = bld1(rX, o); m[h][q] = bld1(rM, o);
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = c0 + ci + 4 * h;
                float part = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float dx = x[h][q] * m[h][q];
                    gw[q] = fmaf(dx, cv[h], gw[q]); part = fmaf(dx, wu[q], part); gb += dx;
                }
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) part += __shfl_xor(part, sft);
                if (lane == 0 && c < A0) dC[((size_t)b * A0 + c) * Tf + f] = part;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[w][lane + 64 * q] = gw[q];
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) gb += __shfl_xor(gb, sft);
    __syncthreads();
    if (tid < U) atomicAdd(gwup + tid, red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
    if (lane == 0 && gb != 0.f) atomicAdd(gbup, gb);
}

// The same backward with d xm as bf16 TIME-MAJOR rows [b][u][A0x] (the fused BL6 dropout path: what its in_x data-gradient
// GEMM stores, csrc/swn_bwd_bl6.hip).  Workgroup = (FR frames, 64 channels, utterance); a frame's U x 64 tile is staged in LDS
// (pitch 33 dwords: the lanes of a wave then read one channel of 64 positions without bank conflicts), the mask is read
// along its contiguous axis as above.
constexpr int XM16_CC = 64;
__global__ __launch_bounds__(256) void xm_bwd16_kernel(const unsigned short* __restrict__ dxm16, const float* __restrict__ drop_x,
                                                       const float* __restrict__ C, const float* __restrict__ P, size_t wup,
                                                       float* __restrict__ dC, float* __restrict__ gwup, float* __restrict__ gbup,
                                                       int A0, int A0x, int Tf, int U, int coff, int Tx, int FR) {
    __shared__ unsigned tile[256 * 33];                          // U <= 256 positions x 32 dwords (64 bf16) + 1 pad
    __shared__ float red[4][256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, b = blockIdx.z;
    const int c0 = blockIdx.y * XM16_CC;
    float wu[4], gw[4] = {0.f, 0.f, 0.f, 0.f}, gb = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) wu[q] = lane + 64 * q < U ? P[wup + lane + 64 * q] : 0.f;
    const __amdgpu_buffer_rsrc_t rM = rsrc_of(drop_x + (size_t)b * A0 * Tx);
    const int f1 = (blockIdx.x + 1) * FR < Tf ? (blockIdx.x + 1) * FR : Tf;
    const int nq = (U + 63) / 64;
    for (int f = blockIdx.x * FR; f < f1; ++f) {
        __syncthreads();                                         // the previous frame's tile has been read
        for (int e = tid; e < U * 8; e += 256) {                 // (position, 16-byte chunk of its 128-byte segment)
            const int j = e >> 3, ch = e & 7, u = f * U + j - coff;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (u >= 0 && u < Tx && c0 + 8 * ch < A0x) v = *reinterpret_cast<const uint4*>(dxm16 + ((size_t)b * Tx + u) * A0x + c0 + 8 * ch);
            unsigned* d = tile + j * 33 + 4 * ch;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        unsigned off[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = lane + 64 * q, u = f * U + j - coff;
            off[q] = (j < U && u >= 0 && u < Tx) ? (unsigned)(u * 4) : SWN_OOB;
        }
        // a wave's 16 channels (w, w + 4, ...) in two passes of eight: all mask loads of a pass in flight together (two per
        // pass, as in the fp32 form, left a frame eight dependent round trips long)
        for (int half = 0; half < 2; ++half) {
            float m[8][4], cv[8];
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const int c = c0 + w + 4 * (8 * half + h);
                const unsigned ro = c < A0 ? (unsigned)((size_t)c * Tx * 4) : SWN_OOB;
                cv[h] = c < A0 ? C[((size_t)b * A0 + c) * Tf + f] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) m[h][q] = q < nq ? bld1(rM, ((off[q] | ro) & SWN_OOB) ? SWN_OOB : off[q] + ro) : 0.f;
            }
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const int cl = w + 4 * (8 * half + h), c = c0 + cl;
                float part = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (q >= nq) break;
                    const int j = lane + 64 * q;
                    const unsigned wd = j < U ? tile[j * 33 + (cl >> 1)] : 0u;
                    const float x = __builtin_bit_cast(float, (cl & 1) ? (wd & 0xffff0000u) : (wd << 16));
                    const float dx = x * m[h][q];
                    gw[q] = fmaf(dx, cv[h], gw[q]); part = fmaf(dx, wu[q], part); gb += dx;
                }
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) part += __shfl_xor(part, sft);
                if (lane == 0 && c < A0) dC[((size_t)b * A0 + c) * Tf + f] = part;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[w][lane + 64 * q] = gw[q];
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) gb += __shfl_xor(gb, sft);
    __syncthreads();
    if (tid < U) atomicAdd(gwup + tid, red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
    if (lane == 0 && gb != 0.f) atomicAdd(gbup, gb);
}

}  // namespace

// relu(skip) and relu(out_1) from fp32 hidden states already in `work` (layout of swn_forward_work_floats) through the
// contraction kernels of this file, i.e. in the arithmetic mode of the call (SwnModeScope): the tail of the training
// forward when the hidden states come from the BL6-class bf16 layer kernels (swn_bf16_work_to_f32).
int swn_train_head_acts(const SwnGeom& g, const float* packed, float* work, int B, long Tp, hipStream_t st) {
    if (g.Hp != g.H) return SWN_E_UNSUPPORTED;
    SwnLayout y; swn_make_layout(&g, &y);
    float* skipb = work + r64((size_t)B * (g.L + 1) * g.H * Tp);
    float* o1b = skipb + r64((size_t)B * g.S * Tp);
    (void)hipGetLastError();
    {   // skip[b][c][t] = relu(bsk[c] + sum_{l,i} Wsk[c][l*H+i] h_{l+1}[b][i][t])
        TimeGemm t = {packed + y.wsk, (long)g.L * g.Hp, 0, 1, work + (size_t)g.H * Tp, (long)(g.L + 1) * g.H * Tp, Tp, 1,
                      skipb, (long)g.S * Tp, Tp, nullptr, 0, 0, g.S, 1, g.L * g.H, (int)Tp, 1, 0, 1, 0};
        t.bias = packed + y.bsk; t.relu = 1;
        launch_time(t, B, st);
    }
    {   // o1 = relu(b1 + W1 skip)
        TimeGemm t = {packed + y.w1, g.Sp, 0, 1, skipb, (long)g.S * Tp, Tp, 1, o1b, (long)g.O1 * Tp, Tp, nullptr, 0, 0,
                      g.O1, 1, g.S, (int)Tp, 1, 0, 1, 0};
        t.bias = packed + y.b1; t.relu = 1;
        launch_time(t, B, st);
    }
    return swn_launch_status("swn_bf16_work_to_f32");
}

// Gated layers of the dropout-mode FORWARD in the mixed-precision mode: per layer the dilated conv as the bf16-operand time
// GEMM (the kernel the backward recomputes it with) on the masked input, then the gate / highway element-wise
// (cswnv_shift1.py:269-278 with the sample-rate in_x products `gx`).  The fp32 parity kernel tf_layer_kernel (exact-fp32
// MFMA, 1.6 ms per layer at the run.sh geometry) stays the forward of the fp32 mode.
namespace {
template <int KIND>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const GateBwd a, float* __restrict__ hs_out) {
    const SwnGeom& g = a.g;
    const int t = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y, b = blockIdx.z;
    if (t >= a.Tp) return;
    const int H = g.H, H2 = 2 * g.H, l = a.l;
    const float* P = a.P;
    const size_t hb = ((size_t)b * (g.L + 1)) * H * a.Tp;
    const float im = a.in_mul ? a.in_mul[((size_t)b * H + o) * a.Tp + t] : 1.f;
    const float hprev = a.hs[hb + ((size_t)l * H + o) * a.Tp + t] * im;
    const float* gr = a.gx + (((size_t)b * g.L + l) * H2) * a.Tp + t;
    float gz = gr[(size_t)o * a.Tp] + P[a.y.bxr + (size_t)l * H2 + o];
    float gc = gr[(size_t)(H + o) * a.Tp] + P[a.y.bxr + (size_t)l * H2 + H + o];
    if (KIND == SWN_KIND_SOFTMAX && g.audio_in) {
        int idx = reinterpret_cast<const int*>(a.audio)[(size_t)b * a.Tp + t] % g.Q; idx = idx < 0 ? idx + g.Q : idx;
        const float* wa = P + a.y.wxa + ((size_t)l * g.Q + idx) * H2;
        gz += wa[o]; gc += wa[H + o];
    }
    const float sz = a.a_da[((size_t)b * H2 + o) * a.Tp + t] + P[a.y.bd + (size_t)l * H2 + o];
    const float sc = a.a_da[((size_t)b * H2 + H + o) * a.Tp + t] + P[a.y.bd + (size_t)l * H2 + H + o];
    const float z = sigm(gz * sz), c = tanhf(gc * sc);
    hs_out[hb + ((size_t)(l + 1) * H + o) * a.Tp + t] = (1.f - z) * c + z * hprev;
}
}  // namespace

int swn_train_layers_forward_drop(const SwnGeom& g, const SwnLayout& y, const float* packed, const void* audio, const float* gx,
                                  const float* const* drop_h, float* hs, float* a_scr, float* hmask, int B, int n_frames, int Tp,
                                  hipStream_t st) {
    GateBwd ga{};
    ga.g = g; ga.y = y; ga.P = packed; ga.cond = nullptr; ga.audio = audio; ga.hs = hs; ga.dhs = nullptr; ga.a_da = a_scr;
    ga.dgx = nullptr; ga.B = B; ga.Tf = n_frames; ga.Tp = Tp; ga.coff = g.kind == SWN_KIND_SOFTMAX ? 1 : g.seg;
    ga.gx = gx; ga.gwxa = nullptr; ga.a_in = nullptr;
    const int H = g.H, H2 = 2 * g.H;
    const long hsb = (long)(g.L + 1) * H * Tp;
    for (int l = 0; l < g.L; ++l) {
        ga.l = l;
        const float* in_mul = l > 0 ? drop_h[l - 1] : nullptr;
        ga.in_mul = in_mul;
        const float* xin = hs + (size_t)l * H * Tp; long xin_sb = hsb;
        if (in_mul) {
            const long npb = (long)H * Tp;
            hipLaunchKernelGGL(mask_mul_kernel, dim3((unsigned)((npb / 4 + 255) / 256 + 1), B), dim3(256), 0, st, xin, hsb, in_mul, hmask, npb);
            xin = hmask; xin_sb = npb;
        }
        float* al = a_scr + (size_t)l * r64((size_t)B * H2 * Tp);          // this layer's slot (kept for the backward)
        ga.a_da = al;
        TimeGemm t = {packed + y.wd + (size_t)l * H2 * g.K * g.Hp, (long)g.K * g.Hp, g.Hp, 1, xin, xin_sb, Tp, 1, al, (long)H2 * Tp, Tp,
                      nullptr, 0, 0, H2, g.K, H, Tp, 1, g.K - 1, g.dil[l], 0};
        launch_time(t, B, st);
        const dim3 grid((Tp + 255) / 256, H, B);
        if (g.kind == SWN_KIND_LAPLACE) hipLaunchKernelGGL(gate_fwd_kernel<SWN_KIND_LAPLACE>, grid, dim3(256), 0, st, ga, hs);
        else hipLaunchKernelGGL(gate_fwd_kernel<SWN_KIND_SOFTMAX>, grid, dim3(256), 0, st, ga, hs);
    }
    return SWN_OK;
}

// bf16 copies of the in_x matrix [N][A0p] (seg == 1: row n = l*2H + o) for the bf16-copy contraction kernels (swn_drop_inx16):
// rows[n][c] over the A0x = 32-aligned conditioning rows (zeros from A0 on) for the forward product, and / or the transpose
// cols[c][n] for the data gradient
namespace {
__global__ __launch_bounds__(256) void wx16_kernel(const float* __restrict__ wx, const int N, const int A0, const int A0p, const int A0x,
                                                   unsigned short* __restrict__ rows, unsigned short* __restrict__ cols) {
    const size_t n_el = (size_t)N * A0x;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n_el; e += (size_t)gridDim.x * 256) {
        const int n = (int)(e / A0x), c = (int)(e - (size_t)n * A0x);
        const unsigned short v = c < A0 ? (unsigned short)(swn_pack_bf16(wx[(size_t)n * A0p + c], 0.f) & 0xffffu) : (unsigned short)0;
        if (rows) rows[e] = v;
        if (cols) cols[(size_t)c * N + n] = v;
    }
}
}  // namespace

int swn_train_inx_forward(const SwnGeom& g, const SwnLayout& y, const float* packed, const float* xm, float* gx,
                          int B, int Tx, int Tp, hipStream_t st, unsigned short* wx16, bool g4) {
    const int H2 = 2 * g.H;
    if (g.seg == 1) {                 // one launch over the L * 2H rows of the [N][A0p] matrix (row n = l*2H + o): gx is (B, L*2H, Tp)
        const int A0x = swn_a0x(&g);
        TimeGemm t = {packed + y.wx, g.A0p, 0, 1, xm, (long)A0x * Tx, Tx, 1, gx, (long)g.L * H2 * Tp, Tp, nullptr, 0, 0,
                      g.L * H2, 1, A0x, Tp, 1, 0, 1, 0};
        t.XT = Tx;
        if (wx16) {                   // swn_drop_inx16: xm holds bf16 rows of swn_pitch16(Tx) elements, wx16 receives the bf16 matrix
            hipLaunchKernelGGL(wx16_kernel, dim3(512), dim3(256), 0, st, packed + y.wx, g.L * H2, g.A0, g.A0p, A0x, wx16, nullptr);
            const long px = swn_pitch16(Tx);
            t.A16 = wx16; t.a16_sm = A0x; t.a16_stap = 0;
            t.X16 = reinterpret_cast<const unsigned short*>(xm); t.x16_sb = (long)A0x * px; t.x16_sc = px; t.x16_odd = 0;
            t.y_g4 = g4 ? 1 : 0;
        }
        launch_time(t, B, st);
        return SWN_OK;
    }
    for (int l = 0; l < g.L; ++l) {   // gx[b][l][o][t] = sum_{s,c} W[l][o][c*seg+s] xm[b][c][t+s]
        // k runs over the A0x = 32-aligned rows of xm: beyond A0 the activations are zero rows and the weights whatever
        // follows in the packed buffer (padding, then the next row: finite), so the products vanish
        const int A0x = swn_a0x(&g);
        TimeGemm t = {packed + y.wx + (size_t)l * g.seg * H2 * g.A0p, g.A0p, (long)H2 * g.A0p, 1,
                      xm, (long)A0x * Tx, Tx, 1, gx + (size_t)l * H2 * Tp, (long)g.L * H2 * Tp, Tp, nullptr, 0, 0,
                      H2, g.seg, A0x, Tp, 1, 0, 1, 0};
        t.XT = Tx;
        launch_time(t, B, st);
    }
    return SWN_OK;
}

// fused per-layer backward of the BL6 class in the mixed-precision mode (csrc/swn_bwd_bl6.hip)
bool swn_bl6_bwd_supported(const SwnGeom& g, int B, long Tp, int n_frames);
size_t swn_bl6_bwd_scratch_bytes(const SwnGeom& g, int B, long Tp);
size_t swn_bl6_bwd_drop_scratch_bytes(const SwnGeom& g, int B, long Tp);
int swn_bl6_bwd_stack(const SwnGeom& g, const SwnLayout& y, const float* packed, const float* cond, const float* audio,
                      const void* hs_bf16, const float* grad_out, float* dcond, float* gpacked, void* scratch, int B, int n_frames,
                      long Tp, hipStream_t st, const unsigned short* gx16 = nullptr, const unsigned short* xm16 = nullptr,
                      unsigned short* dxm16 = nullptr);

namespace {

// hs_bf16 != null: everything at sample rate runs through swn_bl6_bwd_stack (csrc/swn_bwd_bl6.hip; the caller has checked
// swn_bl6_bwd_supported); `work` is then [dcond | front-end gradients | swn_bl6_bwd_scratch_bytes] and fwd_work is not read
int backward_impl(const swn_net_desc* d, const float* packed, const float* aux, const float* cond,
                  const float* fe_work, const void* audio, const float* fwd_work, const float* hs_opt,
                  const float* drop_x, const float* const* drop_h,
                  const float* grad_out, int batch, int n_frames, float* work, float* gpacked, void* stream_,
                  const char* where, const void* hs_bf16 = nullptr, const float* a_keep = nullptr) {
    GateBwd ga{};
    int rc = swn_make_geom(d, &ga.g);
    if (rc < 0) return rc;
    const SwnGeom& g = ga.g;
    const bool drop = drop_x != nullptr;
    if (!packed || !aux || (!drop && !cond) || (drop && !drop_h) || !fe_work || !audio || (!fwd_work && !hs_bf16) || !grad_out || !work ||
        !gpacked || batch < 1 || batch > 65535 || n_frames < 1) return SWN_E_BADARG;
    if (g.Hp != g.H || g.K > 8 || g.U > 256) return SWN_E_UNSUPPORTED;
    swn_make_layout(&ga.g, &ga.y);
    const SwnLayout& y = ga.y;
    const long T = (long)n_frames * g.U;
    const int Tp = (int)(g.kind == SWN_KIND_SOFTMAX ? T - 1 : T - 2 * g.seg + 1);
    if (Tp < 1) return SWN_E_BADARG;
    // dropout mode, BL6 class, mixed-precision mode: the forward of the same call mode was swn_bl6_drop_forward (same predicate),
    // fwd_work holds its SwnBl6DropLayout and the layers take the fused backward with sample-rate in_x operands
    const bool drop_fused = drop && !hs_opt && !hs_bf16 && mode_bf16() && swn_bl6_drop_supported(g, batch, Tp, n_frames, drop_h);
    SwnBl6DropLayout dlo{};
    if (drop_fused) {
        dlo = swn_bl6_drop_layout(g, batch, Tp);
        hs_bf16 = reinterpret_cast<const unsigned char*>(fwd_work) + dlo.hs16;
    }
    {   // the contraction kernels address one utterance's operands with 32-bit byte offsets
        size_t widest = (size_t)(g.L + 1) * g.H;
        if ((size_t)g.S > widest) widest = g.S;
        if ((size_t)g.O1 > widest) widest = g.O1;
        if ((size_t)g.A0 > widest) widest = g.A0;
        if (widest * (size_t)T * sizeof(float) >= (1ull << 31)) {
            swn_set_error_detail(where, "one utterance's activations exceed 2 GiB: split the chunk");
            return SWN_E_UNSUPPORTED;
        }
    }
    hipStream_t st = (hipStream_t)stream_;
    (void)hipGetLastError();
    const int B = batch, H = g.H, H2 = 2 * g.H, L = g.L, S = g.S, O1 = g.O1, NO = g.NO;
    // forward buffers (layout of swn_forward_work_floats): hs | s1 = relu(skip) | r1 = relu(out_1)
    const size_t hs_floats = r64((size_t)B * (L + 1) * H * Tp);
    const float* hs = hs_opt ? hs_opt : fwd_work;
    const float* s1 = fwd_work + hs_floats;
    const float* r1 = s1 + r64((size_t)B * S * Tp);
    // scratch
    float* do1 = work;
    float* dskip = do1 + r64((size_t)B * O1 * Tp);
    float* dhs = dskip + r64((size_t)B * S * Tp);
    float* a_da = dhs + r64((size_t)B * (L + 1) * H * Tp);
    float* dgx = a_da + r64((size_t)B * H2 * Tp);
    float* dcond = dgx + r64((size_t)B * H2 * Tp);
    float* dfe = dcond + r64((size_t)B * n_frames * g.N);
    // dropout mode: forward work tail = xm | gx (swn_forward_drop); scratch tail = dxm
    const int coff = g.kind == SWN_KIND_SOFTMAX ? 1 : g.seg;
    const int Tx = (int)(T - coff);
    size_t fe_tot = (size_t)g.n_aux; for (int i = 0; i < g.auxl; ++i) fe_tot += g.aux_cout[i];
    const float* xm = r1 + r64((size_t)B * O1 * Tp);
    const float* gx = xm + r64((size_t)B * swn_a0x(&g) * Tx);
    // dropout mode in the mixed-precision mode: swn_forward_drop of the same mode kept every layer's gate pre-activations
    // behind gx (the arithmetic mode must not change between a forward and its backward)
    const float* saved_a = (drop && mode_bf16() && !hs_opt && swn_drop_bf16_forward(&g))
                               ? gx + r64((size_t)B * L * H2 * Tp) : a_keep;   // a_keep: swn_forward_bf16_keep's buffer (same slots)
    float* dxm = dfe + r64(fe_tot * B * n_frames);
    float* hmask = dxm + r64((size_t)B * g.A0 * Tx);               // dropout mode only: masked input of a layer (B, H, Tp)
    float* bl6_scratch = nullptr;
    if (hs_bf16) {                                     // compact layout: none of the fp32 sample-rate scratch exists
        dcond = work;
        dfe = dcond + r64((size_t)B * n_frames * g.N);
        dxm = dfe + r64(fe_tot * B * n_frames);
        bl6_scratch = drop_fused ? dxm + r64((size_t)B * Tx * swn_a0x(&g) / 2) : dxm;      // dropout mode: d xm first, bf16 [B][Tx][A0x]
    }
    if (hipMemsetAsync(gpacked, 0, y.total * sizeof(float), st) != hipSuccess) return SWN_E_LAUNCH;
    // teacher-forced chain without dropout: the partial g w_up copies of cond_bwd_kernel sit where the dropout mode keeps dxm
    float* wup_part = (!hs_bf16 && !drop) ? dxm : nullptr;
    // mixed-precision chain: gate_bwd also leaves da as bf16 rows (pitch = Tp rounded up to 32) for the layer weight gradients
    const long da16_pitch = (Tp + 2 + 31) & ~31L, da16_odd = (long)B * H2 * da16_pitch;
    unsigned short* da16 = nullptr;
    if (!hs_bf16 && mode_bf16())
        da16 = reinterpret_cast<unsigned short*>(drop ? hmask + r64((size_t)B * H * Tp) : dxm + (size_t)SWN_WUP_COPIES * 256);
    ga.da16 = da16; ga.da16_pitch = da16_pitch; ga.da16_odd = da16_odd;
    // ... and the layer's (masked) input, for the weight gradient's Q operand; second copy within 32-bit reach as for da
    const long h16_odd = (long)B * H * da16_pitch;
    unsigned short* h16 = nullptr;
    if (da16 && Tp >= 256 && ((size_t)h16_odd + (size_t)H * da16_pitch) * 2 < (1ull << 31))
        h16 = da16 + 2 * r64((size_t)B * H2 * da16_pitch) + 2 * r64((size_t)L * g.K * H * H2 / 2);
    ga.h16 = h16; ga.h16_odd = h16_odd;
    unsigned short* wdt16 = nullptr;             // [l][tap][i][o2]
    // (the kernel reaches the second da copy through a 32-bit byte offset from an utterance's rows: it must stay below 2 GiB,
    //  else the data gradients keep their fp32 X operand)
    if (da16 && g.Hp == H && H2 % 32 == 0 && ((size_t)da16_odd + (size_t)H2 * da16_pitch) * 2 < (1ull << 31)) {
        wdt16 = da16 + 2 * r64((size_t)B * H2 * da16_pitch);
        hipLaunchKernelGGL(wd_t16_kernel, dim3(1024), dim3(256), 0, st, packed + y.wd, wdt16, L, g.K, H, g.Hp);
        // launch_reduce / launch_time take their bf16-copy kernels under exactly these conditions: nobody reads the fp32 da then
        ga.skip_da32 = Tp >= 256 ? 1 : 0;
    }
    if (wup_part && hipMemsetAsync(wup_part, 0, (size_t)SWN_WUP_COPIES * 256 * sizeof(float), st) != hipSuccess) return SWN_E_LAUNCH;
    // (the fused layer path writes d h_0 whole and keeps the other carries in its own buffers)
    // Only level 0 needs zeros: levels 1..L are written whole by the skip data gradient below (accumulate = 0) before anything
    // is added to them; d h_0 only ever receives additions.  (Zeroing all L + 1 levels was 710 MB = 0.11 ms per step at REF6.)
    // (hipMemset2DAsync's fill kernel took 273 us for these 8 rows of 12.7 MB; a plain grid-stride store kernel is at the HBM rate)
    if (!hs_bf16) hipLaunchKernelGGL(zero_rows_kernel, dim3(512, B), dim3(256), 0, st, dhs, (size_t)(L + 1) * H * Tp, (size_t)H * Tp);
    const long hsb = (long)(L + 1) * H * Tp;

    // ---- head: out_2, out_1, skip
    if (!hs_bf16) {
    {   // do1 = relu'(r1) . W2^T dY ; gW2 += dY r1^T
        TimeGemm t = {packed + y.w2, 1, 0, g.O1p, grad_out, (long)NO * Tp, Tp, 1, do1, (long)O1 * Tp, Tp, r1, (long)O1 * Tp, Tp,
                      O1, 1, NO, Tp, 1, 0, 1, 0};
        launch_time(t, B, st);
        ReduceGemm r = {grad_out, (long)NO * Tp, Tp, 1, r1, (long)O1 * Tp, Tp, 1, gpacked + y.w2, g.O1p, 0, 1, gpacked + y.b2,
                        NO, 1, O1, Tp, 1, 0, 1, 0};
        launch_reduce(r, B, st);
    }
    {   // dskip = relu'(s1) . W1^T do1 ; gW1 += do1 s1^T
        TimeGemm t = {packed + y.w1, 1, 0, g.Sp, do1, (long)O1 * Tp, Tp, 1, dskip, (long)S * Tp, Tp, s1, (long)S * Tp, Tp,
                      S, 1, O1, Tp, 1, 0, 1, 0};
        launch_time(t, B, st);
        ReduceGemm r = {do1, (long)O1 * Tp, Tp, 1, s1, (long)S * Tp, Tp, 1, gpacked + y.w1, g.Sp, 0, 1, gpacked + y.b1,
                        O1, 1, S, Tp, 1, 0, 1, 0};
        launch_reduce(r, B, st);
    }
    {   // dh_l (skip part) = Wsk_l^T dskip for all l at once ; gWsk += dskip hcat^T
        TimeGemm t = {packed + y.wsk, 1, 0, (long)L * g.Hp, dskip, (long)S * Tp, Tp, 1, dhs + (size_t)H * Tp, hsb, Tp, nullptr, 0, 0,
                      L * H, 1, S, Tp, 1, 0, 1, 0};
        launch_time(t, B, st);
        ReduceGemm r = {dskip, (long)S * Tp, Tp, 1, hs + (size_t)H * Tp, hsb, Tp, 1, gpacked + y.wsk, (long)L * g.Hp, 0, 1, gpacked + y.bsk,
                        S, 1, L * H, Tp, 1, 0, 1, 0};
        launch_reduce(r, B, st);
    }
    }
    ga.P = packed; ga.cond = cond; ga.audio = audio; ga.hs = hs; ga.dhs = dhs; ga.a_da = a_da; ga.dgx = dgx; ga.dgx_sb = (long)H2 * Tp;
    // dropout chain, seg == 1: every layer's d gx is kept - (B, L, 2H, Tp) at the end of the work buffer - so that the gradient wrt
    // the masked conditioning is ONE contraction over the L * 2H rows of in_x behind the loop instead of L read-modify-write
    // passes over d xm (257 MB each way at the run.sh geometry)
    float* dgx_all = (drop && !drop_fused && g.seg == 1) ? work + drop_chain_floats(g, B, n_frames) : nullptr;
    // swn_drop_inx16 (the forward of the same mode left xm as bf16 rows): d gx is kept as bf16 rows in that section instead, and the
    // transposed bf16 in_x matrix sits behind it
    const bool inx16 = dgx_all && !hs_opt && mode_bf16() && da16 && swn_drop_inx16(&g, Tp);
    unsigned short* dgx16_all = inx16 ? reinterpret_cast<unsigned short*>(dgx_all) : nullptr;
    unsigned short* wxt16 = inx16 ? reinterpret_cast<unsigned short*>(dgx_all + r64((size_t)B * L * H2 * Tp)) : nullptr;
    ga.B = B; ga.Tf = n_frames; ga.Tp = Tp; ga.coff = coff;
    ga.gx = drop ? gx : nullptr;
    // the GEMM stack keeps its pre-activations - and, when it ran the dropout-mode forward, read its in_x rows - in the G4 layout
    ga.g4 = (a_keep || (drop && !drop_fused && !hs_opt && mode_bf16() && swn_drop_g16(d, B, Tp) && swn_drop_inx16(&g, Tp))) ? 1 : 0;
    ga.gwxa = g.audio_in ? gpacked + y.wxa : nullptr;
    // ---- layers, last to first
    if (hs_bf16) {
        if ((drop && !drop_fused) || !swn_bl6_bwd_supported(g, B, Tp, n_frames)) return SWN_E_UNSUPPORTED;
        const unsigned char* fw = reinterpret_cast<const unsigned char*>(fwd_work);
        const int rcl = swn_bl6_bwd_stack(g, y, packed, cond, reinterpret_cast<const float*>(audio), hs_bf16, grad_out, dcond, gpacked,
                                          bl6_scratch, B, n_frames, Tp, st,
                                          drop_fused ? reinterpret_cast<const unsigned short*>(fw + dlo.gx16) : nullptr,
                                          drop_fused ? reinterpret_cast<const unsigned short*>(fw + dlo.xm16) : nullptr,
                                          drop_fused ? reinterpret_cast<unsigned short*>(dxm) : nullptr);
        if (rcl < 0) return rcl;
    }
    for (int l = hs_bf16 ? -1 : L - 1; l >= 0; --l) {
        ga.l = l;
        if (dgx_all) { dgx = dgx_all + (size_t)l * H2 * Tp; ga.dgx = dgx; ga.dgx_sb = (long)L * H2 * Tp; }
        if (dgx16_all) { ga.dgx16 = dgx16_all + (size_t)l * H2 * da16_pitch; ga.dgx16_sb = (long)L * H2 * da16_pitch; }
        else if (!drop && a_keep && da16 && da16_pitch <= 2L * Tp) {   // GEMM-stack class, no dropout: cond_bwd_kernel reads the bf16 rows
            // (same section, half of it; the small nets keep fp32 here: the rounding moved a 2e-2 gradient check of the 32-channel fixtures)
            ga.dgx16 = reinterpret_cast<unsigned short*>(dgx); ga.dgx16_sb = (long)H2 * da16_pitch;
        }
        // dropout mode: this layer's input is h_{l-1} times the mask drawn for layer l-1's output (cswnv_shift1.py:269-273)
        const float* in_mul = (drop && l > 0) ? drop_h[l - 1] : nullptr;
        ga.in_mul = in_mul;
        const float* Wd = packed + y.wd + (size_t)l * H2 * g.K * g.Hp;                 // [o2][tap][i]
        const float* xin = hs + (size_t)l * H * Tp; long xin_sb = hsb;                 // the layer's input as the GEMMs read it
        if (in_mul) {
            const long npb = (long)H * Tp;
            hipLaunchKernelGGL(mask_mul_kernel, dim3((unsigned)((npb / 4 + 255) / 256 + 1), B), dim3(256), 0, st, xin, hsb, in_mul, hmask, npb);
            xin = hmask; xin_sb = npb;
        }
        ga.a_in = nullptr;
        if (saved_a) ga.a_in = saved_a + (size_t)l * r64((size_t)B * H2 * Tp);   // the forward of the same mode kept them
        else {   // a = Wd (*) h_{l-1}   (bias added in the gate kernel)
            TimeGemm t = {Wd, (long)g.K * g.Hp, g.Hp, 1, xin, xin_sb, Tp, 1, a_da, (long)H2 * Tp, Tp, nullptr, 0, 0,
                          H2, g.K, H, Tp, 1, g.K - 1, g.dil[l], 0};
            launch_time(t, B, st);
        }
        if (ga.g4 && ga.a_in) {
            dim3 grid((Tp + 255) / 256, H / 4, B);
            if (g.kind == SWN_KIND_LAPLACE) hipLaunchKernelGGL(gate_bwd_g4_kernel<SWN_KIND_LAPLACE>, grid, dim3(256), 0, st, ga);
            else hipLaunchKernelGGL(gate_bwd_g4_kernel<SWN_KIND_SOFTMAX>, grid, dim3(256), 0, st, ga);
        } else {
            dim3 grid((Tp + 255) / 256, H, B);
            if (g.kind == SWN_KIND_LAPLACE) hipLaunchKernelGGL(gate_bwd_kernel<SWN_KIND_LAPLACE>, grid, dim3(256), 0, st, ga);
            else hipLaunchKernelGGL(gate_bwd_kernel<SWN_KIND_SOFTMAX>, grid, dim3(256), 0, st, ga);
        }
        {   // gWd += da X^T (taps shifted back), gbd += rowsum(da)
            ReduceGemm r = {a_da, (long)H2 * Tp, Tp, 1, xin, xin_sb, Tp, 1,
                            gpacked + y.wd + (size_t)l * H2 * g.K * g.Hp, (long)g.K * g.Hp, g.Hp, 1, gpacked + y.bd + (size_t)l * H2,
                            H2, g.K, H, Tp, 1, g.K - 1, g.dil[l], 0};
            r.P16 = da16; r.p16_sb = (long)H2 * da16_pitch; r.p16_sm = da16_pitch;
            r.Q16 = h16; r.q16_sb = (long)H * da16_pitch; r.q16_sc = da16_pitch; r.q16_odd = h16_odd;
            launch_reduce(r, B, st);
        }
        {   // dh_{l-1} += Wd^T (*) da  (taps shifted forward): A(m=i, tap, c=o2) = Wd[o2][tap][i]
            TimeGemm t = {Wd, 1, g.Hp, (long)g.K * g.Hp, a_da, (long)H2 * Tp, Tp, 1, dhs + (size_t)l * H * Tp, hsb, Tp, nullptr, 0, 0,
                          H, g.K, H2, Tp, -1, g.K - 1, g.dil[l], 1};
            t.ymul = in_mul; t.ym_sb = (long)H * Tp; t.ym_sm = Tp;
            if (wdt16) {
                t.A16 = wdt16 + (size_t)l * g.K * H * H2; t.a16_sm = H2; t.a16_stap = (long)H * H2;
                t.X16 = da16; t.x16_sb = (long)H2 * da16_pitch; t.x16_sc = da16_pitch; t.x16_odd = da16_odd;
            }
            launch_time(t, B, st);
        }
        if (!drop) {
            // frames per workgroup: as many as still leave ~768 workgroups
            const int rb = (H2 + 63) / 64;
            int FR = (int)(((long)rb * n_frames * B) / 768); FR = FR < 1 ? 1 : (FR > 16 ? 16 : FR);
            const dim3 cgrid(rb, (n_frames + FR - 1) / FR, B);
            const size_t clds = (size_t)64 * ((g.U + g.seg - 1) | 1) * sizeof(float);
            const int cch = (g.U + g.seg - 1 + 63) / 64;                              // 64-column chunks of a frame's tile (U <= 256: at most 5)
            if (cch <= 2) hipLaunchKernelGGL(cond_bwd_kernel<2>, cgrid, dim3(256), clds, st, ga, dcond, gpacked + y.bx, wup_part, FR);
            else hipLaunchKernelGGL(cond_bwd_kernel<5>, cgrid, dim3(256), clds, st, ga, dcond, gpacked + y.bx, wup_part, FR);
        } else {
            const float* Wx = packed + y.wx + (size_t)l * g.seg * H2 * g.A0p;          // [s][o][c]
            if (!dgx_all) {   // g in_x.W[l][o][c*seg+s] += sum dgx[o][t] xm[c][t+s] ; g b_inx += rowsum(dgx)
                ReduceGemm r = {dgx, ga.dgx_sb, Tp, 1, xm, (long)swn_a0x(&g) * Tx, Tx, 1,
                                gpacked + y.wx + (size_t)l * g.seg * H2 * g.A0p, g.A0p, (long)H2 * g.A0p, 1,
                                gpacked + y.bxr + (size_t)l * H2, H2, g.seg, g.A0, Tp, 1, 0, 1, 0};
                r.QT = Tx;
                launch_reduce(r, B, st);
            }
            if (!dgx_all) {   // dxm[c][u] (+)= sum_{s,o} W[l][o][c*seg+s] dgx[o][u-s]
                TimeGemm t = {Wx, 1, (long)H2 * g.A0p, g.A0p, dgx, (long)H2 * Tp, Tp, 1, dxm, (long)g.A0 * Tx, Tx, nullptr, 0, 0,
                              g.A0, g.seg, H2, Tx, -1, 0, 1, l == L - 1 ? 0 : 1};
                t.XT = Tp;
                launch_time(t, B, st);
            }
        }
    }
    if (dgx_all) {   // the two in_x contractions over all layers at once (seg == 1: row n = l*2H + o of the [N][A0p] matrix)
        // g in_x.W[n][c] += sum dgx[n][t] xm[c][t] ; g b_inx[n] += rowsum(dgx[n])
        ReduceGemm r = {dgx_all, (long)L * H2 * Tp, Tp, 1, xm, (long)swn_a0x(&g) * Tx, Tx, 1, gpacked + y.wx, g.A0p, 0, 1,
                        gpacked + y.bxr, L * H2, 1, g.A0, Tp, 1, 0, 1, 0};
        r.QT = Tx;
        if (inx16) {
            const long px = swn_pitch16(Tx);
            r.P16 = dgx16_all; r.p16_sb = (long)L * H2 * da16_pitch; r.p16_sm = da16_pitch;
            r.Q16 = reinterpret_cast<const unsigned short*>(xm); r.q16_sb = (long)swn_a0x(&g) * px; r.q16_sc = px; r.q16_odd = 0;
        }
        launch_reduce(r, B, st);
        // dxm[c][u] = sum_n in_x.W[n][c] dgx[n][u]
        TimeGemm t = {packed + y.wx, 1, 0, g.A0p, dgx_all, (long)L * H2 * Tp, Tp, 1, dxm, (long)g.A0 * Tx, Tx, nullptr, 0, 0,
                      g.A0, 1, L * H2, Tx, -1, 0, 1, 0};
        t.XT = Tp;
        if (inx16) {
            hipLaunchKernelGGL(wx16_kernel, dim3(512), dim3(256), 0, st, packed + y.wx, L * H2, g.A0, g.A0p, swn_a0x(&g), nullptr, wxt16);
            t.A16 = wxt16; t.a16_sm = L * H2; t.a16_stap = 0;
            t.X16 = dgx16_all; t.x16_sb = (long)L * H2 * da16_pitch; t.x16_sc = da16_pitch; t.x16_odd = 0;
        }
        launch_time(t, B, st);
    }
    if (wup_part) hipLaunchKernelGGL(wup_fold_kernel, dim3(1), dim3(256), 0, st, wup_part, gpacked + y.wup, g.U);
    // ---- input layer (the fused layer path has done it from its accumulators)
    if (hs_bf16) {}
    else if (g.kind == SWN_KIND_LAPLACE) hipLaunchKernelGGL(input_bwd_kernel<SWN_KIND_LAPLACE>, dim3(H, B), dim3(256), 0, st, ga, gpacked);
    else hipLaunchKernelGGL(input_bwd_kernel<SWN_KIND_SOFTMAX>, dim3(H, B), dim3(256), 0, st, ga, gpacked);
    // ---- frame-rate front end: cond = Wx . C ; C = conv_aux(scale_in(aux))
    {
        const size_t bt = (size_t)B * n_frames;
        // forward activations inside fe_work: scaled | aux conv outputs...
        const float* act[SWN_MAXAUX + 1]; float* dact[SWN_MAXAUX + 1];
        int chn[SWN_MAXAUX + 1];
        const float* p = fe_work; float* q = dfe;
        chn[0] = g.n_aux; act[0] = p; dact[0] = q; p += bt * g.n_aux; q += bt * g.n_aux;
        for (int i = 0; i < g.auxl; ++i) { chn[i + 1] = g.aux_cout[i]; act[i + 1] = p; dact[i + 1] = q; p += bt * g.aux_cout[i]; q += bt * g.aux_cout[i]; }
        const float* C = act[g.auxl];
        if (drop_fused) {
            const int cb = (g.A0 + XM16_CC - 1) / XM16_CC;
            int FRx = (int)(((long)cb * n_frames * B) / 1024); FRx = FRx < 1 ? 1 : (FRx > 16 ? 16 : FRx);
            hipLaunchKernelGGL(xm_bwd16_kernel, dim3((n_frames + FRx - 1) / FRx, cb, B), dim3(256), 0, st,
                               reinterpret_cast<const unsigned short*>(dxm), drop_x, C, packed, y.wup, dact[g.auxl], gpacked + y.wup,
                               gpacked + y.bup, g.A0, swn_a0x(&g), n_frames, g.U, coff, Tx, FRx);
        } else if (drop) {
            const int cb = (g.A0 + XM_CC - 1) / XM_CC;
            int FRx = (int)(((long)cb * n_frames * B) / 1024); FRx = FRx < 1 ? 1 : (FRx > 16 ? 16 : FRx);
            hipLaunchKernelGGL(xm_bwd_kernel, dim3((n_frames + FRx - 1) / FRx, cb, B), dim3(256), 0, st, dxm, drop_x, C, packed, y.wup,
                               dact[g.auxl], gpacked + y.wup, gpacked + y.bup, g.A0, n_frames, g.U, coff, Tx, FRx);
        } else
        {   // dC[b][c][f] = sum_n Wx[n][c] dcond[b][f][n] ; gWx[n][c] += sum_{b,f} dcond[b][f][n] C[b][c][f]
            TimeGemm t = {packed + y.wx, 1, 0, g.A0p, dcond, (long)n_frames * g.N, 1, g.N, dact[g.auxl], (long)g.A0 * n_frames, n_frames,
                          nullptr, 0, 0, g.A0, 1, g.N, n_frames, 1, 0, 1, 0};
            launch_time(t, B, st);
            ReduceGemm r = {dcond, (long)n_frames * g.N, 1, g.N, C, (long)g.A0 * n_frames, n_frames, 1, gpacked + y.wx, g.A0p, 0, 1, nullptr,
                            g.N, 1, g.A0, n_frames, 1, 0, 1, 0};
            launch_reduce(r, B, st);
        }
        for (int i = g.auxl - 1; i >= 0; --i) {
            const int ci = g.aux_cin[i], co = g.aux_cout[i], ks = g.auxk, half = (g.auxk - 1) / 2, dl = g.aux_dil[i];
            {   // gW[co][ci][k] += sum dY[co][f] X[ci][f + (k-half)*dil] ; gb
                ReduceGemm r = {dact[i + 1], (long)co * n_frames, n_frames, 1, act[i], (long)ci * n_frames, n_frames, 1,
                                gpacked + y.aux_w[i], (long)ci * ks, 1, ks, gpacked + y.aux_b[i], co, ks, ci, n_frames, 1, half, dl, 0};
                launch_reduce(r, B, st);
            }
            {   // dX[ci][f] = sum_{k,co} W[co][ci][k] dY[co][f - (k-half)*dil]
                TimeGemm t = {packed + y.aux_w[i], ks, 1, (long)ci * ks, dact[i + 1], (long)co * n_frames, n_frames, 1, dact[i],
                              (long)ci * n_frames, n_frames, nullptr, 0, 0, ci, ks, co, n_frames, -1, half, dl, 0};
                launch_time(t, B, st);
            }
        }
        {   // scale_in: 1x1 on the raw features
            ReduceGemm r = {dact[0], (long)g.n_aux * n_frames, n_frames, 1, aux, (long)g.n_aux * n_frames, n_frames, 1,
                            gpacked + y.scale_w, g.n_aux, 0, 1, gpacked + y.scale_b, g.n_aux, 1, g.n_aux, n_frames, 1, 0, 1, 0};
            launch_reduce(r, B, st);
        }
    }
    return swn_launch_status(where);
}

}  // namespace

extern "C" int swn_backward(const swn_net_desc* d, const float* packed, const float* aux, const float* cond,
                            const float* fe_work, const void* audio, const float* fwd_work, const float* hs_opt,
                            const float* grad_out, int batch, int n_frames, float* work, float* gpacked, int precision,
                            void* stream_) {
    if (!swn_precision_ok(precision)) return SWN_E_BADARG;
    SwnModeScope mode(precision);
    return backward_impl(d, packed, aux, cond, fe_work, audio, fwd_work, hs_opt, nullptr, nullptr, grad_out, batch, n_frames,
                         work, gpacked, stream_, "swn_backward");
}

// mixed-precision backward of the BL6 class with the gated layers fused per layer: work_bf16_dev is the work buffer
// swn_forward_bf16 filled (bf16 time-major hidden states), fwd_work_dev its fp32 expansion (swn_bf16_work_to_f32).
extern "C" size_t swn_backward_bf16_work_floats(const swn_net_desc* d, int batch, int n_frames) {
    SwnGeom g; if (swn_make_geom(d, &g) < 0 || batch < 1 || n_frames < 1) return 0;
    const long Tp = (long)n_frames * g.U - 2 * g.seg + 1;
    if (g.kind != SWN_KIND_LAPLACE || Tp < 1 || !swn_bl6_bwd_supported(g, batch, Tp, n_frames)) return 0;
    size_t fw = (size_t)g.n_aux; for (int i = 0; i < g.auxl; ++i) fw += g.aux_cout[i];
    return r64((size_t)batch * n_frames * g.N) + r64(fw * batch * n_frames) + r64((swn_bl6_bwd_scratch_bytes(g, batch, Tp) + 3) / 4);
}

extern "C" int swn_backward_bf16(const swn_net_desc* d, const float* packed, const float* aux, const float* cond,
                                 const float* fe_work, const void* audio, const float* fwd_work, const void* work_bf16,
                                 const float* grad_out, int batch, int n_frames, float* work, float* gpacked, void* stream_) {
    if (!work_bf16) return SWN_E_BADARG;
    if (swn_backward_bf16_work_floats(d, batch, n_frames) == 0) return SWN_E_UNSUPPORTED;
    SwnModeScope mode(SWN_PRECISION_BF16);                 // this path exists in the mixed-precision arithmetic only
    return backward_impl(d, packed, aux, cond, fe_work, audio, fwd_work, nullptr, nullptr, nullptr, grad_out, batch, n_frames,
                         work, gpacked, stream_, "swn_backward_bf16", work_bf16);
}

// swn_backward after swn_forward_bf16_keep (GEMM-stack geometries, mixed-precision mode): the gate pre-activations come from
// a_keep_dev, the recompute GEMM of every layer is skipped.  Everything else as swn_backward (same work size).
extern "C" int swn_backward_keep(const swn_net_desc* d, const float* packed, const float* aux, const float* cond,
                                 const float* fe_work, const void* audio, const float* fwd_work, const float* a_keep,
                                 const float* grad_out, int batch, int n_frames, float* work, float* gpacked, void* stream_) {
    if (!a_keep) return SWN_E_BADARG;
    SwnModeScope mode(SWN_PRECISION_BF16);                 // follows swn_forward_bf16_keep: mixed precision by construction
    return backward_impl(d, packed, aux, cond, fe_work, audio, fwd_work, nullptr, nullptr, nullptr, grad_out, batch, n_frames,
                         work, gpacked, stream_, "swn_backward_keep", nullptr, a_keep);
}

extern "C" size_t swn_backward_drop_work_floats(const swn_net_desc* d, int batch, int n_frames) {
    SwnGeom g; if (swn_make_geom(d, &g) < 0) return 0;
    if (!swn_backward_work_floats(d, batch, n_frames)) return 0;
    const long T = (long)n_frames * g.U;
    const long Tp = g.kind == SWN_KIND_SOFTMAX ? T - 1 : T - 2 * g.seg + 1;
    // the generic chain's layout, + (seg == 1) every layer's d gx behind it
    const size_t chain = drop_chain_floats(g, batch, n_frames) + (g.seg == 1 ? r64((size_t)batch * g.L * 2 * g.H * Tp) +
                                                                   r64((size_t)swn_a0x(&g) * g.L * 2 * g.H / 2 + 1) : 0);   // + transposed bf16 in_x matrix
    size_t fused = 0;                                  // the fused BL6 path: d cond (unused) | front-end gradients | d xm | its scratch
    if (g.kind == SWN_KIND_LAPLACE && swn_bl6_bwd_supported(g, batch, Tp, n_frames)) {
        size_t fw = (size_t)g.n_aux; for (int i = 0; i < g.auxl; ++i) fw += g.aux_cout[i];
        fused = r64((size_t)batch * n_frames * g.N) + r64(fw * batch * n_frames) + r64((size_t)batch * swn_a0x(&g) * (T - g.seg) / 2) +
                r64((swn_bl6_bwd_drop_scratch_bytes(g, batch, Tp) + 3) / 4);
    }
    return chain > fused ? chain : fused;
}

extern "C" int swn_backward_drop(const swn_net_desc* d, const float* packed, const float* aux, const float* fe_work,
                                 const void* audio, const float* fwd_work, const float* hs_opt, const float* drop_x,
                                 const float* const* drop_h, const float* grad_out, int batch, int n_frames, float* work,
                                 float* gpacked, int precision, void* stream_) {
    if (!drop_x || !swn_precision_ok(precision)) return SWN_E_BADARG;
    SwnModeScope mode(precision);
    return backward_impl(d, packed, aux, nullptr, fe_work, audio, fwd_work, hs_opt, drop_x, drop_h, grad_out, batch, n_frames,
                         work, gpacked, stream_, "swn_backward_drop");
}

extern "C" int swn_laplace_head_backward(const swn_net_desc* d, const float* raw, int batch, int tp, const float* gmu,
                                         const float* gb, const float* glogb, const float* ga, const float* gbc,
                                         const float* glc, float* graw, void* stream_) {
    SwnGeom g; int rc = swn_make_geom(d, &g);
    if (rc < 0) return rc;
    if (g.kind != SWN_KIND_LAPLACE) return SWN_E_BADDESC;
    if (!raw || !graw || batch < 1 || batch > 65535 || tp < 1) return SWN_E_BADARG;
    (void)hipGetLastError();
    hipLaunchKernelGGL(laplace_head_bwd_kernel, dim3((tp + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream_,
                       raw, tp, g.seg, g.lpc, gmu, gb, glogb, ga, gbc, glc, graw);
    return swn_launch_status("swn_laplace_head_backward");
}
