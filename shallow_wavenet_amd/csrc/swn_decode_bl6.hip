// BL6-class autoregressive decode on gfx950 (BASELINE-literal shape: 1x6 dilated stack, H=64,
// K=2, dilations 1..32, rf=64; Laplace S=128 with seg/lpc variants, softmax S=Q=256).
//
// One persistent 512-thread workgroup (8 waves, 2 per SIMD) per utterance.  What makes it
// different from the generic kernel (swn_decode.hip):
//   * the six dilated-conv weight matrices (6 x 128 x 128 fp32 = 384 KB) live in VGPRs for the
//     whole utterance: thread (o,p) owns gate row o and candidate row o+64 over 16 of the
//     128 inputs, 192 registers; rows are reduced over 8 adjacent lanes with DPP adds;
//   * the per-dilation history rings (cswnv_shift1.py:324-334 output_buffer) are LDS rings
//     with power-of-two lengths: 32-36 KB; a tap is one broadcast ds_read_b128 per 16 bytes;
//   * the gate  z=sigmoid, tanh, highway  epilogue is fused behind the reduction (no second
//     pass, one s_barrier per layer); with seg>1 lane p finishes position p;
//   * out_skip is accumulated layer by layer in the shadow of the next layer's phase, so only
//     one 128x64 slice sits on the critical path; out_1/out_skip stream from L2 in a
//     lane-tiled layout (1 KiB contiguous per wave instruction);
//   * conditioning: the current and next frame of the hoisted in_x product sit in LDS, the
//     rank-1 upsampler is one fma per gate input (cswnv_shift1.py:37-65,276).
// 9 s_barriers per generated step (10 for softmax).
#include <hip/hip_runtime.h>
#include "swn_geom.hpp"
#include "swn_noise.hpp"

namespace {

constexpr int NT = 512;
constexpr int H = 64;
constexpr int L = 6;
constexpr int RF = 64;
#ifdef SWN_STAMP
constexpr bool HEADS_ON = false;
#else
constexpr bool HEADS_ON = true;
#endif

struct B6Args {
    const float* P;
    SwnLayout y;
    const float* cond;
    const float* noise;            // classic mode: the host-drawn stream
    const void* forced;
    void* out;
    float* heads;
    int B, Tf, n_steps, U, N;
    // extended mode only (in-kernel generator, noise dump, caller's seed waveform)
    SwnNoise nz;
    const void* seed;
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int pow2ceil(int x) { int r = 1; while (r < x) r <<= 1; return r; }
constexpr int r4(int x) { return (x + 3) & ~3; }

template <int S_, int SEG_, int LPC_, int KIND_, int Q_, bool EXT_ = false>
struct Tr {
    static constexpr int S = S_, SEG = SEG_, LPC = LPC_, KIND = KIND_, Q = Q_;
    static constexpr bool EXT = EXT_;     // extended mode: in-kernel noise generator / noise dump / seed waveform
    using Ext = Tr<S_, SEG_, LPC_, KIND_, Q_, true>;
    static constexpr int O1 = KIND_ == SWN_KIND_SOFTMAX ? Q_ : S_;
    static constexpr int NO = KIND_ == SWN_KIND_SOFTMAX ? Q_ : 2 * SEG_ + LPC_;
    static constexpr int WN = cmax(1, LPC_) + SEG_;
    static constexpr int ring_len(int l) { return pow2ceil((1 << l) + SEG_); }
    static constexpr int ring_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += ring_len(i) * H; return o; }
    static constexpr int PF = L * SEG_ * 2 * H;          // floats of one conditioning frame
    static constexpr int NREG = 5;                        // layers whose weights stay in VGPRs; the rest sit in LDS
    // LDS carve (float offsets)
    static constexpr int o_ring = 0;
    static constexpr int o_hcat = o_ring + ring_off(L);
    static constexpr int o_gp = o_hcat + L * H;
    static constexpr int o_bx = o_gp + 2 * PF;
    static constexpr int o_bd = o_bx + L * 2 * H;
    static constexpr int o_wup = o_bd + L * 2 * H;
    static constexpr int o_skip = o_wup + 256;
    static constexpr int o_o1 = o_skip + S_;
    static constexpr int o_o2 = o_o1 + O1;
    static constexpr int o_hist = o_o2 + r4(NO);
    // sampling noise staged off the critical path: Laplace [4 chunks][64 steps][SEG] transformed deviates (wave 1 fills
    // a chunk of 64 future steps at once, lane = step); softmax [2 steps][Q] Exp(1) draws (wave 2, one step ahead)
    static constexpr int NZC = 64, NZB = 4;
    static constexpr int o_tnz = o_hist + r4(WN);
    // (the classic Laplace instantiation keeps the round-1 carve: [2][8] deviates of the next step - LDS offsets past
    //  64 KB cost an address add each, so the layout is part of the measured kernel)
    static constexpr int o_cz = o_tnz + (KIND_ == SWN_KIND_LAPLACE ? (EXT_ ? r4(NZB * NZC * SEG_) : 16) : 2 * Q_);   // cb[64], cv[2][64], cc[2][64]
    static constexpr int o_w2 = o_cz + 5 * H;             // laplace: out_2 rows [NO][S] (+b2)
    static constexpr int o_bias = o_w2 + (KIND_ == SWN_KIND_LAPLACE ? NO * S_ + r4(NO) : 0);   // bsk[S], b1[O1], (softmax) b2[NO]
    static constexpr int o_wl = o_bias + S_ + O1 + (KIND_ == SWN_KIND_SOFTMAX ? NO : 0);          // [L-NREG][8][512][4]
    // most of the out_1 matrix (input slices 0..W1L-1 of 8, lane-tiled like the global copy; as many as the 160 KB
    // of LDS allow) stays in LDS for the single-sample Laplace net: the streamed rest's L2 latency then hides
    // under the resident slices' FMAs
    static constexpr int W1L = (KIND_ == SWN_KIND_LAPLACE && S_ == 128 && SEG_ <= 2) ? 4 : 0;
    static constexpr int o_w1l = o_wl + (L - NREG) * 8 * NT * 4;
    static constexpr int o_end = o_w1l + W1L * O1 * 16;
    static constexpr size_t lds_bytes = (size_t)o_end * sizeof(float);
};

// Streamed weights go through a buffer resource: one 32-bit per-thread offset plus scalar /
// immediate offsets per load, instead of a 64-bit VGPR address per load (which spilled).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, soff_bytes, 0));
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum4(float v) {      // all 4 lanes of a quad get the quad sum
    v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ float sum8(float v) {      // all 8 lanes of an aligned octet get the sum
    v = sum4(v);
    v += dpp_f<0x141>(v);     // row_half_mirror
    return v;
}
// exp(x) = 2^(x*log2e) on the transcendental unit; the product's rounding error is fed back as a
// first-order correction, so the result stays within ~1.5 ulp (libm-grade) at 6 instructions.
__device__ __forceinline__ float exp_c(float x) {
    const float t = x * 1.44269504f;
    const float lo = fmaf(x, 1.44269504f, -t) + x * 1.92596299e-8f;
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, lo * 0.693147181f, e);
}
__device__ __forceinline__ float rcp_c(float x) {        // v_rcp_f32 + one Newton step (<= 1 ulp)
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(r, fmaf(-x, r, 1.f), r);
}
__device__ __forceinline__ float sigm(float x) { return rcp_c(1.f + exp_c(-x)); }
__device__ __forceinline__ float tanh_c(float x) {        // (1 - e^-2|x|) / (1 + e^-2|x|), abs error ~1e-7
    if (fabsf(x) > 9.02f) return copysignf(1.f, x);    // saturated in fp32 (also keeps this a branchy block)
    const float t = exp_c(-2.f * fabsf(x));
    return copysignf((1.f - t) * rcp_c(1.f + t), x);
}
__device__ __forceinline__ float ssign(float x) { return x * rcp_c(1.f + fabsf(x)); }

// Workgroup barrier that orders LDS traffic only: the cross-wave hand-offs of this kernel all go
// through LDS; a plain __syncthreads() also drains (vmcnt(0)) global loads that may stay in flight.
__device__ __forceinline__ void lds_barrier() {
#ifdef SWN_SYNC
    __syncthreads(); return;
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int NP>
__device__ __forceinline__ float pick(const float (&a)[NP], int p) {
    float v = a[0];
#pragma unroll
    for (int j = 1; j < NP; ++j) v = (p == j) ? a[j] : v;
    return v;
}

// One DCRNN layer for NP consecutive positions q0..q0+NP-1 (cswnv_shift1.py:281-285).
template <class T, int LAYER, int NP>
__device__ __forceinline__ void layer_phase(float* lds, const float (&w)[2][16], int q0,
                                            const float (&wj)[T::SEG], const int (&pb)[T::SEG]) {
    constexpr int dil = 1 << LAYER;
    constexpr int R = T::ring_len(LAYER);
    const int tid = threadIdx.x, o = tid >> 3, p = tid & 7;
    const float* ring = lds + T::o_ring + T::ring_off(LAYER);
    float az[NP], ac[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) { az[j] = 0.f; ac[j] = 0.f; }
    // operands of the gate epilogue are fetched up front so their LDS latency hides under the FMAs
    // (lanes 0/1 of an octet: gate / candidate row; other lanes read valid but unused words)
    [[maybe_unused]] float e_bd = 0.f, e_gx = 0.f, e_hp = 0.f;
    if constexpr (NP == 1) {
        const int pr = p & 1;
        e_bd = lds[T::o_bd + LAYER * 2 * H + pr * H + o];
        e_gx = lds[T::o_bx + LAYER * 2 * H + pr * H + o];
#pragma unroll
        for (int s = 0; s < T::SEG; ++s)
            e_gx = fmaf(wj[s], lds[T::o_gp + pb[s] + (LAYER * T::SEG + s) * 2 * H + pr * H + o], e_gx);
        e_hp = ring[(q0 & (R - 1)) * H + o];
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        // inputs 32m+4p..+3 of [tap0 = position q-dil | tap1 = position q]
        float4 w0, w1;
        if (LAYER < T::NREG) {
            w0 = make_float4(w[0][4 * m], w[0][4 * m + 1], w[0][4 * m + 2], w[0][4 * m + 3]);
            w1 = make_float4(w[1][4 * m], w[1][4 * m + 1], w[1][4 * m + 2], w[1][4 * m + 3]);
        } else {
            constexpr int wl = LAYER < T::NREG ? 0 : LAYER - T::NREG;
            w0 = *reinterpret_cast<const float4*>(lds + T::o_wl + ((wl * 8 + m) * NT + tid) * 4);
            w1 = *reinterpret_cast<const float4*>(lds + T::o_wl + ((wl * 8 + 4 + m) * NT + tid) * 4);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int pos = q0 + j - (m < 2 ? dil : 0);
            const float4 x = *reinterpret_cast<const float4*>(ring + (pos & (R - 1)) * H + 32 * (m & 1) + 4 * p);
            az[j] = fmaf(w0.x, x.x, az[j]); ac[j] = fmaf(w1.x, x.x, ac[j]);
            az[j] = fmaf(w0.y, x.y, az[j]); ac[j] = fmaf(w1.y, x.y, ac[j]);
            az[j] = fmaf(w0.z, x.z, az[j]); ac[j] = fmaf(w1.z, x.z, ac[j]);
            az[j] = fmaf(w0.w, x.w, az[j]); ac[j] = fmaf(w1.w, x.w, ac[j]);
        }
        // keep the scheduler from hoisting every tap read of the phase at once (80 VGPRs at seg=5)
        if (NP > 2) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) { az[j] = sum8(az[j]); ac[j] = sum8(ac[j]); }
    __builtin_amdgcn_sched_barrier(0);              // keep the gate epilogue behind the reduction
#ifndef SWN_NO_SPLIT
    if constexpr (NP == 1) {
        // one position: lane 0 evaluates the gate (sigmoid), lane 1 the candidate (tanh), concurrently
        if (p < 2) {
            const int q = q0;
            const float sa = (p == 0 ? az[0] : ac[0]) + e_bd;
            const float v = e_gx * sa;
            float res;
            if (fabsf(v) > 9.02f && p == 1) {
                res = copysignf(1.f, v);                                       // tanh saturated in fp32
            } else {
                const float e = exp_c(p == 0 ? -v : -2.f * fabsf(v));
                const float r = rcp_c(1.f + e);
                res = p == 0 ? r : copysignf((1.f - e) * r, v);                // z | tanh
            }
            const float c = dpp_f<0xF5>(res);                                  // quad_perm [1,1,3,3]: lane 0 <- lane 1
            if (p == 0) {
                const float hn = (1.f - res) * c + res * e_hp;
                if (LAYER + 1 < L) {
                    constexpr int R2 = T::ring_len(LAYER + 1 < L ? LAYER + 1 : LAYER);
                    lds[T::o_ring + T::ring_off(LAYER + 1 < L ? LAYER + 1 : LAYER) + (q & (R2 - 1)) * H + o] = hn;
                }
                lds[T::o_hcat + LAYER * H + o] = hn;
            }
        }
    } else
#endif
    if (p < NP) {                                   // lane p finishes position q0+p
        const int q = q0 + p;
        const float sz = pick<NP>(az, p) + lds[T::o_bd + LAYER * 2 * H + o];
        const float sc = pick<NP>(ac, p) + lds[T::o_bd + LAYER * 2 * H + H + o];
        float gz = lds[T::o_bx + LAYER * 2 * H + o];
        float gc = lds[T::o_bx + LAYER * 2 * H + H + o];
#pragma unroll
        for (int s = 0; s < T::SEG; ++s) {
            const float* pr = lds + T::o_gp + pb[s] + (LAYER * T::SEG + s) * 2 * H;
            gz = fmaf(wj[s], pr[o], gz);
            gc = fmaf(wj[s], pr[H + o], gc);
        }
        const float z = sigm(gz * sz);
        const float c = tanh_c(gc * sc);
        const float hp = ring[(q & (R - 1)) * H + o];
        const float hn = (1.f - z) * c + z * hp;
        if (LAYER + 1 < L) {
            constexpr int R2 = T::ring_len(LAYER + 1 < L ? LAYER + 1 : LAYER);
            lds[T::o_ring + T::ring_off(LAYER + 1 < L ? LAYER + 1 : LAYER) + (q & (R2 - 1)) * H + o] = hn;
        }
        if (p == NP - 1) lds[T::o_hcat + LAYER * H + o] = hn;
    }
    __builtin_amdgcn_sched_barrier(0);
}

// out_skip slice of one layer, 4 lanes per row, weights streamed from the lane-tiled copy.
template <class T, int LAYER>
__device__ __forceinline__ void skip_slice(const float* lds, __amdgpu_buffer_rsrc_t wsk2,
                                           float (&sacc)[T::S / 128]) {
    constexpr int layer = LAYER;
    const int hr = threadIdx.x >> 2, hp = threadIdx.x & 3;
#pragma unroll
    for (int ps = 0; ps < T::S / 128; ++ps) {
        const int row = hr + 128 * ps;
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            const float4 w = buf_ld4(wsk2, (unsigned)(row * 4 + hp) * 16u, (unsigned)((layer * 4 + mm) * T::S) * 64u);
            const float4 x = *reinterpret_cast<const float4*>(lds + T::o_hcat + layer * H + 16 * mm + 4 * hp);
            sacc[ps] = fmaf(w.x, x.x, sacc[ps]); sacc[ps] = fmaf(w.y, x.y, sacc[ps]);
            sacc[ps] = fmaf(w.z, x.z, sacc[ps]); sacc[ps] = fmaf(w.w, x.w, sacc[ps]);
        }
    }
}

// Software-pipelined form: the weights of slice LAYER are issued into registers one phase ahead and
// stay in flight across the LDS-only barrier.
template <class T, int LAYER>
__device__ __forceinline__ void skip_issue(__amdgpu_buffer_rsrc_t wsk2, float4 (&wsl)[4 * (T::S / 128)]) {
    const int hr = threadIdx.x >> 2, hp = threadIdx.x & 3;
#pragma unroll
    for (int ps = 0; ps < T::S / 128; ++ps)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm)
            wsl[ps * 4 + mm] = buf_ld4(wsk2, (unsigned)((hr + 128 * ps) * 4 + hp) * 16u,
                                       (unsigned)((LAYER * 4 + mm) * T::S) * 64u);
}
template <class T, int LAYER>
__device__ __forceinline__ void skip_consume(const float* lds, const float4 (&wsl)[4 * (T::S / 128)],
                                             float (&sacc)[T::S / 128]) {
    const int hp = threadIdx.x & 3;
#pragma unroll
    for (int ps = 0; ps < T::S / 128; ++ps)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm) {
            const float4 w = wsl[ps * 4 + mm];
            const float4 x = *reinterpret_cast<const float4*>(lds + T::o_hcat + LAYER * H + 16 * mm + 4 * hp);
            sacc[ps] = fmaf(w.x, x.x, sacc[ps]); sacc[ps] = fmaf(w.y, x.y, sacc[ps]);
            sacc[ps] = fmaf(w.z, x.z, sacc[ps]); sacc[ps] = fmaf(w.w, x.w, sacc[ps]);
        }
}

// rows x NI mat-vec, 4 lanes per row, lane-tiled weights [NI/16][rows][4][4] streamed from L2.
template <int ROWS, int NI>
__device__ __forceinline__ void tiled_matvec(__amdgpu_buffer_rsrc_t wt, const float* bias,
                                             const float* x, float* y, bool relu) {
    const int hr = threadIdx.x >> 2, hp = threadIdx.x & 3;
#pragma unroll
    for (int ps = 0; ps < ROWS / 128; ++ps) {
        const int row = hr + 128 * ps;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int mm = 0; mm < NI / 16; mm += 2) {
            const float4 w0 = buf_ld4(wt, (unsigned)(row * 4 + hp) * 16u, (unsigned)(mm * ROWS) * 64u);
            const float4 w1 = buf_ld4(wt, (unsigned)(row * 4 + hp) * 16u, (unsigned)((mm + 1) * ROWS) * 64u);
            const float4 x0 = *reinterpret_cast<const float4*>(x + 16 * mm + 4 * hp);
            const float4 x1 = *reinterpret_cast<const float4*>(x + 16 * (mm + 1) + 4 * hp);
            a0 = fmaf(w0.x, x0.x, a0); a0 = fmaf(w0.y, x0.y, a0); a0 = fmaf(w0.z, x0.z, a0); a0 = fmaf(w0.w, x0.w, a0);
            a1 = fmaf(w1.x, x1.x, a1); a1 = fmaf(w1.y, x1.y, a1); a1 = fmaf(w1.z, x1.z, a1); a1 = fmaf(w1.w, x1.w, a1);
            if ((mm & 7) == 6 && mm + 2 < NI / 16) __builtin_amdgcn_sched_barrier(0);   // <= 8 loads in flight
        }
        float v = sum4(a0 + a1);
        if (hp == 0) {
            v += bias[row];
            y[row] = relu ? fmaxf(v, 0.f) : v;
        }
    }
}

// EXT = false: the classic instantiation (host-drawn noise stream, zero seed) - the code the round-1 measurements
// belong to, kept instruction for instruction; EXT = true adds the in-kernel generator / noise dump / seed waveform.
template <class T>
__global__ __launch_bounds__(NT) void decode_bl6_kernel(const B6Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SEG = T::SEG, S = T::S, KIND = T::KIND;
    constexpr bool EXT = T::EXT;
    const int tid = threadIdx.x, b = blockIdx.x;
    const float* __restrict__ P = a.P;
    const int U = a.U;

    // ---- one-time loads: LDS constants, register-resident dilated-conv weights
    for (int e = tid; e < T::o_end; e += NT) lds[e] = 0.f;
    __syncthreads();
    for (int e = tid; e < L * 2 * H; e += NT) { lds[T::o_bx + e] = P[a.y.bx + e]; lds[T::o_bd + e] = P[a.y.bd + e]; }
    for (int e = tid; e < U; e += NT) lds[T::o_wup + e] = P[a.y.wup + e];
    for (int e = tid; e < H; e += NT) lds[T::o_cz + e] = P[a.y.cb + e];
    // head biases live in LDS: a global load consumed inside a phase would drain (in-order vmcnt) every
    // weight load that is deliberately in flight
    for (int e = tid; e < S; e += NT) lds[T::o_bias + e] = P[a.y.bsk + e];
    for (int e = tid; e < T::O1; e += NT) lds[T::o_bias + S + e] = P[a.y.b1 + e];
    if (KIND == SWN_KIND_SOFTMAX)
        for (int e = tid; e < T::NO; e += NT) lds[T::o_bias + S + T::O1 + e] = P[a.y.b2 + e];
    if (KIND == SWN_KIND_LAPLACE) {
        for (int e = tid; e < 2 * H; e += NT) { lds[T::o_cz + H + e] = P[a.y.cv + e]; lds[T::o_cz + 3 * H + e] = P[a.y.cc + e]; }
        for (int e = tid; e < T::NO * S; e += NT) lds[T::o_w2 + e] = P[a.y.w2 + (size_t)(e / S) * r4(S) + (e % S)];
        for (int e = tid; e < T::NO; e += NT) lds[T::o_w2 + T::NO * S + e] = P[a.y.b2 + e];
        for (int e = tid; e < T::W1L * T::O1 * 16; e += NT) lds[T::o_w1l + e] = P[a.y.w12 + e];
    }
    // conditioning frames are copied 16 B per lane through a buffer resource (32-bit offsets)
    const __amdgpu_buffer_rsrc_t condr =
        make_rsrc(a.cond + (size_t)b * a.Tf * a.N, (unsigned)((size_t)a.Tf * a.N * sizeof(float)));
    auto load_frame = [&](int fr) {
        float* dst = lds + T::o_gp + (fr & 1) * T::PF;
#pragma unroll
        for (int it = 0; it < (T::PF / 4 + NT - 1) / NT; ++it) {
            const int e4 = it * NT + tid;
            if (e4 < T::PF / 4)
                *reinterpret_cast<float4*>(dst + 4 * e4) =
                    buf_ld4(condr, (unsigned)tid * 16u, (unsigned)(fr * a.N * 4 + it * NT * 16));
        }
    };
    for (int fr = 0; fr < 2 && fr < a.Tf; ++fr) load_frame(fr);
    float wreg[L][2][16];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const float4* src = reinterpret_cast<const float4*>(P + a.y.wd2 + ((size_t)l * NT + tid) * 32);
#pragma unroll
        for (int v = 0; v < 8; ++v) {
            const float4 t = src[v];
            if (l < T::NREG) {
                wreg[l][v >> 2][4 * (v & 3) + 0] = t.x; wreg[l][v >> 2][4 * (v & 3) + 1] = t.y;
                wreg[l][v >> 2][4 * (v & 3) + 2] = t.z; wreg[l][v >> 2][4 * (v & 3) + 3] = t.w;
            } else {
                *reinterpret_cast<float4*>(lds + T::o_wl + (((l - T::NREG) * 8 + v) * NT + tid) * 4) = t;
                wreg[l][v >> 2][4 * (v & 3) + 0] = 0.f; wreg[l][v >> 2][4 * (v & 3) + 1] = 0.f;
                wreg[l][v >> 2][4 * (v & 3) + 2] = 0.f; wreg[l][v >> 2][4 * (v & 3) + 3] = 0.f;
            }
        }
    }
    __syncthreads();

    int fb = 0, tb = 0;                       // base conditioning frame resident in buffer fb&1
    const int p = tid & 7;
    const int n_pro = RF - SEG + 1;

    // conditioning taps of the position this lane finishes: upsampler weight and frame buffer offset
    // `single`: one position per pass (prologue, or seg == 1): every lane takes the taps of q0 itself,
    // because the split gate epilogue runs on lanes 0 AND 1 of an octet for that one position.
    auto cond_taps = [&](int q0, float (&wj)[SEG], int (&pb)[SEG], bool single) {
        const int t0 = q0 - RF;
        const int tlo = t0 < 0 ? 0 : t0;
        if (tlo >= tb + U) {                  // step crossed into the next frame: refill the free buffer
            fb += 1; tb += U;
            if (fb + 1 < a.Tf) load_frame(fb + 1);
        }
#pragma unroll
        for (int s = 0; s < SEG; ++s) {
            int tt = t0 + (single ? 0 : p) + s; tt = tt < 0 ? 0 : tt;
            int rel = tt - tb;
            int fsel = fb;
            if (rel >= U) { rel -= U; fsel = fb + 1; }
            rel = rel < U ? rel : U - 1;
            wj[s] = lds[T::o_wup + rel];
            pb[s] = (fsel & 1) * T::PF;
        }
    };

    // input layer h0 = softsign(causal(lift(S)))  (wave 0; fused wav_conv+causal taps).
    // Prologue form: all seed samples are zero / the mu-law zero index, only tap validity matters.
#define c_b  lds[T::o_cz + o]
#define c_v0 lds[T::o_cz + H + o]
#define c_v1 lds[T::o_cz + 2 * H + o]
#define c_c0 lds[T::o_cz + 3 * H + o]
#define c_c1 lds[T::o_cz + 4 * H + o]
    auto input_seed = [&](int q) {
        if (tid < H) {
            const int o = tid;
            constexpr int R0 = T::ring_len(0);
            float acc = c_b;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = q - (1 - k);
                if (KIND == SWN_KIND_LAPLACE) {
                    if (r >= -(SEG - 1)) acc += fmaf(k ? c_v1 : c_v0, 0.f, k ? c_c1 : c_c0);
                } else {
                    if (r >= 0) acc += P[a.y.ct + ((size_t)k * T::Q + T::Q / 2) * H + o];
                }
            }
            lds[T::o_ring + (q & (R0 - 1)) * H + o] = ssign(acc);
        }
    };
    // Generation form: the sample window lives in registers of wave 0 (uniform across its lanes):
    // win[] = S[qe-WN+1 .. qe], qe = last position whose input sample is known.
    constexpr int WNF = KIND == SWN_KIND_LAPLACE ? T::WN : 1, WNI = KIND == SWN_KIND_SOFTMAX ? T::WN : 1;
    float win[WNF];
    int iwin[WNI];
#pragma unroll
    for (int k = 0; k < WNF; ++k) win[k] = 0.f;
#pragma unroll
    for (int k = 0; k < WNI; ++k) iwin[k] = T::Q / 2;
    if (EXT && a.seed) {       // seed waveform `audio` of batch_fast_generate: the newest SEG samples / the newest class
        if constexpr (KIND == SWN_KIND_LAPLACE) {
#pragma unroll
            for (int j = 0; j < SEG; ++j) win[WNF - SEG + j] = reinterpret_cast<const float*>(a.seed)[(size_t)b * SEG + j];
        } else {
            iwin[WNI - 1] = reinterpret_cast<const int*>(a.seed)[b];
        }
    }
    auto input_gen = [&](int q0n) {
        if (tid < H) {
            const int o = tid;
            constexpr int R0 = T::ring_len(0);
#pragma unroll
            for (int j = 0; j < SEG; ++j) {
                float acc = c_b;
                if constexpr (KIND == SWN_KIND_LAPLACE) {
                    acc += fmaf(c_v0, win[T::WN - SEG + j - 1], c_c0);
                    acc += fmaf(c_v1, win[T::WN - SEG + j], c_c1);
                } else {
                    acc += P[a.y.ct + ((size_t)0 * T::Q + iwin[T::WN - SEG + j - 1]) * H + o];
                    acc += P[a.y.ct + ((size_t)1 * T::Q + iwin[T::WN - SEG + j]) * H + o];
                }
                lds[T::o_ring + ((q0n + j) & (R0 - 1)) * H + o] = ssign(acc);
            }
        }
    };
    // Sampling noise, staged in LDS off the critical path (host stream or in-kernel generator: swn_noise.hpp).
    // Laplace: wave 1 transforms the draws of a whole chunk of 64 future steps at once (lane = step):
    //   tn = sign(e) * log1p(-2|e|)   (cswnv_shift1.py:374-376); chunk c lives in buffer c & 3 and is filled two chunks
    //   ahead of its first reader, so the filler needs no barrier of its own.
    auto noise_chunk = [&](int c) {
        if (EXT && KIND == SWN_KIND_LAPLACE && tid >= 64 && tid < 128) {
            const int k = tid - 64, step = c * T::NZC + k;
            if (step < a.n_steps) {
#pragma unroll
                for (int j = 0; j < SEG; ++j) {
                    const float e = swn_noise_laplace(a.nz, b, step, j, a.n_steps, SEG);
                    const float sg = (e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f);
                    lds[T::o_tnz + ((c & (T::NZB - 1)) * T::NZC + k) * SEG + j] = sg * log1pf(-2.f * fabsf(e));
                }
            }
        }
    };
    // softmax: wave 2 fetches / draws the Q Exp(1) deviates of step `step` (lane t: classes 4t..4t+3) into buffer step & 1
    auto q_ahead = [&](int step) {
        if constexpr (KIND == SWN_KIND_SOFTMAX) {
            static_assert(T::Q == 256, "one float4 of classes per lane");
            if (tid >= 128 && tid < 192 && step < a.n_steps) {
                float4 q4;
                if constexpr (EXT) q4 = swn_noise_exp1x4(a.nz, b, step, tid - 128, a.n_steps, T::Q);
                else q4 = *reinterpret_cast<const float4*>(a.noise + ((size_t)b * a.n_steps + step) * T::Q + 4 * (tid - 128));
                *reinterpret_cast<float4*>(lds + T::o_tnz + (step & 1) * T::Q + 4 * (tid - 128)) = q4;
            }
        }
    };
    // classic mode: wave 1 transforms the deviate of the next step, its raw draw fetched one call earlier still
    float e_next = 0.f;
    auto noise_ahead = [&](int step) {
        if (!EXT && KIND == SWN_KIND_LAPLACE && tid >= 64 && tid < 64 + SEG) {
            const int j = tid - 64;
            if (step < a.n_steps) {
                const float e = e_next;
                const float sg = (e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f);
                lds[T::o_tnz + (step & 1) * 8 + j] = sg * log1pf(-2.f * fabsf(e));
            }
            if (step + 1 < a.n_steps) e_next = a.noise[((size_t)b * a.n_steps + step + 1) * SEG + j];
        }
    };
    if (!EXT && KIND == SWN_KIND_LAPLACE && tid >= 64 && tid < 64 + SEG && a.n_steps > 0)
        e_next = a.noise[(size_t)b * a.n_steps * SEG + (tid - 64)];
    noise_chunk(0);
    noise_chunk(1);
    q_ahead(0);

    // ---- prologue: seed positions 0..rf-seg, one position per pass (cswnv_shift1.py:321-334)
#pragma unroll 1
    for (int q = 0; q < n_pro; ++q) {
        float wj[SEG]; int pb[SEG];
        cond_taps(q, wj, pb, true);
        input_seed(q);
        lds_barrier();
        layer_phase<T, 0, 1>(lds, wreg[0], q, wj, pb); lds_barrier();
        layer_phase<T, 1, 1>(lds, wreg[1], q, wj, pb); lds_barrier();
        layer_phase<T, 2, 1>(lds, wreg[2], q, wj, pb); lds_barrier();
        layer_phase<T, 3, 1>(lds, wreg[3], q, wj, pb); lds_barrier();
        layer_phase<T, 4, 1>(lds, wreg[4], q, wj, pb); lds_barrier();
        layer_phase<T, 5, 1>(lds, wreg[5], q, wj, pb); lds_barrier();
    }

    // ---- generation (cswnv_shift1.py:348-402 / dswnv.py:338-374)
    const __amdgpu_buffer_rsrc_t wsk2 = make_rsrc(P + a.y.wsk2, (unsigned)(L * S * 64 * sizeof(float)));
    const __amdgpu_buffer_rsrc_t w12 = make_rsrc(P + a.y.w12, (unsigned)(T::O1 * S * sizeof(float)));
    const __amdgpu_buffer_rsrc_t w22 = make_rsrc(P + a.y.w22, (unsigned)(T::NO * T::O1 * sizeof(float)));
#ifdef SWN_STAMP
    // diagnostic build only (tools/stamp_decode.py): per-phase cycle sums of wave 0 leave the kernel
    // through the `heads` debug buffer, which this build writes nothing else into.
    unsigned long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = 0;
#define STAMP(k) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tacc[k] += tn - tprev; tprev = tn; }
#else
#define STAMP(k)
#endif
    input_gen(RF + 1 - SEG);
    noise_ahead(0);       // classic mode: visible to wave 0 after the barriers of step 0
    auto gen_step = [&](const int i) __attribute__((always_inline)) {
        const int q0 = RF + 1 - SEG + i * SEG;
        float wj[SEG]; int pb[SEG];
#ifdef SWN_STAMP
        tprev = __builtin_amdgcn_s_memtime();
#endif
        cond_taps(q0, wj, pb, SEG == 1);
        float sacc[S / 128];
#pragma unroll
        for (int ps = 0; ps < S / 128; ++ps) sacc[ps] = 0.f;
        [[maybe_unused]] float4 w1p[8];
        lds_barrier(); STAMP(0)
#ifdef SWN_NO_PREF
        if constexpr (false) {
#else
        if constexpr (S == 128 && SEG <= 2) {
#endif
            // out_skip weights are issued one phase ahead and stay in flight across the LDS-only barrier
            float4 wsl[4];
            layer_phase<T, 0, SEG>(lds, wreg[0], q0, wj, pb); skip_issue<T, 0>(wsk2, wsl); lds_barrier(); STAMP(1)
            layer_phase<T, 1, SEG>(lds, wreg[1], q0, wj, pb); skip_consume<T, 0>(lds, wsl, sacc); skip_issue<T, 1>(wsk2, wsl); lds_barrier(); STAMP(2)
            layer_phase<T, 2, SEG>(lds, wreg[2], q0, wj, pb); skip_consume<T, 1>(lds, wsl, sacc); skip_issue<T, 2>(wsk2, wsl); lds_barrier(); STAMP(3)
            layer_phase<T, 3, SEG>(lds, wreg[3], q0, wj, pb); skip_consume<T, 2>(lds, wsl, sacc); skip_issue<T, 3>(wsk2, wsl); lds_barrier(); STAMP(4)
            layer_phase<T, 4, SEG>(lds, wreg[4], q0, wj, pb); skip_consume<T, 3>(lds, wsl, sacc); skip_issue<T, 4>(wsk2, wsl); lds_barrier(); STAMP(5)
            layer_phase<T, 5, SEG>(lds, wreg[5], q0, wj, pb); skip_consume<T, 4>(lds, wsl, sacc); skip_issue<T, 5>(wsk2, wsl); lds_barrier(); STAMP(6)
            skip_consume<T, 5>(lds, wsl, sacc);
#ifdef SWN_W1PREF
            // out_1 weights: issued now, they fly during the reduction and the barrier
            {
                const int hr = tid >> 2, hp = tid & 3;
#pragma unroll
                for (int mm = 0; mm < 8; ++mm)
                    w1p[mm] = buf_ld4(w12, (unsigned)(hr * 4 + hp) * 16u, (unsigned)(mm * T::O1) * 64u);
            }
#endif
        } else {
            layer_phase<T, 0, SEG>(lds, wreg[0], q0, wj, pb); lds_barrier(); STAMP(1)
            layer_phase<T, 1, SEG>(lds, wreg[1], q0, wj, pb); skip_slice<T, 0>(lds, wsk2, sacc); lds_barrier(); STAMP(2)
            layer_phase<T, 2, SEG>(lds, wreg[2], q0, wj, pb); skip_slice<T, 1>(lds, wsk2, sacc); lds_barrier(); STAMP(3)
            layer_phase<T, 3, SEG>(lds, wreg[3], q0, wj, pb); skip_slice<T, 2>(lds, wsk2, sacc); lds_barrier(); STAMP(4)
            layer_phase<T, 4, SEG>(lds, wreg[4], q0, wj, pb); skip_slice<T, 3>(lds, wsk2, sacc); lds_barrier(); STAMP(5)
            layer_phase<T, 5, SEG>(lds, wreg[5], q0, wj, pb); skip_slice<T, 4>(lds, wsk2, sacc); lds_barrier(); STAMP(6)
            skip_slice<T, 5>(lds, wsk2, sacc);
        }
        {
            const int hr = tid >> 2, hp = tid & 3;
#pragma unroll
            for (int ps = 0; ps < S / 128; ++ps) {
                const float v = sum4(sacc[ps]);
                if (hp == 0) lds[T::o_skip + hr + 128 * ps] = fmaxf(v + lds[T::o_bias + hr + 128 * ps], 0.f);
            }
        }
        lds_barrier(); STAMP(7)
        // softmax: wave 2 stages the draws of step i+1 (into the buffer wave 0 read in step i-1's tail, which is behind
        // this step's barriers) while the out_1 weights are in flight
        noise_ahead(i + 1);
        q_ahead(i + 1);
#ifndef SWN_W1PREF
        if constexpr (false) {
#else
        if constexpr (S == 128 && SEG <= 2) {
#endif
            const int hr = tid >> 2, hp = tid & 3;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int mm = 0; mm < 8; mm += 2) {
                const float4 x0 = *reinterpret_cast<const float4*>(lds + T::o_skip + 16 * mm + 4 * hp);
                const float4 x1 = *reinterpret_cast<const float4*>(lds + T::o_skip + 16 * (mm + 1) + 4 * hp);
                a0 = fmaf(w1p[mm].x, x0.x, a0); a0 = fmaf(w1p[mm].y, x0.y, a0);
                a0 = fmaf(w1p[mm].z, x0.z, a0); a0 = fmaf(w1p[mm].w, x0.w, a0);
                a1 = fmaf(w1p[mm + 1].x, x1.x, a1); a1 = fmaf(w1p[mm + 1].y, x1.y, a1);
                a1 = fmaf(w1p[mm + 1].z, x1.z, a1); a1 = fmaf(w1p[mm + 1].w, x1.w, a1);
            }
            const float v = sum4(a0 + a1);
            if (hp == 0) lds[T::o_o1 + hr] = fmaxf(v + lds[T::o_bias + S + hr], 0.f);
        } else {
            if constexpr (T::W1L > 0) {
                // out_1, 128 x 128: the streamed slices are requested from L2 first, the resident ones come from LDS meanwhile
                constexpr int NG = S / 16 - T::W1L;
                const int hr = tid >> 2, hp = tid & 3;
                float4 wg[NG];
#pragma unroll
                for (int mm = 0; mm < NG; ++mm) wg[mm] = buf_ld4(w12, (unsigned)(hr * 4 + hp) * 16u, (unsigned)((T::W1L + mm) * T::O1) * 64u);
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int mm = 0; mm < T::W1L; ++mm) {
                    const float4 w0 = *reinterpret_cast<const float4*>(lds + T::o_w1l + ((mm * T::O1 + hr) * 4 + hp) * 4);
                    const float4 x0 = *reinterpret_cast<const float4*>(lds + T::o_skip + 16 * mm + 4 * hp);
                    float& acc = (mm & 1) ? a1 : a0;
                    acc = fmaf(w0.x, x0.x, acc); acc = fmaf(w0.y, x0.y, acc); acc = fmaf(w0.z, x0.z, acc); acc = fmaf(w0.w, x0.w, acc);
                }
#pragma unroll
                for (int mm = 0; mm < NG; ++mm) {
                    const float4 x0 = *reinterpret_cast<const float4*>(lds + T::o_skip + 16 * (T::W1L + mm) + 4 * hp);
                    float& acc = ((T::W1L + mm) & 1) ? a1 : a0;
                    acc = fmaf(wg[mm].x, x0.x, acc); acc = fmaf(wg[mm].y, x0.y, acc); acc = fmaf(wg[mm].z, x0.z, acc); acc = fmaf(wg[mm].w, x0.w, acc);
                }
                const float v = sum4(a0 + a1);
                if (hp == 0) lds[T::o_o1 + hr] = fmaxf(v + lds[T::o_bias + S + hr], 0.f);
            } else {
                tiled_matvec<T::O1, S>(w12, lds + T::o_bias + S, lds + T::o_skip, lds + T::o_o1, true);
            }
        }
        lds_barrier(); STAMP(8)

        if (KIND == SWN_KIND_LAPLACE) {
            if (tid < 64) {
                // out_2: NO <= 14 rows, 4 lanes per row, weights resident in LDS
                const int r = tid >> 2, pp = tid & 3;
                float acc = 0.f;
                if (r < T::NO) {
#pragma unroll
                    for (int mm = 0; mm < S / 16; ++mm) {
                        const float4 w = *reinterpret_cast<const float4*>(lds + T::o_w2 + r * S + 16 * mm + 4 * pp);
                        const float4 x = *reinterpret_cast<const float4*>(lds + T::o_o1 + 16 * mm + 4 * pp);
                        acc = fmaf(w.x, x.x, acc); acc = fmaf(w.y, x.y, acc);
                        acc = fmaf(w.z, x.z, acc); acc = fmaf(w.w, x.w, acc);
                    }
                }
                acc = sum4(acc);
                if (r < T::NO) acc += lds[T::o_w2 + T::NO * S + r];
                if (HEADS_ON && a.heads && pp == 0 && r < T::NO) a.heads[((size_t)b * a.n_steps + i) * T::NO + r] = acc;
                // park the NO head outputs in LDS; every lane re-reads them (same wave: LDS ops are in
                // order, the fences only pin the compiler)
                if (pp == 0 && r < T::NO) lds[T::o_o2 + r] = acc;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const float* o2 = lds + T::o_o2;
                {
#pragma clang fp contract(off)
                    // Laplace head, cswnv_shift1.py:368-391, evaluated uniformly by every lane of wave 0
                    // so the new samples are in registers for the next input layer (no LDS round trip).
                    float* outp = reinterpret_cast<float*>(a.out) + (size_t)b * a.n_steps * SEG + (size_t)i * SEG;
#pragma unroll
                    for (int j = 0; j < SEG; ++j) {
                        const float mu = o2[j];
                        const float bsc = sigm(o2[SEG + j]);                 // exp(logsigmoid(y))
                        float lpv = 0.f;
#pragma unroll
                        for (int k = 0; k < T::LPC; ++k) lpv += o2[2 * SEG + T::LPC - 1 - k] * win[T::WN - T::LPC + k];
                        
                        const float t = bsc * (EXT ? lds[T::o_tnz + (((i >> 6) & (T::NZB - 1)) * T::NZC + (i & (T::NZC - 1))) * SEG + j]
                                                   : lds[T::o_tnz + (i & 1) * 8 + j]);
                        float sv = (T::LPC > 0) ? (lpv + mu) - t : mu - t;
                        sv = fminf(fmaxf(sv, -1.f), 1.f);
                        if (tid == 0) outp[j] = sv;
                        const float fd = a.forced ? reinterpret_cast<const float*>(a.forced)[((size_t)b * a.n_steps + i) * SEG + j] : sv;
#pragma unroll
                        for (int k = 0; k + 1 < T::WN; ++k) win[k] = win[k + 1];
                        win[T::WN - 1] = fd;
                    }
                }
            }
        } else {
            constexpr int Q = T::Q;
            constexpr int QL = Q > 0 ? (Q + 63) / 64 : 1;     // classes per lane (this branch is also compiled for Laplace nets)
            tiled_matvec<T::NO, T::O1>(w22, lds + T::o_bias + S + T::O1, lds + T::o_o1, lds + T::o_o2, false);
            lds_barrier();
            if (HEADS_ON && a.heads)
                for (int e = tid; e < T::NO; e += NT) a.heads[((size_t)b * a.n_steps + i) * T::NO + e] = lds[T::o_o2 + e];
            if (tid < 64) {
                // softmax head, dswnv.py:361-369: p = softmax(logits); p /= sum(p); index = argmax(p / q).
                // lane t owns classes 4t..4t+3 (one 16-byte LDS read each for logits and noise); exp(logit - max) is
                // evaluated once per class and kept in registers for the three passes.
                static_assert(KIND != SWN_KIND_SOFTMAX || QL == 4, "four classes per lane");
                const float4 l4 = *reinterpret_cast<const float4*>(lds + T::o_o2 + 4 * tid);
                const float4 q4 = *reinterpret_cast<const float4*>(lds + T::o_tnz + (i & 1) * Q + 4 * tid);
                const float lg[4] = {l4.x, l4.y, l4.z, l4.w}, qv[4] = {q4.x, q4.y, q4.z, q4.w};
                float ex[4];
                float m = fmaxf(fmaxf(lg[0], lg[1]), fmaxf(lg[2], lg[3]));
                for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
                float sum = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) { ex[k] = expf(lg[k] - m); sum += ex[k]; }
                for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
                float sum2 = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) { ex[k] = ex[k] / sum; sum2 += ex[k]; }
                for (int d = 32; d >= 1; d >>= 1) sum2 += __shfl_xor(sum2, d, 64);
                float best = -1.f; int bi = 0x7fffffff;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float r = (ex[k] / sum2) / qv[k];
                    if (r > best) { best = r; bi = 4 * tid + k; }
                }
                for (int d = 32; d >= 1; d >>= 1) {
                    const float ob = __shfl_xor(best, d, 64);
                    const int oi = __shfl_xor(bi, d, 64);
                    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
                }
                if (tid == 0) reinterpret_cast<int*>(a.out)[(size_t)b * a.n_steps + i] = bi;
                const int fd = a.forced ? reinterpret_cast<const int*>(a.forced)[(size_t)b * a.n_steps + i] : bi;
#pragma unroll
                for (int k = 0; k + 1 < T::WN; ++k) iwin[k] = iwin[k + 1];
                iwin[T::WN - 1] = fd;
            }
        }
        // next step's input layer, still inside wave 0 (no barrier between sampling and h0)
        if (i + 1 < a.n_steps) input_gen(q0 + SEG);
        STAMP(9)
    };
    if constexpr (EXT) {
        // chunks of 64 steps: wave 1 fills the noise of chunk c+2 at the top of chunk c (buffer (c+2)&3 held chunk c-2,
        // whose readers are two chunks behind)
#pragma unroll 1
        for (int i0 = 0; i0 < a.n_steps; i0 += T::NZC) {
            noise_chunk((i0 >> 6) + 2);
            const int iend = i0 + T::NZC < a.n_steps ? i0 + T::NZC : a.n_steps;
#pragma unroll 1
            for (int i = i0; i < iend; ++i) gen_step(i);
        }
    } else {
#pragma unroll 1
        for (int i = 0; i < a.n_steps; ++i) gen_step(i);
    }
#ifdef SWN_STAMP
    if (tid == 0 && b == 0 && a.heads)
        for (int k = 0; k < 10; ++k) a.heads[k] = (float)((double)tacc[k] / (double)a.n_steps);
#endif
}

template <class T>
int launch_mode(const B6Args& a, hipStream_t st) {
    static_assert(T::lds_bytes <= 160 * 1024, "LDS budget");
    auto kern = decode_bl6_kernel<T>;
    if (T::lds_bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)T::lds_bytes) != hipSuccess)
            return SWN_E_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(NT), T::lds_bytes, st, a);
    return swn_launch_status("swn_decode(bl6)");
}

template <class T>
int launch(const B6Args& a, hipStream_t st) {
    const bool ext = !a.nz.ptr || a.nz.dump || a.seed;
    return ext ? launch_mode<typename T::Ext>(a, st) : launch_mode<T>(a, st);
}

}  // namespace

extern "C" int swn_decode_bl6_try(const swn_net_desc* d, const float* packed, const float* cond, int batch,
                                  int n_frames, int n_steps, const SwnNoise* nz, const void* forced,
                                  const void* seed, void* out, float* heads, void* stream_) {
    SwnGeom g; int rc = swn_make_geom(d, &g);
    if (rc < 0) return rc;
    if (!g.bl6 || g.U > 256 || g.U < 2 * g.seg || g.audio_in) return SWN_E_UNSUPPORTED;
    B6Args a;
    swn_make_layout(&g, &a.y);
    a.P = packed; a.cond = cond; a.noise = nz->ptr; a.nz = *nz; a.forced = forced; a.seed = seed; a.out = out; a.heads = heads;
    a.B = batch; a.Tf = n_frames; a.n_steps = n_steps; a.U = g.U; a.N = g.N;
    hipStream_t st = (hipStream_t)stream_;
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls; only our launches are reported
    if (g.kind == SWN_KIND_LAPLACE && g.S == 128) {
        if (g.seg == 1 && g.lpc == 0) return launch<Tr<128, 1, 0, SWN_KIND_LAPLACE, 0>>(a, st);
        if (g.seg == 1 && g.lpc == 4) return launch<Tr<128, 1, 4, SWN_KIND_LAPLACE, 0>>(a, st);
        if (g.seg == 2 && g.lpc == 4) return launch<Tr<128, 2, 4, SWN_KIND_LAPLACE, 0>>(a, st);
        if (g.seg == 5 && g.lpc == 0) return launch<Tr<128, 5, 0, SWN_KIND_LAPLACE, 0>>(a, st);
        if (g.seg == 5 && g.lpc == 4) return launch<Tr<128, 5, 4, SWN_KIND_LAPLACE, 0>>(a, st);
    }
    if (g.kind == SWN_KIND_SOFTMAX && g.S == 256 && g.Q == 256)
        return launch<Tr<256, 1, 0, SWN_KIND_SOFTMAX, 256>>(a, st);
    return SWN_E_UNSUPPORTED;
}
