// BL6-class autoregressive decode on gfx950, WAVE-SPECIALISED form (single-sample Laplace nets: 1x6 dilated stack, H = 64, K = 2,
// S = O1 = 128, seg = 1, lpc 0 / 4).
//
// swn_decode_bl6.hip gives all eight waves of the workgroup the same program: between two barriers the two waves of a SIMD run the
// same dependent chain (LDS read -> 16-deep FMA chains -> DPP reduction -> transcendentals -> LDS write) and stall in the same
// places, so a SIMD takes the time of two waves that do not overlap (9 900 cycles per step against 2 400 of VALU
// issue per wave).  A conv of kernel size 2 offers the asymmetry that lets them overlap: a layer's pre-activation is
//     W[:, tap 1] h_{l-1}(t)  +  W[:, tap 0] h_{l-1}(t - dil)  +  b
// and the second product does not depend on the step in progress - for every layer it is known one step ahead (dil >= 1).  So:
//   * group A (waves 0-3, one per SIMD) owns the chain: per layer the CURRENT-tap half (128 rows x 64 inputs: thread (o, p) two
//     rows x 16 inputs, reduced over 4 lanes), the gate epilogue, the hand-off through LDS;
//   * group B (waves 4-7, the other wave of every SIMD) works one step AHEAD and off the chain: the OLDER-tap halves (+ bias) of
//     all six layers for step t + 1, left in LDS (formed where group B has room: one in the first layer phase, three beside group
//     A's out_1, two beside wave 0's tail), and the whole out_skip accumulation of step t (slice l in the phase after layer l);
//   * group A keeps its halves of all six matrices in registers (192 per thread), group B five of them (160) and the sixth in LDS
//     (it also holds a slice of out_skip weights in flight); the out_1 matrix is LDS-resident whole (the 64 KB the sixth layer of
//     the symmetric kernel took), only out_skip streams from L2;
//   * out_1 is group A's alone (two rows per thread), out_2 + sampling + the next input layer wave 0's, as before.
// Same arithmetic per element as the symmetric kernel up to the order of the partial sums (1e-7 relative); same noise, seed,
// forced-input and heads interface (swn_decode_bl6.hip: classic = host-drawn noise, zero seed; extended = in-kernel generator,
// noise dump, caller's seed waveform).  9 barriers per step.  cswnv_shift1.py:281-430.
#include <hip/hip_runtime.h>
#include "swn_geom.hpp"
#include "swn_noise.hpp"

namespace {

constexpr int NT = 512;
constexpr int NG = 256;            // threads per group
constexpr int H = 64;
constexpr int L = 6;
constexpr int RF = 64;
constexpr int S = 128;
constexpr int O1 = 128;
#ifdef SWN_STAMP
constexpr bool HEADS_ON = false;
#else
constexpr bool HEADS_ON = true;
#endif

struct W6Args {
    const float* P;
    SwnLayout y;
    const float* cond;
    const float* noise;            // classic mode: the host-drawn stream
    const void* forced;
    void* out;
    float* heads;
    int B, Tf, n_steps, U, N;
    SwnNoise nz;                   // extended mode
    const void* seed;
};

constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int pow2ceil(int x) { int r = 1; while (r < x) r <<= 1; return r; }
constexpr int r4(int x) { return (x + 3) & ~3; }

template <int LPC_, bool EXT_>
struct Tw {
    static constexpr int LPC = LPC_;
    static constexpr bool EXT = EXT_;
    using Ext = Tw<LPC_, true>;
    static constexpr int NO = 2 + LPC_;
    static constexpr int WN = cmax(1, LPC_) + 1;
    static constexpr int ring_len(int l) { return pow2ceil((1 << l) + 1); }
    static constexpr int ring_off(int l) { int o = 0; for (int i = 0; i < l; ++i) o += ring_len(i) * H; return o; }
    static constexpr int PF = L * 2 * H;                  // floats of one conditioning frame
    static constexpr int NZC = 64, NZB = 4;
    // LDS carve (float offsets); everything the step loop addresses with immediates sits below 64 KB, the out_1 matrix last
    static constexpr int o_ring = 0;
    static constexpr int o_hcat = o_ring + ring_off(L);
    static constexpr int o_gp = o_hcat + L * H;
    static constexpr int o_bx = o_gp + 2 * PF;
    static constexpr int o_bd = o_bx + L * 2 * H;
    static constexpr int o_old = o_bd + L * 2 * H;         // [2][L][2H]: older-tap products + bias of the step in progress / the next
    static constexpr int o_wup = o_old + 2 * L * 2 * H;
    static constexpr int o_skip = o_wup + 256;
    static constexpr int o_o1 = o_skip + S;
    static constexpr int o_o2 = o_o1 + O1;
    static constexpr int o_tnz = o_o2 + r4(NO);
    static constexpr int o_cz = o_tnz + (EXT_ ? r4(NZB * NZC) : 16);   // cb[64], cv[2][64], cc[2][64]
    static constexpr int o_w2 = o_cz + 5 * H;              // out_2 rows [NO][S] (+b2)
    static constexpr int o_bias = o_w2 + NO * S + r4(NO);  // bsk[S], b1[O1]
    static constexpr int o_w1 = o_bias + S + O1;           // out_1, lane-tiled [8][O1][4][4] like the global copy w12
    static constexpr int o_wl = o_w1 + 8 * O1 * 16;        // group B's half of the LAST layer, [8][256 threads][4]: with it in registers
                                                           // too (192 + the out_skip weights in flight) the kernel spilled 22-35 registers
    static constexpr int o_end = o_wl + 8 * NG * 4;
    static constexpr size_t lds_bytes = (size_t)o_end * sizeof(float);
};

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
// exp, reciprocal, sigmoid as in swn_decode_bl6.hip (<= 1.5 ulp: the 1e-5 bar is held through 66 000 recurrent steps)
__device__ __forceinline__ float exp_c(float x) {
    const float t = x * 1.44269504f;
    const float lo = fmaf(x, 1.44269504f, -t) + x * 1.92596299e-8f;
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, lo * 0.693147181f, e);
}
__device__ __forceinline__ float rcp_c(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(r, fmaf(-x, r, 1.f), r);
}
__device__ __forceinline__ float sigm(float x) { return rcp_c(1.f + exp_c(-x)); }
__device__ __forceinline__ float ssign(float x) { return x * rcp_c(1.f + fabsf(x)); }

__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- group A: layer LAYER at position q - current tap, gate epilogue, hand-off.  Thread (o, p): rows o (gate) and o + 64
//      (candidate) over inputs 16 p .. 16 p + 15 of h_{l-1}(q); lanes 0 / 1 of the quad finish the gate / the candidate.
template <class T, int LAYER>
__device__ __forceinline__ void layer_a(float* lds, const float (&w)[2][16], const int q, const float wj, const int pb, const int ta) {
    constexpr int R = T::ring_len(LAYER);
    const int o = ta >> 2, p = ta & 3, pr = p & 1;
    const float* ring = lds + T::o_ring + T::ring_off(LAYER);
    // epilogue operands first: their LDS latency hides under the FMAs
    const float e_old = lds[T::o_old + (q & 1) * (L * 2 * H) + LAYER * 2 * H + pr * H + o];
    const float e_gx = fmaf(wj, lds[T::o_gp + pb + LAYER * 2 * H + pr * H + o], lds[T::o_bx + LAYER * 2 * H + pr * H + o]);
    const float e_hp = ring[(q & (R - 1)) * H + o];
    float az = 0.f, ac = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float4 x = *reinterpret_cast<const float4*>(ring + (q & (R - 1)) * H + 16 * p + 4 * m);
        az = fmaf(w[0][4 * m], x.x, az); ac = fmaf(w[1][4 * m], x.x, ac);
        az = fmaf(w[0][4 * m + 1], x.y, az); ac = fmaf(w[1][4 * m + 1], x.y, ac);
        az = fmaf(w[0][4 * m + 2], x.z, az); ac = fmaf(w[1][4 * m + 2], x.z, ac);
        az = fmaf(w[0][4 * m + 3], x.w, az); ac = fmaf(w[1][4 * m + 3], x.w, ac);
    }
    az = sum4(az); ac = sum4(ac);
    __builtin_amdgcn_sched_barrier(0);              // keep the gate epilogue behind the reduction
    if (p < 2) {
        const float sa = (p == 0 ? az : ac) + e_old;
        const float v = e_gx * sa;
        float res;
        if (fabsf(v) > 9.02f && p == 1) {
            res = copysignf(1.f, v);                                       // tanh saturated in fp32
        } else {
            // (plain v_exp_f32 / v_rcp_f32 without the corrections - nine instructions fewer on the chain - measured the same step
            //  time: the phase waits for both groups, not for this arithmetic)
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
    __builtin_amdgcn_sched_barrier(0);
}

// ---- group B: older-tap product of layer LAYER for position qn = q + 1 (its operand h_{l-1}(qn - dil) is at least one step old),
//      bias added, left for group A in the buffer of qn's parity
template <class T, int LAYER>
__device__ __forceinline__ void older_sum(const float* lds, const float (&w)[2][16], const int qn, const int tb, float& az, float& ac) {
    constexpr int dil = 1 << LAYER;
    constexpr int R = T::ring_len(LAYER);
    const int p = tb & 3;
    const float* ring = lds + T::o_ring + T::ring_off(LAYER);
    az = 0.f; ac = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float4 x = *reinterpret_cast<const float4*>(ring + ((qn - dil) & (R - 1)) * H + 16 * p + 4 * m);
        float4 w0, w1;
        if (LAYER < L - 1) {
            w0 = make_float4(w[0][4 * m], w[0][4 * m + 1], w[0][4 * m + 2], w[0][4 * m + 3]);
            w1 = make_float4(w[1][4 * m], w[1][4 * m + 1], w[1][4 * m + 2], w[1][4 * m + 3]);
        } else {                                              // the last layer's half comes from LDS
            w0 = *reinterpret_cast<const float4*>(lds + T::o_wl + (m * NG + tb) * 4);
            w1 = *reinterpret_cast<const float4*>(lds + T::o_wl + ((4 + m) * NG + tb) * 4);
        }
        az = fmaf(w0.x, x.x, az); ac = fmaf(w1.x, x.x, ac);
        az = fmaf(w0.y, x.y, az); ac = fmaf(w1.y, x.y, ac);
        az = fmaf(w0.z, x.z, az); ac = fmaf(w1.z, x.z, ac);
        az = fmaf(w0.w, x.w, az); ac = fmaf(w1.w, x.w, ac);
    }
    az = sum4(az); ac = sum4(ac);
}
template <class T, int LAYER>
__device__ __forceinline__ void older_store(float* lds, const int qn, const int tb, const float az, const float ac) {
    const int o = tb >> 2, p = tb & 3, pr = p & 1;            // (called under p < 2)
    lds[T::o_old + (qn & 1) * (L * 2 * H) + LAYER * 2 * H + pr * H + o] = (p == 0 ? az : ac) + lds[T::o_bd + LAYER * 2 * H + pr * H + o];
}
template <class T, int LAYER>
__device__ __forceinline__ void older_b(float* lds, const float (&w)[2][16], const int qn, const int tb) {
    float az, ac;
    older_sum<T, LAYER>(lds, w, qn, tb, az, ac);
    if ((tb & 3) < 2) older_store<T, LAYER>(lds, qn, tb, az, ac);
}
// three layers at once: one basic block of three independent chains (the stores' lane branch between them kept the scheduler from
// overlapping the products: 570 cycles each, one after the other)
template <class T, int LA>
__device__ __forceinline__ void older_b2(float* lds, const float (&w)[L][2][16], const int qn, const int tb) {
    float az0, ac0, az1, ac1;
    older_sum<T, LA>(lds, w[LA], qn, tb, az0, ac0);
    older_sum<T, LA + 1>(lds, w[LA + 1], qn, tb, az1, ac1);
    if ((tb & 3) < 2) {
        older_store<T, LA>(lds, qn, tb, az0, ac0);
        older_store<T, LA + 1>(lds, qn, tb, az1, ac1);
    }
}
template <class T, int LA>
__device__ __forceinline__ void older_b3(float* lds, const float (&w)[L][2][16], const int qn, const int tb) {
    float az0, ac0, az1, ac1, az2, ac2;
    older_sum<T, LA>(lds, w[LA], qn, tb, az0, ac0);
    older_sum<T, LA + 1>(lds, w[LA + 1], qn, tb, az1, ac1);
    older_sum<T, LA + 2>(lds, w[LA + 2], qn, tb, az2, ac2);
    if ((tb & 3) < 2) {
        older_store<T, LA>(lds, qn, tb, az0, ac0);
        older_store<T, LA + 1>(lds, qn, tb, az1, ac1);
        older_store<T, LA + 2>(lds, qn, tb, az2, ac2);
    }
}

// ---- group B: out_skip, slice by slice.  Thread (r, hp): rows r and r + 64 over inputs 16 mm + 4 hp .. + 3 of the slice (the
//      lane-tiled global copy wsk2 of the symmetric kernel: 1 KiB contiguous per wave instruction); weights issued a phase ahead.
// (`step0` is an opaque zero the caller renews every step: the addresses are loop-invariant, and hoisted out of the step loop the
//  six slices - 192 registers - would be kept resident beside the 160 weight registers)
template <int LAYER>
__device__ __forceinline__ void skip_issue_b(__amdgpu_buffer_rsrc_t wsk2, float4 (&wsl)[8], const int tb, const unsigned step0) {
    const int r = tb >> 2, hp = tb & 3;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int mm = 0; mm < 4; ++mm)
            wsl[ps * 4 + mm] = buf_ld4(wsk2, (unsigned)((r + 64 * ps) * 4 + hp) * 16u, (unsigned)((LAYER * 4 + mm) * S) * 64u + step0);
}
template <class T, int LAYER>
__device__ __forceinline__ void skip_consume_b(const float* lds, const float4 (&wsl)[8], float (&sacc)[2], const int tb) {
    const int hp = tb & 3;
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
        const float4 x = *reinterpret_cast<const float4*>(lds + T::o_hcat + LAYER * H + 16 * mm + 4 * hp);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const float4 w = wsl[ps * 4 + mm];
            sacc[ps] = fmaf(w.x, x.x, sacc[ps]); sacc[ps] = fmaf(w.y, x.y, sacc[ps]);
            sacc[ps] = fmaf(w.z, x.z, sacc[ps]); sacc[ps] = fmaf(w.w, x.w, sacc[ps]);
        }
    }
    // pin the partial sums to this phase: only the end of the step uses them, and the optimiser otherwise sinks all six slices'
    // multiply-adds down to that use - with their 48 operand registers per slice live until then (238 registers spilled)
    asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]));
}

// The program of one group (GA: group A).  The two groups run the SAME sequence of barriers; they are separate instantiations -
// not one body with a run-time branch per phase - so that each has its own register allocation: as one body the allocation was the
// union (192 weight registers of A + the out_skip weights B keeps in flight: 22-65 registers spilled inside the step loop).
template <class T, bool GA>
__device__ __forceinline__ void decode_body(const W6Args& a, float* lds) {
    constexpr bool EXT = T::EXT;
    constexpr bool grpA = GA;
    const int tid = threadIdx.x, b = blockIdx.x;
    __builtin_assume(GA ? tid < NG : tid >= NG);              // group B compiles none of wave 0's and wave 1's side jobs (head, noise, input layer)
    // (a raised s_setprio for group A - the chain first at the SIMD's issue arbitration - measured no gain: 283.6 / 278.8 / 282.9 k
    //  samples/s at priority 0 / 1 / 3)
    const int tg = tid & (NG - 1);                            // index inside the group
    const float* __restrict__ P = a.P;
    const int U = a.U;

    // ---- one-time loads
    for (int e = tid; e < T::o_end; e += NT) lds[e] = 0.f;
    __syncthreads();
    for (int e = tid; e < L * 2 * H; e += NT) {
        lds[T::o_bx + e] = P[a.y.bx + e];
        const float bd = P[a.y.bd + e];
        lds[T::o_bd + e] = bd;
        lds[T::o_old + e] = bd;                               // position 0: the older tap reads the zero padding
    }
    for (int e = tid; e < U; e += NT) lds[T::o_wup + e] = P[a.y.wup + e];
    for (int e = tid; e < H; e += NT) lds[T::o_cz + e] = P[a.y.cb + e];
    for (int e = tid; e < S; e += NT) lds[T::o_bias + e] = P[a.y.bsk + e];
    for (int e = tid; e < O1; e += NT) lds[T::o_bias + S + e] = P[a.y.b1 + e];
    for (int e = tid; e < 2 * H; e += NT) { lds[T::o_cz + H + e] = P[a.y.cv + e]; lds[T::o_cz + 3 * H + e] = P[a.y.cc + e]; }
    for (int e = tid; e < T::NO * S; e += NT) lds[T::o_w2 + e] = P[a.y.w2 + (size_t)(e / S) * r4(S) + (e % S)];
    for (int e = tid; e < T::NO; e += NT) lds[T::o_w2 + T::NO * S + e] = P[a.y.b2 + e];
    for (int e = tid; e < 8 * O1 * 16; e += NT) lds[T::o_w1 + e] = P[a.y.w12 + e];
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
    // register-resident halves of the six matrices: group A tap 1 (current), group B tap 0 (older); rows (o, o + 64), inputs 16 p ..
    float wreg[L][2][16];                                     // (group B: the last layer's entries stay unused - its half is in LDS)
    {
        const int o = tg >> 2, p = tg & 3, k = grpA ? 1 : 0;
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float4* src = reinterpret_cast<const float4*>(P + a.y.wd + (((size_t)l * 2 * H + o + r * H) * 2 + k) * H + 16 * p);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float4 t = src[v];
                    if (l == L - 1 && !grpA) {                // group B keeps its half of the last layer in LDS
                        *reinterpret_cast<float4*>(lds + T::o_wl + ((r * 4 + v) * NG + tg) * 4) = t;
                        wreg[l][r][4 * v] = 0.f; wreg[l][r][4 * v + 1] = 0.f; wreg[l][r][4 * v + 2] = 0.f; wreg[l][r][4 * v + 3] = 0.f;
                    } else {
                        wreg[l][r][4 * v] = t.x; wreg[l][r][4 * v + 1] = t.y; wreg[l][r][4 * v + 2] = t.z; wreg[l][r][4 * v + 3] = t.w;
                    }
                }
            }
    }
    __syncthreads();

    int fb = 0, tb0 = 0;                      // base conditioning frame resident in buffer fb & 1
    const int n_pro = RF;                     // seed positions 0 .. rf - 1 (seg = 1)
    auto cond_taps = [&](int q, float& wj, int& pb) {
        const int t0 = q - RF;
        const int tlo = t0 < 0 ? 0 : t0;
        if (tlo >= tb0 + U) {                 // step crossed into the next frame: refill the free buffer
            fb += 1; tb0 += U;
            if (fb + 1 < a.Tf) load_frame(fb + 1);
        }
        int rel = tlo - tb0;
        int fsel = fb;
        if (rel >= U) { rel -= U; fsel = fb + 1; }
        rel = rel < U ? rel : U - 1;
        wj = lds[T::o_wup + rel];
        pb = (fsel & 1) * T::PF;
    };

#define c_b  lds[T::o_cz + o]
#define c_v0 lds[T::o_cz + H + o]
#define c_v1 lds[T::o_cz + 2 * H + o]
#define c_c0 lds[T::o_cz + 3 * H + o]
#define c_c1 lds[T::o_cz + 4 * H + o]
    // input layer h0 = softsign(causal(lift(S)))  (wave 0).  Prologue: the seed samples are zero, only tap validity matters
    auto input_seed = [&](int q) {
        if (tid < H) {
            const int o = tid;
            float acc = c_b;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int r = q - (1 - k);
                if (r >= 0) acc += fmaf(k ? c_v1 : c_v0, 0.f, k ? c_c1 : c_c0);
            }
            lds[T::o_ring + (q & 1) * H + o] = ssign(acc);
        }
    };
    float win[T::WN];
#pragma unroll
    for (int k = 0; k < T::WN; ++k) win[k] = 0.f;
    if (EXT && a.seed) win[T::WN - 1] = reinterpret_cast<const float*>(a.seed)[b];
    // (the five input-layer constants of wave 0's channel stay in registers: the tail phase is one wave's dependent chain)
    float kcb = 0.f, kv0 = 0.f, kv1 = 0.f, kc0 = 0.f, kc1 = 0.f;
    if (tid < H) { const int o = tid; kcb = c_b; kv0 = c_v0; kv1 = c_v1; kc0 = c_c0; kc1 = c_c1; }
    auto input_gen = [&](int qn) {
        if (tid < H) {
            float acc = kcb;
            acc += fmaf(kv0, win[T::WN - 2], kc0);
            acc += fmaf(kv1, win[T::WN - 1], kc1);
            lds[T::o_ring + (qn & 1) * H + tid] = ssign(acc);
        }
    };
    // sampling noise staged off the chain (swn_decode_bl6.hip): wave 1 transforms tn = sign(e) log1p(-2|e|)
    auto noise_chunk = [&](int c) {
        if (EXT && tid >= 64 && tid < 128) {
            const int k = tid - 64, step = c * T::NZC + k;
            if (step < a.n_steps) {
                const float e = swn_noise_laplace(a.nz, b, step, 0, a.n_steps, 1);
                const float sg = (e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f);
                lds[T::o_tnz + (c & (T::NZB - 1)) * T::NZC + k] = sg * log1pf(-2.f * fabsf(e));
            }
        }
    };
    float e_next = 0.f;
    auto noise_ahead = [&](int step) {
        if (!EXT && tid == 64) {
            if (step < a.n_steps) {
                const float e = e_next;
                const float sg = (e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f);
                lds[T::o_tnz + (step & 1) * 8] = sg * log1pf(-2.f * fabsf(e));
            }
            if (step + 1 < a.n_steps) e_next = a.noise[(size_t)b * a.n_steps + step + 1];
        }
    };
    if (!EXT && tid == 64 && a.n_steps > 0) e_next = a.noise[(size_t)b * a.n_steps];
    noise_chunk(0);
    noise_chunk(1);

    const __amdgpu_buffer_rsrc_t wsk2 = make_rsrc(P + a.y.wsk2, (unsigned)(L * S * 64 * sizeof(float)));

#ifdef SWN_STAMP
    // diagnostic build only (tools/stamp_decode_w.py): per phase, the cycles each group's first wave WORKS between two barriers
    // (barrier waits excluded) leave through the `heads` debug buffer: [0..9) group A, [9] A's whole step, [10..19) group B
    unsigned long long tw[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tstep = 0, tlast = 0, tbeg = 0;
#define SWN_BAR(k) { tw[k] += __builtin_amdgcn_s_memtime() - tlast; lds_barrier(); tlast = __builtin_amdgcn_s_memtime(); }
#else
#define SWN_BAR(k) lds_barrier();
#endif
    // one layer phase: group A runs the chain, group B prepares position q + 1 (and, during generation, accumulates out_skip)
    // (OLD: the layer whose older-tap product for position q + 1 group B forms in this phase, -1 = none)
#define SWN_PHASE(LAYER, GEN, OLD)                                                                                    \
    if constexpr (grpA) layer_a<T, LAYER>(lds, wreg[LAYER], q, wj, pb, tg);                                            \
    else {                                                                                                             \
        if (GEN) {   /* the slice in flight is consumed BEFORE the next one is requested (both live at once are 64 registers  */ \
                     /* beside the 160 of the weights); the request then flies under the older-tap product and the barrier    */ \
            if (LAYER > 0) skip_consume_b<T, (LAYER > 0 ? LAYER - 1 : 0)>(lds, wsl, sacc, tg);                           \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            skip_issue_b<LAYER>(wsk2, wsl, tg, step0);                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
        }                                                                                                              \
        if (OLD >= 0) older_b<T, (OLD >= 0 ? OLD : 0)>(lds, wreg[OLD >= 0 ? OLD : 0], q + 1, tg);                        \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }                                                                                                                  \
    if (GEN) { SWN_BAR(LAYER) } else lds_barrier();

    // ---- prologue: seed positions 0 .. rf - 1 (cswnv_shift1.py:321-334)
#pragma unroll 1
    for (int q = 0; q < n_pro; ++q) {
        float wj; int pb;
        float4 wsl[8]; float sacc[2]; const unsigned step0 = 0;
        cond_taps(q, wj, pb);
        input_seed(q);
        lds_barrier();
        SWN_PHASE(0, false, 0) SWN_PHASE(1, false, 1) SWN_PHASE(2, false, 2) SWN_PHASE(3, false, 3) SWN_PHASE(4, false, 4) SWN_PHASE(5, false, 5)
        (void)wsl; (void)sacc; (void)step0;
    }

    // ---- generation (cswnv_shift1.py:348-402)
    input_gen(RF);
    noise_ahead(0);
    auto gen_step = [&](const int i) __attribute__((always_inline)) {
        const int q = RF + i;
        float wj; int pb;
        cond_taps(q, wj, pb);
        float4 wsl[8];
        float sacc[2] = {0.f, 0.f};
        unsigned step0 = 0;
        asm volatile("" : "+s"(step0));                       // opaque zero, see skip_issue_b
        lds_barrier();
#ifdef SWN_STAMP
        tlast = tbeg = __builtin_amdgcn_s_memtime();
#endif
        // group B's older-tap products of position q + 1 sit in the out_1 phase (layers 0-2: group A runs out_1 alone), in the tail
        // phase (layers 3-4: only wave 0 has work there) and in the first layer phase (layer 5: no slice to consume yet); all six
        // beside the out_skip slices made group B take 1 050 cycles per layer phase against group A's 710
        SWN_PHASE(0, true, 5) SWN_PHASE(1, true, -1) SWN_PHASE(2, true, -1) SWN_PHASE(3, true, -1) SWN_PHASE(4, true, -1) SWN_PHASE(5, true, -1)
        // out_skip: the last slice and the reduction (group B); group A has nothing on this phase
        if constexpr (!grpA) {
            skip_consume_b<T, 5>(lds, wsl, sacc, tg);
            const int r = tg >> 2, hp = tg & 3;
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const float v = sum4(sacc[ps]);
                if (hp == 0) lds[T::o_skip + r + 64 * ps] = fmaxf(v + lds[T::o_bias + r + 64 * ps], 0.f);
            }
        }
        SWN_BAR(6)
        noise_ahead(i + 1);
        if constexpr (grpA) {
            // out_1, 128 x 128, LDS-resident, by group A alone: thread (hr, hp) rows hr and hr + 64 over inputs 16 mm + 4 hp .. + 3 - one
            // read of the inputs serves both rows.  Group B meanwhile forms three of the older-tap products of position q + 1 (with
            // all eight waves on out_1 the phase took 1 020 cycles, and the six products had to sit beside the out_skip slices)
            const int hr = tg >> 2, hp = tg & 3;
            float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
            for (int mm = 0; mm < 8; ++mm) {
                const float4 x0 = *reinterpret_cast<const float4*>(lds + T::o_skip + 16 * mm + 4 * hp);
                const float4 w0 = *reinterpret_cast<const float4*>(lds + T::o_w1 + ((mm * O1 + hr) * 4 + hp) * 4);
                const float4 w1 = *reinterpret_cast<const float4*>(lds + T::o_w1 + ((mm * O1 + hr + 64) * 4 + hp) * 4);
                float& acc = (mm & 1) ? a1 : a0;
                float& bcc = (mm & 1) ? b1 : b0;
                acc = fmaf(w0.x, x0.x, acc); acc = fmaf(w0.y, x0.y, acc); acc = fmaf(w0.z, x0.z, acc); acc = fmaf(w0.w, x0.w, acc);
                bcc = fmaf(w1.x, x0.x, bcc); bcc = fmaf(w1.y, x0.y, bcc); bcc = fmaf(w1.z, x0.z, bcc); bcc = fmaf(w1.w, x0.w, bcc);
                if ((mm & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // at most four slices of operands live beside the 192 weight registers
            }
            const float v = sum4(a0 + a1), u = sum4(b0 + b1);
            if (hp == 0) {
                lds[T::o_o1 + hr] = fmaxf(v + lds[T::o_bias + S + hr], 0.f);
                lds[T::o_o1 + hr + 64] = fmaxf(u + lds[T::o_bias + S + hr + 64], 0.f);
            }
        } else {
            older_b3<T, 0>(lds, wreg, q + 1, tg);
        }
        SWN_BAR(7)
        if constexpr (!grpA) {
            older_b2<T, 3>(lds, wreg, q + 1, tg);
        }
        if (tid < 64) {
            // out_2: NO <= 6 rows, 4 lanes per row, weights resident in LDS; then the Laplace head, evaluated uniformly by every
            // lane of wave 0 so that the new sample is in registers for the next input layer (cswnv_shift1.py:368-391)
            // (eight lanes per row, 16 inputs each, two chains: the serial 32-deep chain of the four-lane form was a third of this phase)
            const int r = tid >> 3, pp = tid & 7;
            float acc = 0.f, acc1 = 0.f;
            if (r < T::NO) {
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    const float4 w = *reinterpret_cast<const float4*>(lds + T::o_w2 + r * S + 16 * pp + 4 * mm);
                    const float4 x = *reinterpret_cast<const float4*>(lds + T::o_o1 + 16 * pp + 4 * mm);
                    float& ac = (mm & 1) ? acc1 : acc;
                    ac = fmaf(w.x, x.x, ac); ac = fmaf(w.y, x.y, ac);
                    ac = fmaf(w.z, x.z, ac); ac = fmaf(w.w, x.w, ac);
                }
            }
            acc = sum4(acc + acc1);
            acc += dpp_f<0x141>(acc);                          // row_half_mirror: the octet's sum
            if (r < T::NO) acc += lds[T::o_w2 + T::NO * S + r];
            if (HEADS_ON && a.heads && pp == 0 && r < T::NO) a.heads[((size_t)b * a.n_steps + i) * T::NO + r] = acc;
            // the NO head outputs sit in lanes 0, 8, 16, ...: broadcast them through scalar registers (an LDS write, a wave barrier and
            // a broadcast read stood here: ~150 cycles of the one wave the whole workgroup waits for)
            float o2[T::NO];
#pragma unroll
            for (int k = 0; k < T::NO; ++k) o2[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc), 8 * k));
            {
#pragma clang fp contract(off)
                float* outp = reinterpret_cast<float*>(a.out) + (size_t)b * a.n_steps + (size_t)i;
                const float mu = o2[0];
                const float bsc = sigm(o2[1]);                 // exp(logsigmoid(y))
                float lpv = 0.f;
#pragma unroll
                for (int k = 0; k < T::LPC; ++k) lpv += o2[2 + T::LPC - 1 - k] * win[T::WN - T::LPC + k];
                const float t = bsc * (EXT ? lds[T::o_tnz + ((i >> 6) & (T::NZB - 1)) * T::NZC + (i & (T::NZC - 1))]
                                           : lds[T::o_tnz + (i & 1) * 8]);
                float sv = (T::LPC > 0) ? (lpv + mu) - t : mu - t;
                sv = fminf(fmaxf(sv, -1.f), 1.f);
                if (tid == 0) outp[0] = sv;
                const float fd = a.forced ? reinterpret_cast<const float*>(a.forced)[(size_t)b * a.n_steps + i] : sv;
#pragma unroll
                for (int k = 0; k + 1 < T::WN; ++k) win[k] = win[k + 1];
                win[T::WN - 1] = fd;
            }
        }
        if (i + 1 < a.n_steps) input_gen(q + 1);
#ifdef SWN_STAMP
        { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tw[8] += tn - tlast; tstep += tn - tbeg; }
#endif
    };
    if constexpr (EXT) {
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
#undef SWN_PHASE
#undef SWN_BAR
#ifdef SWN_STAMP
    if ((tid == 0 || tid == NG) && b == 0 && a.heads) {
        float* h = a.heads + (tid == 0 ? 0 : 10);
        for (int k = 0; k < 9; ++k) h[k] = (float)((double)tw[k] / (double)a.n_steps);
        if (tid == 0) h[9] = (float)((double)tstep / (double)a.n_steps);
    }
#endif
}

template <class T>
__global__ __launch_bounds__(NT) void decode_bl6w_kernel(const W6Args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (threadIdx.x < NG) decode_body<T, true>(a, lds);       // wave-uniform
    else decode_body<T, false>(a, lds);
}

template <class T>
int launch_mode(const W6Args& a, hipStream_t st) {
    static_assert(T::lds_bytes <= 160 * 1024, "LDS budget");
    auto kern = decode_bl6w_kernel<T>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::lds_bytes) != hipSuccess)
        return SWN_E_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(NT), T::lds_bytes, st, a);
    return swn_launch_status("swn_decode(bl6w)");
}

template <class T>
int launch(const W6Args& a, hipStream_t st) {
    const bool ext = !a.nz.ptr || a.nz.dump || a.seed;
    return ext ? launch_mode<typename T::Ext>(a, st) : launch_mode<T>(a, st);
}

}  // namespace

// SWN_E_UNSUPPORTED: not a single-sample Laplace net of the BL6 class (the symmetric kernel, swn_decode_bl6.hip, takes the others)
extern "C" int swn_decode_bl6w_try(const swn_net_desc* d, const float* packed, const float* cond, int batch,
                                   int n_frames, int n_steps, const SwnNoise* nz, const void* forced,
                                   const void* seed, void* out, float* heads, void* stream_) {
    SwnGeom g; int rc = swn_make_geom(d, &g);
    if (rc < 0) return rc;
    if (!g.bl6 || g.U > 256 || g.U < 2 || g.audio_in || g.kind != SWN_KIND_LAPLACE || g.S != 128 || g.O1 != 128 || g.seg != 1 ||
        (g.lpc != 0 && g.lpc != 4))
        return SWN_E_UNSUPPORTED;
    W6Args a;
    swn_make_layout(&g, &a.y);
    a.P = packed; a.cond = cond; a.noise = nz->ptr; a.nz = *nz; a.forced = forced; a.seed = seed; a.out = out; a.heads = heads;
    a.B = batch; a.Tf = n_frames; a.n_steps = n_steps; a.U = g.U; a.N = g.N;
    hipStream_t st = (hipStream_t)stream_;
    (void)hipGetLastError();
    if (g.lpc == 0) return launch<Tw<0, false>>(a, st);
    return launch<Tw<4, false>>(a, st);
}
