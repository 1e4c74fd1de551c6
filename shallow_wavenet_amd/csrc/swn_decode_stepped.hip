// Stepped autoregressive decode for LARGE geometries (reference-shipped REF6: H=192/256, K=7, 15-24 MB
// of weights touched per generated sample) on gfx950.
//
// One CU cannot hold or stream that much per step, and in-kernel cross-CU hand-offs cost 1-3 us each
// (MI355X_MICROARCH.md price list) - about what a dependent kernel boundary costs (1.45 us).  So every
// phase of a step is its OWN launch, spread over many CUs that each read a slice of the weight rows from
// L2 / Infinity Cache:
//     step_layer   x L : rows of one dilated conv + fused gate        one wave per channel pair
//     rowvec       skip (all out_skip 1x1s as one mat-vec), out_1    one wave per row
//     step_tail    out_2 + sampling + history update + the NEXT step's input layer   1 workgroup / utterance
// L+3 launches per generated step, no spinning, no inter-workgroup protocol: the stream order is the
// dependency chain (cswnv_shift1.py:348-402).  The iteration index is a launch argument (no dependent
// load at kernel entry); measured: the chain is bound by the L2 round trips inside each launch, not by
// host launch cost (hipGraph replay of the same chain ran at the same speed, so it is not used).
// State (history rings, hcat, skip, out_1, sample window) lives in the caller's scratch buffer; the math
// and the ring layout are those of the generic persistent kernel (swn_decode.hip).
#include <hip/hip_runtime.h>
#include "swn_geom.hpp"
#include "swn_noise.hpp"

namespace {

struct StArgs {
    SwnGeom g;
    SwnLayout y;
    const float* P; const float* cond; SwnNoise nz; const void* forced; const void* seed;
    float* state; void* out; float* heads;
    int B, Tf, n_steps, n_pro, WN;
    int ring_off[SWN_MAXL], ring_len[SWN_MAXL];
    int o_hcat, o_skip, o_o1, o_o2, o_hist, o_cnt, stride;      // per-utterance float offsets
    int o2_by_rowvec;                                           // out_2 was computed by a rowvec launch into o_o2 (wide heads)
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ int pmod(int r, int m) { int t = r % m; return t < 0 ? t + m : t; }

// Every launch of this chain is a handful of memory round trips, so the loads of a phase must all be in flight
// together.  A conditional load (`ok ? *p : 0`) compiles to an exec-masked branch followed by s_waitcnt vmcnt(0):
// the first version of these kernels paid 12-14 SERIAL round trips per launch (4.5-8 us).  Loads therefore go
// through buffer resources with 32-bit byte offsets, and "not mine / past the end" is the out-of-range offset
// (reads zero, no branch).  The packed parameters and the whole decode state must each stay below 2 GiB.
constexpr unsigned ST_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t st_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 st_ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float st_ld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ float sum64(float v) {
    v += __shfl_xor(v, 32, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);  v += __shfl_xor(v, 1, 64);
    return v;
}
__device__ __forceinline__ float sum32(float v) {
    v += __shfl_xor(v, 16, 32); v += __shfl_xor(v, 8, 32); v += __shfl_xor(v, 4, 32);
    v += __shfl_xor(v, 2, 32);  v += __shfl_xor(v, 1, 32);
    return v;
}

// 64-lane sums of EIGHT values at once: a butterfly in which a lane keeps half of its values at each of the first three
// exchanges (xor 32, 16, 8) and sums the survivor over xor 4, 2, 1 - 10 exchanges instead of 8 x 6, and every value goes
// through exactly the pairings of sum64 in the same order (bit-identical).  Returns utterance (lane >> 3)'s total.
__device__ __forceinline__ float sum64x8(const float (&v)[8], int lane) {
    float a4[4], a2[2], a1;
    const bool h5 = lane & 32, h4 = lane & 16, h3 = lane & 8;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float keep = h5 ? v[k + 4] : v[k], give = h5 ? v[k] : v[k + 4];
        a4[k] = keep + __shfl_xor(give, 32, 64);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float keep = h4 ? a4[k + 2] : a4[k], give = h4 ? a4[k] : a4[k + 2];
        a2[k] = keep + __shfl_xor(give, 16, 64);
    }
    {
        const float keep = h3 ? a2[1] : a2[0], give = h3 ? a2[0] : a2[1];
        a1 = keep + __shfl_xor(give, 8, 64);
    }
    a1 += __shfl_xor(a1, 4, 64); a1 += __shfl_xor(a1, 2, 64); a1 += __shfl_xor(a1, 1, 64);
    return a1;
}

// iteration `it` of the per-utterance counter: it < n_pro is a prologue position, else generation step
struct Iter { bool gen; int i, np, q0; };
__device__ __forceinline__ Iter iter_of(const StArgs& a, int it) {
    Iter r; r.gen = it >= a.n_pro; r.i = it - a.n_pro; r.np = r.gen ? a.g.seg : 1;
    r.q0 = r.gen ? a.g.rf + 1 - a.g.seg + r.i * a.g.seg : it;
    return r;
}

// ---- input layer of iteration `it` -> ring 0 (device function: own launch in the prologue, fused into
//      the tail of the previous step during generation)
template <int KIND>
__device__ __forceinline__ void input_layer(const StArgs& a, float* st, const int it, const int tid, const int nthreads,
                                            const float* win) {
    // win[0..WN): the sample window (LDS copy; float samples or int class indices).  All parameter loads are
    // unconditional and selected afterwards, so the K taps cost one memory round trip, not K.
    const SwnGeom& g = a.g;
    const Iter r = iter_of(a, it);
    const float* P = a.P;
    const int H = g.H, K = g.K, seg = g.seg, WN = a.WN;
    for (int e = tid; e < H * r.np; e += nthreads) {
        const int j = e / H, o = e - j * H, q = r.q0 + j;
        float acc = P[a.y.cb + o];
        for (int k = 0; k < K; ++k) {
            const int rr = q - (K - 1 - k);
            if (KIND == SWN_KIND_LAPLACE) {
                const int qe = r.gen ? g.rf + r.i * seg : g.rf;
                int wi = rr - qe + WN - 1; wi = wi < 0 ? 0 : (wi >= WN ? WN - 1 : wi);
                const float sv = r.gen ? win[wi] : 0.f;
                const float t = fmaf(P[a.y.cv + (size_t)k * H + o], sv, P[a.y.cc + (size_t)k * H + o]);
                acc += (rr >= -(seg - 1)) ? t : 0.f;
            } else {
                const int qe = r.gen ? g.rf + r.i : g.rf;
                int wi = rr - qe + WN - 1; wi = wi < 0 ? 0 : (wi >= WN ? WN - 1 : wi);
                const int idx = r.gen ? __builtin_bit_cast(int, win[wi]) : g.Q / 2;
                const float t = P[a.y.ct + ((size_t)k * g.Q + idx) * H + o];
                acc += (rr >= 0) ? t : 0.f;
            }
        }
        st[a.ring_off[0] + pmod(q, a.ring_len[0]) * g.Hp + o] = acc / (1.f + fabsf(acc));
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void step_in_kernel(const StArgs a, const int it) {
    __shared__ float lwin[32];
    float* st = a.state + (size_t)blockIdx.x * a.stride;
    if ((int)threadIdx.x < a.WN) lwin[threadIdx.x] = st[a.o_hist + threadIdx.x];
    __syncthreads();
    input_layer<KIND>(a, st, it, threadIdx.x, 256, lwin);
}

// ---- step_layer: ONE wave per channel pair (gate row + candidate row), weights kept in registers.
//      Kernel boundaries invalidate the per-XCD L2s, so every launch re-fetches its weight rows from the
//      Infinity Cache at ~30 GB/s per CU: H workgroups of 64 lanes keep each CU's share at ~10 KB.
template <int NI, int KIND, int BT>   // NI = ceil(K*Hp / 256): float4 pieces per lane and row; BT = utterances per tile
__global__ __launch_bounds__(64) void step_layer_kernel(const StArgs a, const int l, const int it) {
    const SwnGeom& g = a.g;
    const int lane = threadIdx.x;
    const int o = blockIdx.x;
    const int H = g.H, Hp = g.Hp, K = g.K, H2 = 2 * g.H, seg = g.seg, KH = K * Hp;
    const bool live = o < H;
    const float* P = a.P;
    const __amdgpu_buffer_rsrc_t rP = st_rsrc(P), rS = st_rsrc(a.state);
    float4 wz[NI], wc[NI];
    {
        const size_t rz = a.y.wd + ((size_t)l * H2 + (live ? o : 0)) * KH, rc = rz + (size_t)H * KH;
#pragma unroll
        for (int pc = 0; pc < NI; ++pc) {
            const int idx = pc * 256 + lane * 4;
            const bool ok = live && idx < KH;
            wz[pc] = st_ld4(rP, ok ? (unsigned)((rz + idx) * 4) : ST_OOB);
            wc[pc] = st_ld4(rP, ok ? (unsigned)((rc + idx) * 4) : ST_OOB);
        }
    }
    const int dil = g.dil[l], R = a.ring_len[l];
    const Iter r = iter_of(a, it);
    const int b0 = blockIdx.y * BT;
    const int nb = a.B - b0 < BT ? a.B - b0 : BT;              // utterances of this tile, processed CONCURRENTLY:
    for (int j = 0; j < r.np; ++j) {                          // lane u finishes utterance b0+u
        const int q = r.q0 + j;
        float az[BT], ac[BT];
#pragma unroll
        for (int u = 0; u < BT; ++u) { az[u] = 0.f; ac[u] = 0.f; }
        float4 xv[NI][BT];                                     // all activation loads of the position in flight at once
#pragma unroll
        for (int pc = 0; pc < NI; ++pc) {
            const int idx = pc * 256 + lane * 4;
            const int ic = idx < KH ? idx : 0;
            const int tap = ic / Hp, i = ic - tap * Hp;
            const size_t xo = a.ring_off[l] + (size_t)pmod(q - (K - 1 - tap) * dil, R) * Hp + i;
#pragma unroll
            for (int u = 0; u < BT; ++u)
                xv[pc][u] = st_ld4(rS, (idx < KH && u < nb) ? (unsigned)(((size_t)(b0 + u) * a.stride + xo) * 4) : ST_OOB);
        }
        // epilogue operands: issued behind the activation loads (this block is an exec-masked branch that waits for
        // its own loads; placed first it would hold the activation loads back by a round trip)
        float gz = 0.f, gc = 0.f, bdz = 0.f, bdc = 0.f, hp = 0.f;
        if (lane < nb && live) {
            const int b = b0 + lane;
            const float* st = a.state + (size_t)b * a.stride;
            gz = P[a.y.bx + (size_t)l * H2 + o]; gc = P[a.y.bx + (size_t)l * H2 + H + o];
            bdz = P[a.y.bd + (size_t)l * H2 + o]; bdc = P[a.y.bd + (size_t)l * H2 + H + o];
            hp = st[a.ring_off[l] + (size_t)pmod(q, R) * Hp + o];
            const float* condb = a.cond + (size_t)b * a.Tf * g.N;
            for (int s = 0; s < seg; ++s) {
                int tt = q + s - g.rf; tt = tt < 0 ? 0 : tt;
                int f = tt / g.U; const int jj = tt - f * g.U;
                f = f < a.Tf ? f : a.Tf - 1;
                const float w = P[a.y.wup + jj];
                const float* cr = condb + (size_t)f * g.N + (size_t)(l * seg + s) * H2;
                gz = fmaf(w, cr[o], gz); gc = fmaf(w, cr[H + o], gc);
            }
            if (KIND == SWN_KIND_SOFTMAX && g.audio_in) {
                const int* ihist = reinterpret_cast<const int*>(st + a.o_hist);
                const int qe = r.gen ? g.rf + r.i : g.rf;
                const int idx = r.gen ? ihist[q - qe + a.WN - 1] : g.Q / 2;
                const float* wa = P + a.y.wxa + ((size_t)l * g.Q + idx) * H2;
                gz += wa[o]; gc += wa[H + o];
            }
        }
#pragma unroll
        for (int pc = 0; pc < NI; ++pc)
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                const float4 x = xv[pc][u];
                az[u] = fmaf(wz[pc].x, x.x, az[u]); az[u] = fmaf(wz[pc].y, x.y, az[u]);
                az[u] = fmaf(wz[pc].z, x.z, az[u]); az[u] = fmaf(wz[pc].w, x.w, az[u]);
                ac[u] = fmaf(wc[pc].x, x.x, ac[u]); ac[u] = fmaf(wc[pc].y, x.y, ac[u]);
                ac[u] = fmaf(wc[pc].z, x.z, ac[u]); ac[u] = fmaf(wc[pc].w, x.w, ac[u]);
            }
        float myz = 0.f, myc = 0.f;
#pragma unroll
        for (int u = 0; u < BT; ++u) {
            if (u < nb) {
                const float sz = sum64(az[u]), sc = sum64(ac[u]);
                if (lane == u) { myz = sz; myc = sc; }
            }
        }
        if (lane < nb && live) {
            float* st = a.state + (size_t)(b0 + lane) * a.stride;
            const float z = sigm(gz * (myz + bdz));
            const float c = tanhf(gc * (myc + bdc));
            const float hn = (1.f - z) * c + z * hp;
            if (l + 1 < g.L) st[a.ring_off[l + 1] + pmod(q, a.ring_len[l + 1]) * Hp + o] = hn;
            if (j == r.np - 1) st[a.o_hcat + l * Hp + o] = hn;
        }
    }
}

// ---- step_layer for MANY utterances: a 512-thread workgroup = 8 channel pairs x 8 utterances.  Per-utterance workgroups
//      re-read a pair's weight rows once per utterance and every pair's wave re-reads the utterance's K H activations:
//      98 MB through L2 per layer launch at 64 utterances, 9.5 us - the launch is bound by that traffic, not by latency.
//      Here wave w keeps the 2 K H weights of pair 8 bx + w in registers (fetched once per 8 utterances), wave u stages the
//      activations of utterance 8 by + u in LDS (fetched once per 8 pairs: 25 MB per launch), and after ONE barrier every
//      wave forms its pair's two sums for the eight utterances out of LDS - the same lane-by-lane sums as the kernels above
//      (bit-identical results).  Lane 8 u of a wave finishes utterance u.
constexpr int ST_TU = 8;                                       // utterances (= waves) of a tile
template <int NI, int KIND>
__global__ __launch_bounds__(64 * ST_TU) void step_layer_tile_kernel(const StArgs a, const int l, const int it) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [ST_TU][NI * 256]
    const SwnGeom& g = a.g;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int o = blockIdx.x * ST_TU + w;
    const int H = g.H, Hp = g.Hp, K = g.K, H2 = 2 * g.H, seg = g.seg, KH = K * Hp;
    const bool live = o < H;
    const float* P = a.P;
    const __amdgpu_buffer_rsrc_t rP = st_rsrc(P), rS = st_rsrc(a.state);
    float4 wz[NI], wc[NI];
    {
        const size_t rz = a.y.wd + ((size_t)l * H2 + (live ? o : 0)) * KH, rc = rz + (size_t)H * KH;
#pragma unroll
        for (int pc = 0; pc < NI; ++pc) {
            const int idx = pc * 256 + lane * 4;
            const bool ok = live && idx < KH;
            wz[pc] = st_ld4(rP, ok ? (unsigned)((rz + idx) * 4) : ST_OOB);
            wc[pc] = st_ld4(rP, ok ? (unsigned)((rc + idx) * 4) : ST_OOB);
        }
    }
    const int dil = g.dil[l], R = a.ring_len[l];
    const Iter r = iter_of(a, it);
    const int b0 = blockIdx.y * ST_TU;
    const int nb = a.B - b0 < ST_TU ? a.B - b0 : ST_TU;
    for (int j = 0; j < r.np; ++j) {
        const int q = r.q0 + j;
        // wave w stages utterance b0 + w
        {
            float4 xv[NI];
#pragma unroll
            for (int pc = 0; pc < NI; ++pc) {
                const int idx = pc * 256 + lane * 4;
                const int ic = idx < KH ? idx : 0;
                const int tap = ic / Hp, i = ic - tap * Hp;
                const size_t xo = a.ring_off[l] + (size_t)pmod(q - (K - 1 - tap) * dil, R) * Hp + i;
                xv[pc] = st_ld4(rS, (idx < KH && w < nb) ? (unsigned)(((size_t)(b0 + w) * a.stride + xo) * 4) : ST_OOB);
            }
            if (j > 0) __syncthreads();                        // the previous position's sums are done with the buffer
#pragma unroll
            for (int pc = 0; pc < NI; ++pc) *reinterpret_cast<float4*>(xs + (w * NI + pc) * 256 + lane * 4) = xv[pc];
        }
        // epilogue operands: issued behind the activation loads (an exec-masked block that waits for its own loads)
        float gz = 0.f, gc = 0.f, bdz = 0.f, bdc = 0.f, hp = 0.f;
        const int ut = lane >> 3;                              // the utterance this lane's octet ends up with (sum64x8)
        const bool fin = (lane & 7) == 0 && ut < nb && live;
        if (fin) {
            const int b = b0 + ut;
            const float* st = a.state + (size_t)b * a.stride;
            gz = P[a.y.bx + (size_t)l * H2 + o]; gc = P[a.y.bx + (size_t)l * H2 + H + o];
            bdz = P[a.y.bd + (size_t)l * H2 + o]; bdc = P[a.y.bd + (size_t)l * H2 + H + o];
            hp = st[a.ring_off[l] + (size_t)pmod(q, R) * Hp + o];
            const float* condb = a.cond + (size_t)b * a.Tf * g.N;
            for (int s = 0; s < seg; ++s) {
                int tt = q + s - g.rf; tt = tt < 0 ? 0 : tt;
                int f = tt / g.U; const int jj = tt - f * g.U;
                f = f < a.Tf ? f : a.Tf - 1;
                const float wv = P[a.y.wup + jj];
                const float* cr = condb + (size_t)f * g.N + (size_t)(l * seg + s) * H2;
                gz = fmaf(wv, cr[o], gz); gc = fmaf(wv, cr[H + o], gc);
            }
            if (KIND == SWN_KIND_SOFTMAX && g.audio_in) {
                const int* ihist = reinterpret_cast<const int*>(st + a.o_hist);
                const int qe = r.gen ? g.rf + r.i : g.rf;
                const int idx = r.gen ? ihist[q - qe + a.WN - 1] : g.Q / 2;
                const float* wa = P + a.y.wxa + ((size_t)l * g.Q + idx) * H2;
                gz += wa[o]; gc += wa[H + o];
            }
        }
        __syncthreads();
        float azv[ST_TU], acv[ST_TU];
#pragma unroll
        for (int u = 0; u < ST_TU; ++u) {
            float az = 0.f, ac = 0.f;
#pragma unroll
            for (int pc = 0; pc < NI; ++pc) {
                const float4 x = *reinterpret_cast<const float4*>(xs + (u * NI + pc) * 256 + lane * 4);
                az = fmaf(wz[pc].x, x.x, az); az = fmaf(wz[pc].y, x.y, az);
                az = fmaf(wz[pc].z, x.z, az); az = fmaf(wz[pc].w, x.w, az);
                ac = fmaf(wc[pc].x, x.x, ac); ac = fmaf(wc[pc].y, x.y, ac);
                ac = fmaf(wc[pc].z, x.z, ac); ac = fmaf(wc[pc].w, x.w, ac);
            }
            azv[u] = az; acv[u] = ac;
        }
        const float myz = sum64x8(azv, lane), myc = sum64x8(acv, lane);
        if (fin) {
            float* st = a.state + (size_t)(b0 + ut) * a.stride;
            const float z = sigm(gz * (myz + bdz));
            const float c = tanhf(gc * (myc + bdc));
            const float hn = (1.f - z) * c + z * hp;
            if (l + 1 < g.L) st[a.ring_off[l + 1] + pmod(q, a.ring_len[l + 1]) * Hp + o] = hn;
            if (j == r.np - 1) st[a.o_hcat + l * Hp + o] = hn;
        }
    }
}

// ---- rowvec: y[b][row] = act(bias[row] + W[row][:] . x[b][:]), ONE wave per row -------------------------
template <int BT>
__global__ __launch_bounds__(64) void rowvec_kernel(const StArgs a, size_t w_off, int ldw, size_t b_off, int rows,
                                                    int ni, int x_off, int y_off, int relu) {
    const int lane = threadIdx.x, row = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rP = st_rsrc(a.P), rS = st_rsrc(a.state);
    const size_t wr = w_off + (size_t)row * ldw;
    const float bias = a.P[b_off + row];
    const int b0 = blockIdx.y * BT;
    const int nb = a.B - b0 < BT ? a.B - b0 : BT;              // utterances of this tile, processed concurrently
    float acc[BT];
#pragma unroll
    for (int u = 0; u < BT; ++u) acc[u] = 0.f;
    constexpr int RV = 5;                                      // 1280 inputs per pass: every row of these nets in one pass
    for (int i0 = 0; i0 < ni; i0 += 256 * RV) {
        float4 wv[RV], xv[RV][BT];
#pragma unroll
        for (int pc = 0; pc < RV; ++pc) {
            const int idx = i0 + pc * 256 + lane * 4;
            const bool ok = idx < ni;
            wv[pc] = st_ld4(rP, ok ? (unsigned)((wr + idx) * 4) : ST_OOB);
#pragma unroll
            for (int u = 0; u < BT; ++u)
                xv[pc][u] = st_ld4(rS, (ok && u < nb) ? (unsigned)(((size_t)(b0 + u) * a.stride + x_off + idx) * 4) : ST_OOB);
        }
#pragma unroll
        for (int pc = 0; pc < RV; ++pc)
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                acc[u] = fmaf(wv[pc].x, xv[pc][u].x, acc[u]); acc[u] = fmaf(wv[pc].y, xv[pc][u].y, acc[u]);
                acc[u] = fmaf(wv[pc].z, xv[pc][u].z, acc[u]); acc[u] = fmaf(wv[pc].w, xv[pc][u].w, acc[u]);
            }
    }
    float mine = 0.f;
#pragma unroll
    for (int u = 0; u < BT; ++u) {
        if (u < nb) { const float sv = sum64(acc[u]); if (lane == u) mine = sv; }
    }
    if (lane < nb) {
        const float v = mine + bias;
        a.state[(size_t)(b0 + lane) * a.stride + y_off + row] = relu ? fmaxf(v, 0.f) : v;
    }
}

// the same for the 1x1 layers: 8 rows x 8 utterances per workgroup, the utterances' input vectors staged in LDS
template <int RV>                                              // host: ni <= 256 RV
__global__ __launch_bounds__(64 * ST_TU) void rowvec_tile_kernel(const StArgs a, size_t w_off, int ldw, size_t b_off, int rows,
                                                                 int ni, int x_off, int y_off, int relu) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [ST_TU][RV * 256]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = blockIdx.x * ST_TU + w;
    const bool live = row < rows;
    const __amdgpu_buffer_rsrc_t rP = st_rsrc(a.P), rS = st_rsrc(a.state);
    const size_t wr = w_off + (size_t)(live ? row : 0) * ldw;
    const float bias = live ? a.P[b_off + row] : 0.f;
    const int b0 = blockIdx.y * ST_TU;
    const int nb = a.B - b0 < ST_TU ? a.B - b0 : ST_TU;
    float4 wv[RV];
    {
        float4 xv[RV];
#pragma unroll
        for (int pc = 0; pc < RV; ++pc) {
            const int idx = pc * 256 + lane * 4;
            const bool ok = idx < ni;
            wv[pc] = st_ld4(rP, (ok && live) ? (unsigned)((wr + idx) * 4) : ST_OOB);
            xv[pc] = st_ld4(rS, (ok && w < nb) ? (unsigned)(((size_t)(b0 + w) * a.stride + x_off + idx) * 4) : ST_OOB);
        }
#pragma unroll
        for (int pc = 0; pc < RV; ++pc) *reinterpret_cast<float4*>(xs + (w * RV + pc) * 256 + lane * 4) = xv[pc];
    }
    __syncthreads();
    float accv[ST_TU];
#pragma unroll
    for (int u = 0; u < ST_TU; ++u) {
        float acc = 0.f;
#pragma unroll
        for (int pc = 0; pc < RV; ++pc) {
            const float4 x = *reinterpret_cast<const float4*>(xs + (u * RV + pc) * 256 + lane * 4);
            acc = fmaf(wv[pc].x, x.x, acc); acc = fmaf(wv[pc].y, x.y, acc);
            acc = fmaf(wv[pc].z, x.z, acc); acc = fmaf(wv[pc].w, x.w, acc);
        }
        accv[u] = acc;
    }
    const float mine = sum64x8(accv, lane);
    const int ut = lane >> 3;
    if ((lane & 7) == 0 && ut < nb && live) {
        const float v = mine + bias;
        a.state[(size_t)(b0 + ut) * a.stride + y_off + row] = relu ? fmaxf(v, 0.f) : v;
    }
}

// ---- step_tail: out_2, sampling, history update, then the input layer of the next step ---------------
template <int KIND>
__global__ __launch_bounds__(256) void step_tail_kernel(const StArgs a, const int it) {
    __shared__ float o2v[4096 + 16];
    __shared__ float lwin[32];                 // the updated sample window, for the fused next input layer
    const SwnGeom& g = a.g;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 31, grp = tid >> 5;
    float* st = a.state + (size_t)b * a.stride;
    const int i = it - a.n_pro, seg = g.seg, WN = a.WN;
    // out_2.  Wide heads (softmax: Q rows) arrive from a rowvec launch - one wave per row over the chip, like skip and
    // out_1 - in the state block; narrow heads (Laplace: <= 2 seg + lpc rows) are computed here:
    // 8 rows per pass, 32 lanes per row; branch-free loads (a row past NO / an input past O1p reads zeros)
    if (a.o2_by_rowvec) {
        for (int e = tid; e < g.NO; e += 256) o2v[e] = st[a.o_o2 + e];
    } else {
        const __amdgpu_buffer_rsrc_t rP = st_rsrc(a.P), rS = st_rsrc(a.state);
        const size_t xb = (size_t)b * a.stride + a.o_o1;
        for (int r0 = 0; r0 < g.NO; r0 += 8) {
            const int row = r0 + grp;
            const bool rok = row < g.NO;
            float acc = 0.f;
            for (int i0 = 0; i0 < g.O1p; i0 += 512) {
                float4 wv[4], xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = i0 + u * 128 + lane * 4;
                    const bool ok = rok && idx < g.O1p;
                    wv[u] = st_ld4(rP, ok ? (unsigned)((a.y.w2 + (size_t)row * g.O1p + idx) * 4) : ST_OOB);
                    xv[u] = st_ld4(rS, ok ? (unsigned)((xb + idx) * 4) : ST_OOB);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc = fmaf(wv[u].x, xv[u].x, acc); acc = fmaf(wv[u].y, xv[u].y, acc);
                    acc = fmaf(wv[u].z, xv[u].z, acc); acc = fmaf(wv[u].w, xv[u].w, acc);
                }
            }
            acc = sum32(acc);
            if (lane == 0 && rok) o2v[row] = acc + a.P[a.y.b2 + row];
        }
    }
    __syncthreads();
    float* shist = st + a.o_hist;
    int* ihist = reinterpret_cast<int*>(shist);
    if (a.heads) for (int e = tid; e < g.NO; e += 256) a.heads[((size_t)b * a.n_steps + i) * g.NO + e] = o2v[e];
    if (KIND == SWN_KIND_LAPLACE) {
        if (tid == 0) {
#pragma clang fp contract(off)
            // Laplace head, cswnv_shift1.py:368-391
            const float* forced = reinterpret_cast<const float*>(a.forced);
            float* outp = reinterpret_cast<float*>(a.out) + (size_t)b * a.n_steps * seg + (size_t)i * seg;
            float lp[16], fed[16];
            const int lpc = g.lpc;
            for (int k = 0; k < lpc; ++k) lp[k] = shist[WN - lpc + k];
            for (int j = 0; j < seg; ++j) {
                const float mu = o2v[j], yv = o2v[seg + j];
                const float bsc = expf(fminf(yv, 0.f) - log1pf(expf(-fabsf(yv))));
                float lpv = 0.f;
                for (int k = 0; k < lpc; ++k) lpv += o2v[2 * seg + lpc - 1 - k] * lp[k];
                const float e = swn_noise_laplace(a.nz, b, i, j, a.n_steps, seg);
                const float sg = (e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f);
                const float t = (bsc * sg) * log1pf(-2.f * fabsf(e));
                float sv = (lpc > 0) ? (lpv + mu) - t : mu - t;
                sv = fminf(fmaxf(sv, -1.f), 1.f);
                outp[j] = sv;
                const float fd = forced ? forced[(size_t)b * a.n_steps * seg + (size_t)i * seg + j] : sv;
                fed[j] = fd;
                for (int k = 0; k + 1 < lpc; ++k) lp[k] = lp[k + 1];
                if (lpc > 0) lp[lpc - 1] = fd;
            }
            for (int k = 0; k + seg < WN; ++k) { const float v = shist[k + seg]; shist[k] = v; lwin[k] = v; }
            for (int j = 0; j < seg; ++j) { shist[WN - seg + j] = fed[j]; lwin[WN - seg + j] = fed[j]; }
        }
    } else if (tid < 64) {
        // softmax head, dswnv.py:361-369
        const int Q = g.Q;
        float m = -INFINITY;
        for (int e = tid; e < Q; e += 64) m = fmaxf(m, o2v[e]);
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
        float sum = 0.f;
        for (int e = tid; e < Q; e += 64) sum += expf(o2v[e] - m);
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
        float sum2 = 0.f;
        for (int e = tid; e < Q; e += 64) sum2 += expf(o2v[e] - m) / sum;
        for (int d = 32; d >= 1; d >>= 1) sum2 += __shfl_xor(sum2, d, 64);
        float best = -1.f; int bi = 0x7fffffff;
        if ((Q & 3) == 0) {          // four classes per generator call / 16-byte noise load
            for (int g4 = tid; 4 * g4 < Q; g4 += 64) {
                const float4 q4 = swn_noise_exp1x4(a.nz, b, i, g4, a.n_steps, Q);
                const float qv[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = 4 * g4 + k;
                    const float r = ((expf(o2v[e] - m) / sum) / sum2) / qv[k];
                    if (r > best) { best = r; bi = e; }
                }
            }
        } else {
            for (int e = tid; e < Q; e += 64) {
                const float r = ((expf(o2v[e] - m) / sum) / sum2) / swn_noise_exp1(a.nz, b, i, e, a.n_steps, Q);
                if (r > best) { best = r; bi = e; }
            }
        }
        for (int d = 32; d >= 1; d >>= 1) {
            const float ob = __shfl_xor(best, d, 64);
            const int oi = __shfl_xor(bi, d, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (tid == 0) {
            const int* forced = reinterpret_cast<const int*>(a.forced);
            reinterpret_cast<int*>(a.out)[(size_t)b * a.n_steps + i] = bi;
            const int fd = forced ? forced[(size_t)b * a.n_steps + i] : bi;
            for (int k = 0; k + 1 < WN; ++k) { const int v = ihist[k + 1]; ihist[k] = v; lwin[k] = __builtin_bit_cast(float, v); }
            ihist[WN - 1] = fd; lwin[WN - 1] = __builtin_bit_cast(float, fd);
        }
    }
    // the sample window was just updated through global memory by thread 0: make it visible to the
    // block (same CU), then run the next step's input layer here - one launch less per step
    __syncthreads();
    if (i + 1 < a.n_steps) input_layer<KIND>(a, st, it + 1, tid, 256, lwin);
}

// ---- step_tail of the Laplace nets with every global operand requested up front.  The kernel above pays its dependent
//      round trips one after the other (out_2 operands, the out_2 bias, the LP window, the noise draw, the window shift, the
//      input layer's taps: 7.4-8.1 us per launch, the longest of the chain); nothing of that depends on out_2 except the
//      arithmetic, so here the sample window, the step's noise draws, the out_2 bias and the K + 1 parameter rows of the
//      next input layer are all in flight with the out_2 operands, and what follows the first barrier works on LDS and
//      registers.  Same formulas in the same order as step_tail_kernel<LAPLACE> + input_layer (bit-identical results).
template <int MAXE>      // elements (channel, position) of the next input layer per thread: ceil(H * seg / 256) <= MAXE
__global__ __launch_bounds__(256) void step_tail_laplace_kernel(const StArgs a, const int it) {
    __shared__ float o2v[64];                  // NO <= 48
    __shared__ float lwin[32];                 // the updated sample window, for the fused next input layer
    __shared__ float lold[32];                 // the window as the step found it
    __shared__ float lnz[16];                  // the step's noise draws
    const SwnGeom& g = a.g;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 31, grp = tid >> 5;
    float* st = a.state + (size_t)b * a.stride;
    const int i = it - a.n_pro, seg = g.seg, WN = a.WN, H = g.H, K = g.K;
    const float* P = a.P;
    float* shist = st + a.o_hist;
    // ---- requests: out_2 rows and bias, window, noise, input-layer parameters
    const __amdgpu_buffer_rsrc_t rP = st_rsrc(P), rS = st_rsrc(a.state);
    const size_t xb = (size_t)b * a.stride + a.o_o1;
    const int nin = g.O1p;                     // <= 512 (host)
    float4 wv[4], wv2[4], xv[4];
    const bool rok = grp < g.NO, rok2 = grp + 8 < g.NO;          // NO <= 16 (host): rows grp and grp + 8 of the 32-lane group
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = u * 128 + lane * 4;
        wv[u] = st_ld4(rP, (rok && idx < nin) ? (unsigned)((a.y.w2 + (size_t)grp * g.O1p + idx) * 4) : ST_OOB);
        wv2[u] = st_ld4(rP, (rok2 && idx < nin) ? (unsigned)((a.y.w2 + (size_t)(grp + 8) * g.O1p + idx) * 4) : ST_OOB);
        xv[u] = st_ld4(rS, (rok && idx < nin) ? (unsigned)((xb + idx) * 4) : ST_OOB);
    }
    const float b2v = st_ld1(rP, (rok && lane == 0) ? (unsigned)((a.y.b2 + grp) * 4) : ST_OOB);
    const float b2v2 = st_ld1(rP, (rok2 && lane == 0) ? (unsigned)((a.y.b2 + grp + 8) * 4) : ST_OOB);
    const float hv = st_ld1(rS, tid < WN ? (unsigned)((((size_t)b * a.stride) + a.o_hist + tid) * 4) : ST_OOB);
    float ev = 0.f;
    if (tid >= 64 && tid < 64 + seg) ev = swn_noise_laplace(a.nz, b, i, tid - 64, a.n_steps, seg);
    // next input layer (iteration it + 1, a generation step): element e = tid + 256 m -> position j = e / H, channel o
    const bool more = i + 1 < a.n_steps;
    float pcb[MAXE], pcv[MAXE][8], pcc[MAXE][8];
#pragma unroll
    for (int m = 0; m < MAXE; ++m) {
        const int e = tid + 256 * m;
        const bool ok = more && e < H * seg;
        const int o = ok ? e % H : 0;
        pcb[m] = st_ld1(rP, ok ? (unsigned)((a.y.cb + o) * 4) : ST_OOB);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            pcv[m][k] = st_ld1(rP, (ok && k < K) ? (unsigned)((a.y.cv + (size_t)k * H + o) * 4) : ST_OOB);
            pcc[m][k] = st_ld1(rP, (ok && k < K) ? (unsigned)((a.y.cc + (size_t)k * H + o) * 4) : ST_OOB);
        }
    }
    // ---- out_2 (NO <= 16 rows, O1p <= 512 inputs: one pass over the inputs per row)
    {
        float acc = 0.f, acc2 = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = fmaf(wv[u].x, xv[u].x, acc); acc = fmaf(wv[u].y, xv[u].y, acc);
            acc = fmaf(wv[u].z, xv[u].z, acc); acc = fmaf(wv[u].w, xv[u].w, acc);
            acc2 = fmaf(wv2[u].x, xv[u].x, acc2); acc2 = fmaf(wv2[u].y, xv[u].y, acc2);
            acc2 = fmaf(wv2[u].z, xv[u].z, acc2); acc2 = fmaf(wv2[u].w, xv[u].w, acc2);
        }
        acc = sum32(acc); acc2 = sum32(acc2);
        if (lane == 0 && rok) o2v[grp] = acc + b2v;
        if (lane == 0 && rok2) o2v[grp + 8] = acc2 + b2v2;
    }
    if (tid < WN) lold[tid] = hv;
    if (tid >= 64 && tid < 64 + seg) lnz[tid - 64] = ev;
    __syncthreads();
    if (a.heads) for (int e = tid; e < g.NO; e += 256) a.heads[((size_t)b * a.n_steps + i) * g.NO + e] = o2v[e];
    if (tid == 0) {
#pragma clang fp contract(off)
        // Laplace head, cswnv_shift1.py:368-391
        const float* forced = reinterpret_cast<const float*>(a.forced);
        float* outp = reinterpret_cast<float*>(a.out) + (size_t)b * a.n_steps * seg + (size_t)i * seg;
        float lp[16], fed[16];
        const int lpc = g.lpc;
        for (int k = 0; k < lpc; ++k) lp[k] = lold[WN - lpc + k];
        for (int j = 0; j < seg; ++j) {
            const float mu = o2v[j], yv = o2v[seg + j];
            const float bsc = expf(fminf(yv, 0.f) - log1pf(expf(-fabsf(yv))));
            float lpv = 0.f;
            for (int k = 0; k < lpc; ++k) lpv += o2v[2 * seg + lpc - 1 - k] * lp[k];
            const float e = lnz[j];
            const float sg = (e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f);
            const float t = (bsc * sg) * log1pf(-2.f * fabsf(e));
            float sv = (lpc > 0) ? (lpv + mu) - t : mu - t;
            sv = fminf(fmaxf(sv, -1.f), 1.f);
            outp[j] = sv;
            const float fd = forced ? forced[(size_t)b * a.n_steps * seg + (size_t)i * seg + j] : sv;
            fed[j] = fd;
            for (int k = 0; k + 1 < lpc; ++k) lp[k] = lp[k + 1];
            if (lpc > 0) lp[lpc - 1] = fd;
        }
        for (int k = 0; k + seg < WN; ++k) { const float v = lold[k + seg]; shist[k] = v; lwin[k] = v; }
        for (int j = 0; j < seg; ++j) { shist[WN - seg + j] = fed[j]; lwin[WN - seg + j] = fed[j]; }
    }
    __syncthreads();
    if (!more) return;
    // ---- input layer of iteration it + 1 out of the prefetched rows (input_layer<LAPLACE>, generation form)
    const int i1 = i + 1;
    const int q0 = g.rf + 1 - seg + i1 * seg, qe = g.rf + i1 * seg;
#pragma unroll
    for (int m = 0; m < MAXE; ++m) {
        const int e = tid + 256 * m;
        if (e < H * seg) {
            const int j = e / H, o = e - j * H, q = q0 + j;
            float acc = pcb[m];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < K) {
                    const int rr = q - (K - 1 - k);
                    int wi = rr - qe + WN - 1; wi = wi < 0 ? 0 : (wi >= WN ? WN - 1 : wi);
                    const float t = fmaf(pcv[m][k], lwin[wi], pcc[m][k]);
                    acc += (rr >= -(seg - 1)) ? t : 0.f;
                }
            }
            st[a.ring_off[0] + pmod(q, a.ring_len[0]) * g.Hp + o] = acc / (1.f + fabsf(acc));
        }
    }
}

// set the sample window after the state was zeroed: softmax = padding class Q/2 (dswnv.py:308) with the caller's seed
// class in the newest slot; Laplace = the caller's seed samples in the newest seg slots (cswnv_shift1.py:300-334)
__global__ void step_seed_kernel(const StArgs a) {
    const int b = blockIdx.x, k = threadIdx.x;
    float* hist = a.state + (size_t)b * a.stride + a.o_hist;
    if (k >= a.WN) return;
    if (a.g.kind == SWN_KIND_SOFTMAX) {
        const int sc = a.seed ? reinterpret_cast<const int*>(a.seed)[b] : a.g.Q / 2;
        reinterpret_cast<int*>(hist)[k] = (k == a.WN - 1) ? sc : a.g.Q / 2;
    } else if (a.seed && k >= a.WN - a.g.seg) {
        hist[k] = reinterpret_cast<const float*>(a.seed)[(size_t)b * a.g.seg + (k - (a.WN - a.g.seg))];
    }
}

int plan(StArgs& a) {
    const SwnGeom& g = a.g;
    int o = 0;
    for (int l = 0; l < g.L; ++l) { a.ring_off[l] = o; a.ring_len[l] = g.pad[l] + g.seg; o += a.ring_len[l] * g.Hp; }
    a.WN = (g.K - 1 > g.lpc ? g.K - 1 : g.lpc) + g.seg;
    a.o_hcat = o; o += g.L * g.Hp;
    a.o_skip = o; o += g.Sp;
    a.o_o1 = o; o += g.O1p;
    a.o_o2 = o; o += swn_round4(g.NO);
    a.o_hist = o; o += swn_round4(a.WN);
    a.o_cnt = o; o += 4;
    a.stride = (o + 63) & ~63;
    return a.stride;
}

}  // namespace

extern "C" size_t swn_decode_stepped_state_floats(const swn_net_desc* d, int batch) {
    StArgs a;
    if (swn_make_geom(d, &a.g) < 0 || batch < 1) return 0;
    return (size_t)plan(a) * batch;
}

extern "C" int swn_decode_stepped(const swn_net_desc* d, const float* packed, const float* cond, int batch, int n_frames,
                                  int n_steps, const SwnNoise* nz, const void* forced, const void* seed, float* state,
                                  void* out, float* heads, void* stream_) {
    StArgs a;
    int rc = swn_make_geom(d, &a.g);
    if (rc < 0) return rc;
    const SwnGeom& g = a.g;
    const int ni = (g.K * g.Hp + 255) / 256;
    if (ni > 8 || g.seg > 16 || g.lpc > 16 || g.NO > 4096) return SWN_E_UNSUPPORTED;
    { StArgs t; t.g = a.g; if ((size_t)plan(t) * batch * sizeof(float) >= (1ull << 31) || t.WN > 32) return SWN_E_UNSUPPORTED; }   // 32-bit buffer offsets; LDS window
    swn_make_layout(&a.g, &a.y);
    plan(a);
    a.P = packed; a.cond = cond; a.nz = *nz; a.forced = forced; a.seed = seed; a.state = state; a.out = out; a.heads = heads;
    a.B = batch; a.Tf = n_frames; a.n_steps = n_steps; a.n_pro = g.rf - g.seg + 1;
    // one 256-thread workgroup evaluates 8 rows per pass: beyond 64 rows (8 passes, ~7 us) a launch of its own is cheaper
    a.o2_by_rowvec = g.NO > 64 ? 1 : 0;
    hipStream_t st = (hipStream_t)stream_;
    if (hipMemsetAsync(state, 0, sizeof(float) * (size_t)a.stride * batch, st) != hipSuccess) return SWN_E_LAUNCH;
    if (g.kind == SWN_KIND_SOFTMAX || seed) hipLaunchKernelGGL(step_seed_kernel, dim3(batch), dim3(64), 0, st, a);
    // up to 64 utterances: one utterance per workgroup (weights re-read per utterance from the Infinity Cache;
    // measured faster than sharing: B=8 63 vs 137 us/step, B=64 162 vs 182 us/step on REF6);
    // otherwise tiles of 8 utterances share one weight fetch and are processed concurrently
    const bool solo = batch <= 64;
    const unsigned by = solo ? (unsigned)batch : (unsigned)((batch + 7) / 8);
    // many utterances: tiles of 8 channel pairs (rows) x 8 utterances (step_layer_tile / rowvec_tile)
    const bool seq = batch >= 24;                              // measured crossover at REF6: 16 utterances 47 (solo) / 53 us per step, 24: 59 / 56
    const unsigned sy = (unsigned)((batch + ST_TU - 1) / ST_TU);
    if (seq) {
        static bool attr_set = false;                          // dynamic LDS above 64 KB (NI = 8: 64 KB, RV = 9: 72 KB)
        if (!attr_set) {
            const int big = 72 * 1024;
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(step_layer_tile_kernel<8, SWN_KIND_LAPLACE>), hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(step_layer_tile_kernel<8, SWN_KIND_SOFTMAX>), hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(rowvec_tile_kernel<9>), hipFuncAttributeMaxDynamicSharedMemorySize, big) != hipSuccess)
                return SWN_E_LAUNCH;
            attr_set = true;
        }
    }
#define SWN_LAYER(NI_, KIND_)                                                                                  \
    do {                                                                                                        \
        if (seq) hipLaunchKernelGGL((step_layer_tile_kernel<NI_, KIND_>), dim3((g.H + ST_TU - 1) / ST_TU, sy), dim3(64 * ST_TU), \
                                    (size_t)ST_TU * NI_ * 256 * sizeof(float), st, a, l, it);                  \
        else if (solo) hipLaunchKernelGGL((step_layer_kernel<NI_, KIND_, 1>), grid, dim3(64), 0, st, a, l, it);      \
        else hipLaunchKernelGGL((step_layer_kernel<NI_, KIND_, 8>), grid, dim3(64), 0, st, a, l, it);           \
    } while (0)
    auto layers = [&](int it) {
        for (int l = 0; l < g.L; ++l) {
            dim3 grid(g.H, by);
            if (g.kind == SWN_KIND_LAPLACE) {
                if (ni <= 1) SWN_LAYER(1, SWN_KIND_LAPLACE);
                else if (ni <= 6) SWN_LAYER(6, SWN_KIND_LAPLACE);
                else SWN_LAYER(8, SWN_KIND_LAPLACE);
            } else {
                if (ni <= 1) SWN_LAYER(1, SWN_KIND_SOFTMAX);
                else if (ni <= 6) SWN_LAYER(6, SWN_KIND_SOFTMAX);
                else SWN_LAYER(8, SWN_KIND_SOFTMAX);
            }
        }
    };
#undef SWN_LAYER
    auto rowvec = [&](int rows, size_t w_off, int ldw, size_t b_off, int nin, int x_off, int y_off, int relu) {
        if (seq && nin <= 1280) hipLaunchKernelGGL(rowvec_tile_kernel<5>, dim3((rows + ST_TU - 1) / ST_TU, sy), dim3(64 * ST_TU), (size_t)ST_TU * 5 * 1024, st, a, w_off, ldw, b_off, rows, nin, x_off, y_off, relu);
        else if (seq && nin <= 2304) hipLaunchKernelGGL(rowvec_tile_kernel<9>, dim3((rows + ST_TU - 1) / ST_TU, sy), dim3(64 * ST_TU), (size_t)ST_TU * 9 * 1024, st, a, w_off, ldw, b_off, rows, nin, x_off, y_off, relu);
        else if (solo) hipLaunchKernelGGL(rowvec_kernel<1>, dim3(rows, by), dim3(64), 0, st, a, w_off, ldw, b_off, rows, nin, x_off, y_off, relu);
        else hipLaunchKernelGGL(rowvec_kernel<8>, dim3(rows, by), dim3(64), 0, st, a, w_off, ldw, b_off, rows, nin, x_off, y_off, relu);
    };
    // Laplace heads of up to 16 rows over up to 512 inputs, K <= 8 taps, <= 1 024 input-layer elements: the prefetching tail
    const bool fast_tail = g.kind == SWN_KIND_LAPLACE && g.NO <= 16 && g.O1p <= 512 && g.K <= 8 && g.H * g.seg <= 1024 &&
                           g.seg <= 16 && a.WN <= 32;
    const int total = a.n_pro + n_steps;
    for (int it = 0; it < total; ++it) {
        // prologue positions and the very first generation step launch their own input layer; later steps
        // get it from the tail of the step before
        if (it <= a.n_pro) {
            if (g.kind == SWN_KIND_LAPLACE) hipLaunchKernelGGL(step_in_kernel<SWN_KIND_LAPLACE>, dim3(batch), dim3(256), 0, st, a, it);
            else hipLaunchKernelGGL(step_in_kernel<SWN_KIND_SOFTMAX>, dim3(batch), dim3(256), 0, st, a, it);
        }
        layers(it);
        if (it < a.n_pro) continue;
        rowvec(g.S, a.y.wsk, g.L * g.Hp, a.y.bsk, g.L * g.Hp, a.o_hcat, a.o_skip, 1);
        rowvec(g.O1, a.y.w1, g.Sp, a.y.b1, g.Sp, a.o_skip, a.o_o1, 1);
        if (a.o2_by_rowvec) rowvec(g.NO, a.y.w2, g.O1p, a.y.b2, g.O1p, a.o_o1, a.o_o2, 0);
        if (fast_tail) {
            if (g.H * g.seg <= 256) hipLaunchKernelGGL(step_tail_laplace_kernel<1>, dim3(batch), dim3(256), 0, st, a, it);
            else hipLaunchKernelGGL(step_tail_laplace_kernel<4>, dim3(batch), dim3(256), 0, st, a, it);
        } else if (g.kind == SWN_KIND_LAPLACE) hipLaunchKernelGGL(step_tail_kernel<SWN_KIND_LAPLACE>, dim3(batch), dim3(256), 0, st, a, it);
        else hipLaunchKernelGGL(step_tail_kernel<SWN_KIND_SOFTMAX>, dim3(batch), dim3(256), 0, st, a, it);
    }
    return swn_launch_status("swn_decode(stepped)");
}
