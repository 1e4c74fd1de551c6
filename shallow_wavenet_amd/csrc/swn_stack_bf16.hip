// bf16 MFMA teacher-forced stack for the BL6 class (H=64, K=2, S=O1=128, Laplace head) on gfx950.
// Same math as swn_stack.hip (CSWNV.forward, cswnv_shift1.py:191-267) with bf16 weights and hidden
// states, fp32 accumulation and fp32 gate / conditioning arithmetic - the training-speed path of
// BASELINE config 4; the fp32 file stays the 1e-5 parity path.
//
// Layout: hidden states are TIME-MAJOR bf16  hs[l][b][t][64]  (128 B per position), so the eight
// consecutive channels an MFMA B-fragment lane needs are one 16-byte load.
//
//   bf16_layer_kernel  one gated layer: D[128 rows][64 pos] = Wd[128][128] . [h(t-dil) ; h(t)]
//                      v_mfma_f32_16x16x32_bf16, A fragments (the whole layer matrix, 32 KB) resident
//                      in VGPRs of a persistent workgroup, B fragments loaded straight from HBM/L2,
//                      epilogue: hoisted conditioning + sigmoid/tanh/highway in fp32, h' staged in
//                      LDS and stored as full 128-B rows.  Algorithmic HBM bytes per position and
//                      layer: 128 (read h) + 128 (write h') = 256 B; 32 768 MAC -> AI 256 flop/B < ridge.
//   bf16_head_kernel   skip (one GEMM over the 6 concatenated hidden states, K=384) -> relu ->
//                      out_1 -> relu -> out_2, intermediates through LDS, fp32 (B, n_out, Tp) output.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include "swn_geom.hpp"

namespace {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

constexpr int H = 64;
constexpr int TN = 64;                 // positions per tile

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
// bf16 path: results are rounded to 8 mantissa bits, so the plain transcendental unit (1 ulp exp2/rcp)
// is ample: sigmoid 4 instructions, tanh 5
__device__ __forceinline__ float sigm(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * fmaxf(x, -80.f)));
}
__device__ __forceinline__ float tanh_c(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.88539008f * fminf(x, 40.f)));
}
__device__ __forceinline__ unsigned short f2bf(float x) {
    return __builtin_bit_cast(unsigned short, (__bf16)x);      // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN kept
}
__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }

struct BfArgs {
    const float* P;              // fp32 packed parameters (biases, conditioning constants)
    SwnLayout y;
    const unsigned short* wbf;   // fragment-ordered bf16 weights
    const float* cond;           // (B, Tf, N)
    const float* audio;          // (B, Tp + seg - 1)
    unsigned short* hs;          // [L+1][B][Tp][64] bf16
    float* out;                  // (B, NO, Tp)
    int B, Tf, Tp, U, N, L, seg, NO, coff;
    size_t off_wd, off_wsk, off_w1, off_w2;   // element offsets into wbf
};

// ---- fp32 packed -> fragment-ordered bf16:  dst[(mt*KS + ks)*64 + lane][8] = W[16mt + (lane&15)][32ks + 8(lane>>4) + j]
__global__ void pack_frag_kernel(const float* __restrict__ src, int ld, int rows, int cols, int MT, int KS,
                                 unsigned short* __restrict__ dst) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= MT * KS * 64 * 8) return;
    const int j = e & 7, lane = (e >> 3) & 63, ks = (e >> 9) % KS, mt = (e >> 9) / KS;
    const int r = 16 * mt + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + j;
    dst[e] = (r < rows && k < cols) ? f2bf(src[(size_t)r * ld + k]) : (unsigned short)0;
}

// ---- input layer: h0[b][t][o] = softsign(causal(lift(audio)))  -> bf16 time-major
__global__ __launch_bounds__(256) void bf16_input_kernel(const BfArgs a) {
    const int o = threadIdx.x & 63, b = blockIdx.y;
    const float* au = a.audio + (size_t)b * (a.Tp + a.seg - 1);
    const float cb = a.P[a.y.cb + o];
    const float v0 = a.P[a.y.cv + o], v1 = a.P[a.y.cv + H + o], c0 = a.P[a.y.cc + o], c1 = a.P[a.y.cc + H + o];
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int t = blockIdx.x * 64 + 4 * i + (threadIdx.x >> 6);
        if (t >= a.Tp) break;
        const int ai = t + a.seg - 1;
        float acc = cb;
        if (ai - 1 >= 0) acc += fmaf(v0, au[ai - 1], c0);
        acc += fmaf(v1, au[ai], c1);
        a.hs[((size_t)b * a.Tp + t) * H + o] = f2bf(acc / (1.f + fabsf(acc)));
    }
}

// ---- one gated layer --------------------------------------------------------------------------
// Wave-independent and software-pipelined: no barriers in the loop.  ONE wave owns all 128 rows of a
// 16-position chunk (A fragments of the whole layer matrix = 128 VGPRs, resident), so every B
// fragment is loaded exactly once per chunk.  The rows of A are permuted (pack_wd_kernel) so that the
// 16 channels a lane finishes are the very channels its own tap-1 B fragments hold:
//   lane (n, g), M-tile m, reg r  <->  channel chan(m,g,r) = 32*(m>>1) + 8g + 4*(m&1) + r
// => the highway input h(t) needs no extra load and h' leaves as two 16-byte stores per lane.
__device__ __host__ constexpr int chan_of(int m, int g, int r) { return 32 * (m >> 1) + 8 * g + 4 * (m & 1) + r; }

// A fragments with the row permutation: dst[l][mt(8)][ks(4)][lane][8]
__global__ void pack_wd_kernel(const float* __restrict__ wd, int L, unsigned short* __restrict__ dst) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= L * 8 * 4 * 512) return;
    const int j = e & 7, lane = (e >> 3) & 63, ks = (e >> 9) & 3, mt = (e >> 11) & 7, l = e >> 14;
    const int g = (lane & 15) >> 2, r = lane & 3;                 // A row inside the M-tile = lane&15 = 4g + r
    const int row = (mt >= 4 ? H : 0) + chan_of(mt & 3, g, r);
    const int k = 32 * ks + 8 * (lane >> 4) + j;
    dst[e] = f2bf(wd[((size_t)l * 128 + row) * 128 + k]);
}

struct LayerOps {
    bf16x8 b[4];        // B fragments: K-steps 0,1 = h(t-dil) channels 0-31 / 32-63, 2,3 = h(t)
    float4 cz[4], cc[4];// hoisted in_x rows of this lane's 16 channels (gate | candidate), first segment
    float wu;           // upsampler tap of position t (seg == 1 fast path)
};
constexpr int NSUB = 2;  // 16-position chunks handled per loop iteration: twice the bytes in flight per wave

// ONE wave owns all 128 rows of a 16-position chunk (A fragments of the whole layer matrix = 128
// VGPRs, resident; one wave per SIMD with the full 512-entry register file), so every B fragment is
// loaded exactly once; the four waves of a workgroup walk different chunks.  Measured alternatives
// (4 waves sharing a chunk, or 2 waves each owning half the channels at 2 waves/SIMD) were 1.5-1.7x
// slower: the vector-memory path, not MFMA, is what this kernel saturates.
__global__ __launch_bounds__(256, 1) void bf16_layer_kernel(const BfArgs a, const int l, const int dil, const int n_chunks) {
    __shared__ __attribute__((aligned(16))) float cst[2 * 128];      // bd[128] | bx[128] of this layer
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform -> scalar chunk arithmetic
    const int n = lane & 15, g = lane >> 4;
    cst[tid] = tid < 128 ? a.P[a.y.bd + (size_t)l * 128 + tid] : a.P[a.y.bx + (size_t)l * 128 + tid - 128];
    bf16x8 A[8][4];
    {
        const bf16x8* src = reinterpret_cast<const bf16x8*>(a.wbf + a.off_wd) + (size_t)l * 8 * 4 * 64;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) A[mt][ks] = src[(mt * 4 + ks) * 64 + lane];
    }
    __syncthreads();
    const size_t lstride = (size_t)a.B * a.Tp * H;
    const unsigned short* hprev = a.hs + (size_t)l * lstride;
    unsigned short* hnext = a.hs + (size_t)(l + 1) * lstride;
    const int chunks_per_b = (a.Tp + 15) / 16;

    auto fetch = [&](int c, LayerOps& op) {                          // c is wave-uniform: scalar divisions
        const int b = c / chunks_per_b, t0 = (c - b * chunks_per_b) * 16, t = t0 + n;
        const unsigned short* hb = hprev + (size_t)b * a.Tp * H;
        const bool ok = t < a.Tp;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ts = t - (ks < 2 ? dil : 0);
            op.b[ks] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            if (ok && ts >= 0) op.b[ks] = *reinterpret_cast<const bf16x8*>(hb + (size_t)ts * H + 32 * (ks & 1) + 8 * g);
        }
        const int tt0 = t0 + a.coff;
        int f = tt0 / a.U, jj = tt0 - f * a.U + n;                   // scalar division, per-lane carry
        if (jj >= a.U) { jj -= a.U; f += 1; }
        f = f < a.Tf ? f : a.Tf - 1;
        op.wu = a.P[a.y.wup + jj];
        const float* cr = a.cond + ((size_t)b * a.Tf + f) * a.N + (size_t)l * a.seg * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q) {              // q: channel group (q>>1)*32 + 8g + 4*(q&1)
            op.cz[q] = *reinterpret_cast<const float4*>(cr + 32 * (q >> 1) + 8 * g + 4 * (q & 1));
            op.cc[q] = *reinterpret_cast<const float4*>(cr + H + 32 * (q >> 1) + 8 * g + 4 * (q & 1));
        }
    };

    const int stride = gridDim.x * 4 * NSUB;                         // every wave walks its own chunk sequence
    int c0 = (blockIdx.x * 4 + w) * NSUB;
    if (c0 >= n_chunks) return;
    LayerOps curs[NSUB], nxts[NSUB];
#pragma unroll
    for (int u = 0; u < NSUB; ++u) fetch(c0 + u < n_chunks ? c0 + u : n_chunks - 1, curs[u]);
    for (; c0 < n_chunks; c0 += stride) {
        const int cn0 = c0 + stride;
        if (cn0 < n_chunks) {
#pragma unroll
            for (int u = 0; u < NSUB; ++u) fetch(cn0 + u < n_chunks ? cn0 + u : n_chunks - 1, nxts[u]);
        }
#pragma unroll
      for (int u = 0; u < NSUB; ++u) {
        const int c = c0 + u;
        if (c >= n_chunks) break;
        const LayerOps& cur = curs[u];
        const int b = c / chunks_per_b, t = (c - b * chunks_per_b) * 16 + n;
        f32x4 acc[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[mt][ks], cur.b[ks], acc[mt], 0, 0, 0);
        }
        if (t < a.Tp) {
            unsigned short hv[16];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int c0 = chan_of(m, g, 0);
                const float4 bdz = *reinterpret_cast<const float4*>(cst + c0);
                const float4 bdc = *reinterpret_cast<const float4*>(cst + H + c0);
                const float4 bxz = *reinterpret_cast<const float4*>(cst + 128 + c0);
                const float4 bxc = *reinterpret_cast<const float4*>(cst + 128 + H + c0);
                const float bdzv[4] = {bdz.x, bdz.y, bdz.z, bdz.w}, bdcv[4] = {bdc.x, bdc.y, bdc.z, bdc.w};
                float gz[4] = {bxz.x, bxz.y, bxz.z, bxz.w}, gc[4] = {bxc.x, bxc.y, bxc.z, bxc.w};
                if (a.seg == 1) {
                    const float czv[4] = {cur.cz[m].x, cur.cz[m].y, cur.cz[m].z, cur.cz[m].w};
                    const float ccv[4] = {cur.cc[m].x, cur.cc[m].y, cur.cc[m].z, cur.cc[m].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) { gz[r] = fmaf(cur.wu, czv[r], gz[r]); gc[r] = fmaf(cur.wu, ccv[r], gc[r]); }
                } else {
                    for (int s = 0; s < a.seg; ++s) {
                        const int tt = t + s + a.coff;
                        int f = tt / a.U; const int jj = tt - f * a.U;
                        f = f < a.Tf ? f : a.Tf - 1;
                        const float wu = a.P[a.y.wup + jj];
                        const float* cr = a.cond + ((size_t)b * a.Tf + f) * a.N + (size_t)(l * a.seg + s) * 128 + c0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { gz[r] = fmaf(wu, cr[r], gz[r]); gc[r] = fmaf(wu, cr[H + r], gc[r]); }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hp = (float)cur.b[2 + (m >> 1)][(m & 1) * 4 + r];      // h(t)[chan]: own tap-1 fragment
                    const float z = sigm(gz[r] * (acc[m][r] + bdzv[r]));
                    const float cd = tanh_c(gc[r] * (acc[4 + m][r] + bdcv[r]));
                    hv[m * 4 + r] = f2bf(fmaf(z, hp - cd, cd));                        // (1-z) c + z h
                }
            }
            uint4 o0, o1;
            o0.x = hv[0] | ((unsigned)hv[1] << 16);   o0.y = hv[2] | ((unsigned)hv[3] << 16);
            o0.z = hv[4] | ((unsigned)hv[5] << 16);   o0.w = hv[6] | ((unsigned)hv[7] << 16);
            o1.x = hv[8] | ((unsigned)hv[9] << 16);   o1.y = hv[10] | ((unsigned)hv[11] << 16);
            o1.z = hv[12] | ((unsigned)hv[13] << 16); o1.w = hv[14] | ((unsigned)hv[15] << 16);
            unsigned short* dst = hnext + ((size_t)b * a.Tp + t) * H;
            *reinterpret_cast<uint4*>(dst + 8 * g) = o0;
            *reinterpret_cast<uint4*>(dst + 32 + 8 * g) = o1;
        }
      }
#pragma unroll
        for (int u = 0; u < NSUB; ++u) curs[u] = nxts[u];
    }
}

// ---- head: skip (K = L*64) -> relu -> out_1 (128x128) -> relu -> out_2 (<=16 x 128) ----------------
template <int LL>
__global__ __launch_bounds__(256) void bf16_head_kernel(const BfArgs a, const int n_tiles) {
    constexpr int ROW2 = 256 + 16;               // [pos][128 ch] bf16 tile pitch
    __shared__ __attribute__((aligned(16))) unsigned char t1[TN * ROW2];
    __shared__ __attribute__((aligned(16))) unsigned char t2[TN * ROW2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    constexpr int KS1 = LL * 2;                            // K-steps of the skip GEMM
    // resident A fragments: skip rows 32w..32w+31 (2 M-tiles), out_1 the same rows, out_2 one M-tile
    bf16x8 Ask[2][KS1], A1[2][4], A2[4];
    {
        const bf16x8* s0 = reinterpret_cast<const bf16x8*>(a.wbf + a.off_wsk);
        const bf16x8* s1 = reinterpret_cast<const bf16x8*>(a.wbf + a.off_w1);
        const bf16x8* s2 = reinterpret_cast<const bf16x8*>(a.wbf + a.off_w2);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) Ask[m][ks] = s0[((2 * w + m) * KS1 + ks) * 64 + lane];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) A1[m][ks] = s1[((2 * w + m) * 4 + ks) * 64 + lane];
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) A2[ks] = s2[ks * 64 + lane];
    }
    float bsk[2][4], b1[2][4], b2[4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bsk[m][r] = a.P[a.y.bsk + 16 * (2 * w + m) + 4 * g + r];
            b1[m][r] = a.P[a.y.b1 + 16 * (2 * w + m) + 4 * g + r];
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) b2[r] = (4 * g + r < a.NO) ? a.P[a.y.b2 + 4 * g + r] : 0.f;
    const size_t lstride = (size_t)a.B * a.Tp * H;
    const int tiles_per_b = (a.Tp + TN - 1) / TN;

    for (int tix = blockIdx.x; tix < n_tiles; tix += gridDim.x) {
        const int b = tix / tiles_per_b, t0 = (tix - b * tiles_per_b) * TN;
        // ---- skip = Wsk . [h_1 .. h_L]
        f32x4 acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int t = t0 + 16 * nt + n;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                bf16x8 bf = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                if (t < a.Tp)
                    bf = *reinterpret_cast<const bf16x8*>(a.hs + (size_t)(1 + ks / 2) * lstride +
                                                          ((size_t)b * a.Tp + t) * H + 32 * (ks & 1) + 8 * g);
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ask[0][ks], bf, acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ask[1][ks], bf, acc[1][nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                unsigned short hv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = f2bf(fmaxf(acc[m][nt][r] + bsk[m][r], 0.f));
                uint2 pk; pk.x = (unsigned)hv[0] | ((unsigned)hv[1] << 16); pk.y = (unsigned)hv[2] | ((unsigned)hv[3] << 16);
                *reinterpret_cast<uint2*>(t1 + (16 * nt + n) * ROW2 + (16 * (2 * w + m) + 4 * g) * 2) = pk;
            }
        __syncthreads();
        // ---- out_1
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[m][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(t1 + (16 * nt + n) * ROW2 + (32 * ks + 8 * g) * 2);
                acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[0][ks], bf, acc[0][nt], 0, 0, 0);
                acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[1][ks], bf, acc[1][nt], 0, 0, 0);
            }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                unsigned short hv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) hv[r] = f2bf(fmaxf(acc[m][nt][r] + b1[m][r], 0.f));
                uint2 pk; pk.x = (unsigned)hv[0] | ((unsigned)hv[1] << 16); pk.y = (unsigned)hv[2] | ((unsigned)hv[3] << 16);
                *reinterpret_cast<uint2*>(t2 + (16 * nt + n) * ROW2 + (16 * (2 * w + m) + 4 * g) * 2) = pk;
            }
        __syncthreads();
        // ---- out_2: one 16-row M-tile; wave w takes N-tile w
        {
            f32x4 o2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(t2 + (16 * w + n) * ROW2 + (32 * ks + 8 * g) * 2);
                o2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2[ks], bf, o2, 0, 0, 0);
            }
            const int t = t0 + 16 * w + n;
            if (t < a.Tp) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * g + r;
                    if (row < a.NO) a.out[((size_t)b * a.NO + row) * a.Tp + t] = o2[r] + b2[r];
                }
            }
        }
        __syncthreads();
    }
}

int bf_geom(const swn_net_desc* d, SwnGeom* g) {
    int rc = swn_make_geom(d, g);
    if (rc < 0) return rc;
    if (!g->bl6 || g->kind != SWN_KIND_LAPLACE || g->S != 128 || g->NO > 16) return SWN_E_UNSUPPORTED;
    return SWN_OK;
}

struct BfOffsets { size_t wd, wsk, w1, w2, total; };
BfOffsets bf_offsets(const SwnGeom& g) {
    BfOffsets o;
    o.wd = 0;
    o.wsk = o.wd + (size_t)g.L * 8 * 4 * 512;           // [L][8 mt][4 ks][64 lanes][8]
    o.w1 = o.wsk + (size_t)8 * (g.L * 2) * 512;
    o.w2 = o.w1 + (size_t)8 * 4 * 512;
    o.total = o.w2 + (size_t)1 * 4 * 512;
    return o;
}

}  // namespace

extern "C" size_t swn_bf16_weight_bytes(const swn_net_desc* d) {
    SwnGeom g; if (bf_geom(d, &g) < 0) return 0;
    return bf_offsets(g).total * sizeof(unsigned short);
}

extern "C" int swn_pack_bf16(const swn_net_desc* d, const float* packed, void* wbf_, void* stream_) {
    SwnGeom g; int rc = bf_geom(d, &g);
    if (rc < 0) return rc;
    if (!packed || !wbf_) return SWN_E_BADARG;
    SwnLayout y; swn_make_layout(&g, &y);
    const BfOffsets o = bf_offsets(g);
    unsigned short* wbf = reinterpret_cast<unsigned short*>(wbf_);
    hipStream_t st = (hipStream_t)stream_;
    (void)hipGetLastError();
    hipLaunchKernelGGL(pack_wd_kernel, dim3((g.L * 8 * 4 * 512 + 255) / 256), dim3(256), 0, st,
                       packed + y.wd, g.L, wbf + o.wd);
    hipLaunchKernelGGL(pack_frag_kernel, dim3((8 * g.L * 2 * 512 + 255) / 256), dim3(256), 0, st,
                       packed + y.wsk, g.L * 64, 128, g.L * 64, 8, g.L * 2, wbf + o.wsk);
    hipLaunchKernelGGL(pack_frag_kernel, dim3((8 * 4 * 512 + 255) / 256), dim3(256), 0, st,
                       packed + y.w1, g.Sp, 128, 128, 8, 4, wbf + o.w1);
    hipLaunchKernelGGL(pack_frag_kernel, dim3((4 * 512 + 255) / 256), dim3(256), 0, st,
                       packed + y.w2, g.O1p, g.NO, 128, 1, 4, wbf + o.w2);
    return swn_launch_status("swn_pack_bf16");
}

extern "C" size_t swn_forward_bf16_work_bytes(const swn_net_desc* d, int batch, int n_frames) {
    SwnGeom g; if (bf_geom(d, &g) < 0 || batch < 1 || n_frames < 1) return 0;
    const long Tp = (long)n_frames * g.U - 2 * g.seg + 1;
    if (Tp < 1) return 0;
    return (size_t)(g.L + 1) * batch * Tp * H * sizeof(unsigned short);
}

extern "C" int swn_forward_bf16(const swn_net_desc* d, const float* packed, const void* wbf, const float* cond,
                                const float* audio, int batch, int n_frames, void* work, float* out, void* stream_) {
    SwnGeom g; int rc = bf_geom(d, &g);
    if (rc < 0) return rc;
    if (!packed || !wbf || !cond || !audio || !work || !out || batch < 1 || batch > 65535 || n_frames < 1) return SWN_E_BADARG;
    const long Tp = (long)n_frames * g.U - 2 * g.seg + 1;
    if (Tp < 1) return SWN_E_BADARG;
    BfArgs a;
    swn_make_layout(&g, &a.y);
    const BfOffsets o = bf_offsets(g);
    a.P = packed; a.wbf = reinterpret_cast<const unsigned short*>(wbf); a.cond = cond; a.audio = audio;
    a.hs = reinterpret_cast<unsigned short*>(work); a.out = out;
    a.B = batch; a.Tf = n_frames; a.Tp = (int)Tp; a.U = g.U; a.N = g.N; a.L = g.L; a.seg = g.seg; a.NO = g.NO;
    a.coff = g.seg;
    a.off_wd = o.wd; a.off_wsk = o.wsk; a.off_w1 = o.w1; a.off_w2 = o.w2;
    hipStream_t st = (hipStream_t)stream_;
    (void)hipGetLastError();
    hipLaunchKernelGGL(bf16_input_kernel, dim3((unsigned)((Tp + 63) / 64), batch), dim3(256), 0, st, a);
    const int n_tiles = batch * (int)((Tp + TN - 1) / TN);
    const int n_chunks = batch * (int)((Tp + 15) / 16);
    const int grid = (n_chunks + 7) / 8 < 256 ? (n_chunks + 7) / 8 : 256;      // persistent: one workgroup (4 waves x 512 VGPRs) per CU
    for (int l = 0; l < g.L; ++l)
        hipLaunchKernelGGL(bf16_layer_kernel, dim3(grid), dim3(256), 0, st, a, l, g.dil[l], n_chunks);
    const int hgrid = n_tiles < 512 ? n_tiles : 512;
    hipLaunchKernelGGL(bf16_head_kernel<6>, dim3(hgrid), dim3(256), 0, st, a, n_tiles);
    return swn_launch_status("swn_forward_bf16");
}
