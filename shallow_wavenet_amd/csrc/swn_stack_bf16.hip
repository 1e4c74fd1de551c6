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
#include <cstdlib>

namespace {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;

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
    const unsigned short* gx16;  // dropout mode: sample-rate in_x products (bias included) [B][Tp][L*128] bf16, else null
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
// thread = (position, group of 8 channels): one 16-byte store per lane, a wave writes 8 whole 128-byte rows
__global__ __launch_bounds__(256) void bf16_input_kernel(const BfArgs a) {
    const int cg = threadIdx.x & 7, b = blockIdx.y;
    const float* au = a.audio + (size_t)b * (a.Tp + a.seg - 1);
    float cb[8], v0[8], v1[8], c0[8], c1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int o = 8 * cg + e;
        cb[e] = a.P[a.y.cb + o];
        v0[e] = a.P[a.y.cv + o]; v1[e] = a.P[a.y.cv + H + o];
        c0[e] = a.P[a.y.cc + o]; c1[e] = a.P[a.y.cc + H + o];
    }
#pragma unroll 2
    for (int i = 0; i < 8; ++i) {
        const int t = blockIdx.x * 256 + 32 * i + (threadIdx.x >> 3);
        if (t >= a.Tp) break;
        const int ai = t + a.seg - 1;
        const float x1 = au[ai], x0 = ai - 1 >= 0 ? au[ai - 1] : 0.f;
        const bool has0 = ai - 1 >= 0;
        unsigned wd[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            float r[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                float acc = cb[e + q];
                if (has0) acc += fmaf(v0[e + q], x0, c0[e + q]);
                acc += fmaf(v1[e + q], x1, c1[e + q]);
                r[q] = acc * __builtin_amdgcn_rcpf(1.f + fabsf(acc));      // 1-ulp reciprocal: the result is rounded to bf16 anyway (an IEEE divide made this kernel VALU-bound)
            }
            wd[e >> 1] = (unsigned)f2bf(r[0]) | ((unsigned)f2bf(r[1]) << 16);
        }
        *reinterpret_cast<uint4*>(a.hs + ((size_t)b * a.Tp + t) * H + 8 * cg) = make_uint4(wd[0], wd[1], wd[2], wd[3]);
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

// Buffer-resource loads: one 32-bit per-lane byte offset per load (an out-of-range offset reads zeros, which is
// how the causal zero padding and ragged tails are produced) instead of 64-bit address arithmetic per load.
constexpr unsigned OOB = 0x80000000u;   // host guarantees every buffer is smaller than 2 GiB
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 buf_ld_bf8(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ float4 buf_ld_f4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ float buf_ld_f1(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}

struct BFrag { bf16x8 b[4]; };          // B fragments: K-steps 0,1 = h(t-dil) channels 0-31 / 32-63, 2,3 = h(t)
struct CondOps {
    float4 cz[4], cc[4];                // hoisted in_x rows of this lane's 16 channels (gate | candidate)
    float wu;                           // upsampler tap of position t
};
struct GxOps { u32x4 z[2], c[2]; };     // dropout mode: the in_x products of position t, channels 8g.. / 32 + 8g.. (gate | candidate), bf16
constexpr int NSLOT = 8;                // B-fragment ring (AccVGPRs): chunk k+7 is fetched while chunk k is computed:
                                        // ~7 KB of HBM reads in flight per wave, what the latency-bandwidth product needs
constexpr float K_SIG = -1.44269504f;   // sigmoid(x) = 1 / (1 + 2^(K_SIG x))
constexpr float K_TANH = 2.88539008f;   // tanh(x)    = 1 - 2 / (1 + 2^(K_TANH x))


// The 32 MFMAs of one chunk, written as asm so that the register classes are what the loop needs: the
// layer matrix (A, 128 registers) and the tap-0 B fragments live in AccVGPRs, which MFMA reads directly,
// while the accumulators are produced in architectural VGPRs where the epilogue's vector ALU can use them.
// (Left to the allocator, ~100 v_accvgpr_read copies per chunk were issued - a quarter of the loop.)
// K-step-major order keeps dependent MFMAs 8 instructions apart; the first K-step takes the dil_h bias as
// its C operand.  The compiler's hazard recognizer cannot see through asm, so the blocks carry their own
// wait states: a leading s_nop for an operand a vector-ALU copy may have written just before, a trailing
// s_nop 15 for the MFMA-result -> VALU-read hazard (8-pass XDL op: 11 required).  Fragments the epilogue
// also reads with the vector ALU (the tap-1 rows, for the highway term) are taken from VGPRs as loaded.
#define SWN_MF "v_mfma_f32_16x16x32_bf16 "
template <class BT>
__device__ __forceinline__ void mfma_first(f32x4 (&acc)[8], const bf16x8 (&A)[8][4], const BT& b, const f32x4 (&c)[8]) {
    asm("s_nop 3\n\t" SWN_MF "%0, %8, %16, %17\n\t" SWN_MF "%1, %9, %16, %18\n\t" SWN_MF "%2, %10, %16, %19\n\t"
        SWN_MF "%3, %11, %16, %20\n\t" SWN_MF "%4, %12, %16, %21\n\t" SWN_MF "%5, %13, %16, %22\n\t"
        SWN_MF "%6, %14, %16, %23\n\t" SWN_MF "%7, %15, %16, %24"
        : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3]), "=&v"(acc[4]), "=&v"(acc[5]), "=&v"(acc[6]), "=&v"(acc[7])
        : "a"(A[0][0]), "a"(A[1][0]), "a"(A[2][0]), "a"(A[3][0]), "a"(A[4][0]), "a"(A[5][0]), "a"(A[6][0]), "a"(A[7][0]),
          "a"(b), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]));
}
template <int KS, bool TAIL, bool B_ACC>
__device__ __forceinline__ void mfma_next(f32x4 (&acc)[8], const bf16x8 (&A)[8][4], const bf16x8& b) {
#define SWN_MF8 SWN_MF "%0, %8, %16, %0\n\t" SWN_MF "%1, %9, %16, %1\n\t" SWN_MF "%2, %10, %16, %2\n\t" \
                SWN_MF "%3, %11, %16, %3\n\t" SWN_MF "%4, %12, %16, %4\n\t" SWN_MF "%5, %13, %16, %5\n\t" \
                SWN_MF "%6, %14, %16, %6\n\t" SWN_MF "%7, %15, %16, %7"
#define SWN_MF_OPS(BC)                                                                                          \
        : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]) \
        : "a"(A[0][KS]), "a"(A[1][KS]), "a"(A[2][KS]), "a"(A[3][KS]), "a"(A[4][KS]), "a"(A[5][KS]), "a"(A[6][KS]),      \
          "a"(A[7][KS]), BC(b)
    if (TAIL && B_ACC) asm("s_nop 3\n\t" SWN_MF8 "\n\ts_nop 15" SWN_MF_OPS("a"));
    else if (TAIL) asm("s_nop 3\n\t" SWN_MF8 "\n\ts_nop 15" SWN_MF_OPS("v"));
    else if (B_ACC) asm("s_nop 3\n\t" SWN_MF8 SWN_MF_OPS("a"));
    else asm("s_nop 3\n\t" SWN_MF8 SWN_MF_OPS("v"));
#undef SWN_MF_OPS
#undef SWN_MF8
}
#undef SWN_MF

// ONE wave owns all 128 rows of a 16-position chunk (A fragments of the whole layer matrix = 128
// registers, resident; one wave per SIMD with the full 512-entry register file), so every B fragment is
// loaded exactly once; the four waves of a workgroup walk different chunks and never synchronise.
// The loop is bound by vector-ALU issue (the gate epilogue), not by MFMA or HBM, so everything around
// the epilogue is kept off the vector ALU: chunk arithmetic is scalar, addresses are 32-bit buffer
// offsets, the sigmoid/tanh scale factors are folded into the conditioning constants, and saturation
// needs no clamps (2^x -> inf -> rcp -> 0 gives the exact limits).
// CM: where the conditioning comes from.  0: hoisted rows, any seg (plain loads)   1: seg == 1 and U >= 16, hoisted rows
// prefetched through the buffer path   2: dropout mode (aux_drop acts at sample rate, cswnv_shift1.py:194-195): the in_x
// products themselves, one bf16 row per position (a.gx16, made by a GEMM over the masked conditioning)
template <int CM>
__global__ __launch_bounds__(256, 1) void bf16_layer_kernel(const BfArgs a, const int l, const int dil, const int n_chunks) {
    __shared__ __attribute__((aligned(16))) float cst[2 * 128];      // bd[128] | prescaled bx[128] of this layer
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform -> scalar chunk arithmetic
    const int n = lane & 15, g = lane >> 4;
    cst[tid] = tid < 128 ? a.P[a.y.bd + (size_t)l * 128 + tid]
                         : a.P[a.y.bx + (size_t)l * 128 + tid - 128] * (tid < 192 ? K_SIG : K_TANH);
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
    const __amdgpu_buffer_rsrc_t rh = make_rsrc(a.hs + (size_t)l * lstride, lstride * 2);
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(a.cond, (size_t)a.B * a.Tf * a.N * 4);
    const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.P + a.y.wup, (size_t)a.U * 4);
    const unsigned gx_row = (unsigned)a.L * 256u;                    // bytes of one position's in_x products
    const __amdgpu_buffer_rsrc_t rg = make_rsrc(a.gx16, CM == 2 ? (size_t)a.B * a.Tp * gx_row : 0);
    unsigned short* hnext = a.hs + (size_t)(l + 1) * lstride;
    const int chunks_per_b = (a.Tp + 15) / 16;
    const unsigned lane_h = (unsigned)(n * H + 8 * g) * 2u, lane_c = (unsigned)(8 * g) * 4u;
    const unsigned dil_bytes = (unsigned)dil * H * 2u;

    auto fetch_b = [&](int c, BFrag& f) {                            // c is wave-uniform
        c = c < n_chunks ? c : n_chunks - 1;
        const int b = c / chunks_per_b, t0 = (c - b * chunks_per_b) * 16, t = t0 + n;
        const unsigned base = (unsigned)(b * a.Tp + t0) * (H * 2u) + lane_h;
        const unsigned o1 = t < a.Tp ? base : OOB;
        const unsigned o0 = (t < a.Tp && t >= dil) ? base - dil_bytes : OOB;
        f.b[0] = buf_ld_bf8(rh, o0); f.b[1] = buf_ld_bf8(rh, o0 + 64u);
        f.b[2] = buf_ld_bf8(rh, o1); f.b[3] = buf_ld_bf8(rh, o1 + 64u);
    };
    auto fetch_c = [&](int c, CondOps& k) {                          // seg == 1 conditioning of chunk c
        c = c < n_chunks ? c : n_chunks - 1;
        const int b = c / chunks_per_b, t0 = (c - b * chunks_per_b) * 16;
        const int tt0 = t0 + a.coff;
        int f0 = tt0 / a.U;                                          // scalar division
        const int j0 = tt0 - f0 * a.U;
        int jj = j0 + n, up = 0;
        if (jj >= a.U) { jj -= a.U; up = 1; }                        // U >= 16: at most one frame boundary inside a chunk
        const int last = a.Tf - 1;
        int f = f0 + up; f = f < last ? f : last;
        f0 = f0 < last ? f0 : last;
        const unsigned rowb = (unsigned)((b * a.Tf + f0) * a.N + l * 128) * 4u + lane_c;
        const unsigned off = rowb + (unsigned)(f - f0) * (unsigned)(a.N * 4);
        k.wu = buf_ld_f1(ru, (unsigned)jj * 4u);
#pragma unroll
        for (int q = 0; q < 4; ++q) {              // q: channel group (q>>1)*32 + 8g + 4*(q&1)
            k.cz[q] = buf_ld_f4(rc, off + (unsigned)(32 * (q >> 1) + 4 * (q & 1)) * 4u);
            k.cc[q] = buf_ld_f4(rc, off + (unsigned)(H + 32 * (q >> 1) + 4 * (q & 1)) * 4u);
        }
    };

    auto fetch_g = [&](int c, GxOps& k) {                            // dropout mode: in_x products of chunk c
        c = c < n_chunks ? c : n_chunks - 1;
        const int b = c / chunks_per_b, t0 = (c - b * chunks_per_b) * 16, t = t0 + n;
        const unsigned off = t < a.Tp ? (unsigned)(b * a.Tp + t) * gx_row + (unsigned)l * 256u + 16u * g : OOB;
        k.z[0] = __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0);
        k.z[1] = __builtin_amdgcn_raw_buffer_load_b128(rg, off + 64u, 0, 0);
        k.c[0] = __builtin_amdgcn_raw_buffer_load_b128(rg, off + 128u, 0, 0);
        k.c[1] = __builtin_amdgcn_raw_buffer_load_b128(rg, off + 192u, 0, 0);
    };

    // per-lane epilogue constants of its 16 channels stay in registers (LDS reads inside the epilogue stalled it)
    f32x4 kbd[8];                      // dil_h bias of accumulator tile mt: the C operand of its first MFMA
    float4 kbxz[4], kbxc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int ch = chan_of(m, g, 0);
        kbd[m] = *reinterpret_cast<const f32x4*>(cst + ch);
        kbd[4 + m] = *reinterpret_cast<const f32x4*>(cst + H + ch);
        kbxz[m] = *reinterpret_cast<const float4*>(cst + 128 + ch);
        kbxc[m] = *reinterpret_cast<const float4*>(cst + 128 + H + ch);
    }
    const int W = gridDim.x * 4;                                     // every wave walks its own chunk sequence
    int c = blockIdx.x * 4 + w;
    if (c >= n_chunks) return;
    constexpr bool fast = CM == 1, gxm = CM == 2;
    BFrag ring[NSLOT];
    CondOps cnd[2];
    GxOps gxo[2];
    if (fast) fetch_c(c, cnd[0]);
    if (gxm) fetch_g(c, gxo[0]);
#pragma unroll
    for (int u = 0; u < NSLOT - 1; ++u) fetch_b(c + u * W, ring[u]);
    // one round = NSLOT chunks with static ring slots.  The first round is peeled (called once ahead of the
    // loop) so that the loop header sees the steady-state queue of outstanding loads and the compiler's
    // vmcnt waits stay exact instead of draining the prefetch ring every round.
    auto round = [&]() __attribute__((always_inline)) -> bool {
#pragma unroll
        for (int u = 0; u < NSLOT; ++u) {
            if (c >= n_chunks) return false;
            if (fast) fetch_c(c + W, cnd[(u + 1) & 1]);
            if (gxm) fetch_g(c + W, gxo[(u + 1) & 1]);
            fetch_b(c + (NSLOT - 1) * W, ring[(u + NSLOT - 1) % NSLOT]);
            const BFrag& cur = ring[u];
            const CondOps& k = cnd[u & 1];
            const GxOps& kg = gxo[u & 1];
            const int b = c / chunks_per_b, t = (c - b * chunks_per_b) * 16 + n;
            f32x4 acc[8];                                   // = bd + Wd . [h(t-dil) ; h(t)]
            mfma_first(acc, A, cur.b[0], kbd);
            mfma_next<1, false, true>(acc, A, cur.b[1]);
            mfma_next<2, false, false>(acc, A, cur.b[2]);
            mfma_next<3, true, false>(acc, A, cur.b[3]);
            const float wuz = k.wu * K_SIG, wuc = k.wu * K_TANH;
            unsigned short hv[16];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int ch = chan_of(m, g, 0);
                const float4 bxz = kbxz[m], bxc = kbxc[m];
                float gz[4] = {bxz.x, bxz.y, bxz.z, bxz.w}, gc[4] = {bxc.x, bxc.y, bxc.z, bxc.w};
                if (gxm) {                         // the bias is part of the products
                    const u32x4 wz = kg.z[m >> 1], wc = kg.c[m >> 1];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const unsigned uz = wz[(m & 1) * 2 + (r >> 1)], uc = wc[(m & 1) * 2 + (r >> 1)];
                        gz[r] = K_SIG * __builtin_bit_cast(float, (r & 1) ? (uz & 0xffff0000u) : (uz << 16));
                        gc[r] = K_TANH * __builtin_bit_cast(float, (r & 1) ? (uc & 0xffff0000u) : (uc << 16));
                    }
                } else if (fast) {
                    const float czv[4] = {k.cz[m].x, k.cz[m].y, k.cz[m].z, k.cz[m].w};
                    const float ccv[4] = {k.cc[m].x, k.cc[m].y, k.cc[m].z, k.cc[m].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) { gz[r] = fmaf(wuz, czv[r], gz[r]); gc[r] = fmaf(wuc, ccv[r], gc[r]); }
                } else {
                    float sz[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {0.f, 0.f, 0.f, 0.f};
                    for (int s = 0; s < a.seg; ++s) {
                        const int tt = t + s + a.coff;
                        int f = tt / a.U; const int jj = tt - f * a.U;
                        f = f < a.Tf ? f : a.Tf - 1;
                        const float wu = a.P[a.y.wup + jj];
                        const float* cr = a.cond + ((size_t)b * a.Tf + f) * a.N + (size_t)(l * a.seg + s) * 128 + ch;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { sz[r] = fmaf(wu, cr[r], sz[r]); sc[r] = fmaf(wu, cr[H + r], sc[r]); }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { gz[r] = fmaf(K_SIG, sz[r], gz[r]); gc[r] = fmaf(K_TANH, sc[r], gc[r]); }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hp = (float)cur.b[2 + (m >> 1)][(m & 1) * 4 + r];      // h(t)[chan]: own tap-1 fragment
                    const float z = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(gz[r] * acc[m][r]));
                    const float q = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(gc[r] * acc[4 + m][r]));
                    const float cd = fmaf(-2.f, q, 1.f);
                    hv[m * 4 + r] = f2bf(fmaf(z, hp - cd, cd));                        // (1-z) c + z h
                }
            }
            if (t < a.Tp) {
                uint4 o0, o1;
                o0.x = hv[0] | ((unsigned)hv[1] << 16);   o0.y = hv[2] | ((unsigned)hv[3] << 16);
                o0.z = hv[4] | ((unsigned)hv[5] << 16);   o0.w = hv[6] | ((unsigned)hv[7] << 16);
                o1.x = hv[8] | ((unsigned)hv[9] << 16);   o1.y = hv[10] | ((unsigned)hv[11] << 16);
                o1.z = hv[12] | ((unsigned)hv[13] << 16); o1.w = hv[14] | ((unsigned)hv[15] << 16);
                unsigned short* dst = hnext + ((size_t)b * a.Tp + t) * H;
                *reinterpret_cast<uint4*>(dst + 8 * g) = o0;
                *reinterpret_cast<uint4*>(dst + 32 + 8 * g) = o1;
            }
            c += W;
        }
        return true;
    };
    if (!round()) return;
    while (round()) {}
}

// ---- frame-unit variant of the gated layer (seg == 1, 16 <= U <= 112): the kernel the bf16 stack runs ------
// Work is cut at conditioning-frame boundaries: a unit = the <= U positions of one utterance that share one
// conditioning frame, walked as NCH = ceil(U/16) chunks of 16 positions (the last one ragged).  Every wave owns
// units j, j + #waves, ...  Why: with one wave per SIMD every instruction of any kind costs an issue
// slot, and all vector-memory returns are counted in order (vmcnt), so
//   * the hoisted in_x row is loaded ONCE per unit (8 loads per ~7 chunks instead of 9 per chunk) and never
//     sits between the streamed h rows in the in-order return queue: the B-fragment ring can run
//     NCH-2 chunks ahead of its consumer without a conditioning wait draining it;
//   * chunk/position arithmetic is a handful of scalar adds per chunk (no divisions);
//   * ragged tails and the causal zero padding are out-of-range buffer offsets (loads return 0, stores are
//     dropped): no exec-mask branches in the loop;
//   * the 32 MFMAs of chunk i+1 are issued in four groups between the four epilogue quarters of chunk i
//     (two accumulator sets), so the matrix pipe runs under the vector-ALU-bound epilogue.
struct CondRow { float4 cz[4], cc[4]; };
struct Unit { int b, f, ts, te, jj0; };        // wave-uniform: utterance, frame, [ts, te) positions, first tap index

// SP > 1 cuts every frame into SP sub-units of NCH chunks (the last one shorter): with few frames per wave (BASELINE cfg4:
// 1 200 frames over 1 024 waves) whole-frame units leave half the waves a second 7-chunk unit while the others idle.
template <int NCH>
__global__ __launch_bounds__(256, 1) void bf16_layer_units_kernel(const BfArgs a, const int l, const int dil,
                                                                  const int n_units, const int Fu, const int SP) {
    __shared__ __attribute__((aligned(16))) float cst[2 * 128];      // bd[128] | prescaled bx[128] of this layer
    __shared__ float wus[128];                                       // upsampler taps (U <= 112)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    cst[tid] = tid < 128 ? a.P[a.y.bd + (size_t)l * 128 + tid]
                         : a.P[a.y.bx + (size_t)l * 128 + tid - 128] * (tid < 192 ? K_SIG : K_TANH);
    if (tid < 128) wus[tid] = tid < a.U ? a.P[a.y.wup + tid] : 0.f;
    bf16x8 A[8][4];
    {
        const bf16x8* src = reinterpret_cast<const bf16x8*>(a.wbf + a.off_wd) + (size_t)l * 8 * 4 * 64;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) A[mt][ks] = src[(mt * 4 + ks) * 64 + lane];
    }
    __syncthreads();
    f32x4 kbd[8];
    float4 kbxz[4], kbxc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int ch = chan_of(m, g, 0);
        kbd[m] = *reinterpret_cast<const f32x4*>(cst + ch);
        kbd[4 + m] = *reinterpret_cast<const f32x4*>(cst + H + ch);
        kbxz[m] = *reinterpret_cast<const float4*>(cst + 128 + ch);
        kbxc[m] = *reinterpret_cast<const float4*>(cst + 128 + H + ch);
    }
    const size_t lstride = (size_t)a.B * a.Tp * H;
    const __amdgpu_buffer_rsrc_t rh = make_rsrc(a.hs + (size_t)l * lstride, lstride * 2);
    const __amdgpu_buffer_rsrc_t rn = make_rsrc(a.hs + (size_t)(l + 1) * lstride, lstride * 2);
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(a.cond, (size_t)a.B * a.Tf * a.N * 4);
    const unsigned lane_h = (unsigned)(n * H + 8 * g) * 2u, lane_c = (unsigned)(8 * g) * 4u;
    const unsigned dil_bytes = (unsigned)dil * H * 2u;

    const int Wn = gridDim.x * 4;
    // a wave walks units j0, j0+Wn, ...: at any time the chip streams one contiguous window of the layer
    // (measured 7 % faster than giving every wave its own contiguous range of units)
    const int js = Wn, j0 = blockIdx.x * 4 + w, j1 = n_units;
    if (j0 >= j1) return;

    auto unit_of = [&](int j) -> Unit {                              // scalar
        Unit u;
        const bool ok = j < j1;
        const int jc = ok ? j : j1 - 1;
        const int fs = jc / SP, part = jc - fs * SP;                  // frame index over all utterances, sub-unit
        u.b = fs / Fu; u.f = fs - u.b * Fu;
        const int s = u.f * a.U - a.coff, e = s + a.U;
        const int ps = s + part * (16 * NCH), pe = ps + 16 * NCH < e ? ps + 16 * NCH : e;
        u.ts = ps > 0 ? ps : 0;
        u.te = ok ? (pe < a.Tp ? pe : a.Tp) : 0;                      // a unit past the range is empty
        if (u.te < u.ts) u.te = u.ts;
        u.jj0 = u.ts - s;
        return u;
    };
    auto fetch_b = [&](const Unit& u, int i, BFrag& fr) {
        const int t0 = u.ts + 16 * i, t = t0 + n;
        const unsigned base = (unsigned)(u.b * a.Tp + t0) * (H * 2u) + lane_h;
        const bool ok = t < u.te;
        const unsigned o1 = ok ? base : OOB;
        const unsigned o0 = (ok && t >= dil) ? base - dil_bytes : OOB;
        fr.b[0] = buf_ld_bf8(rh, o0); fr.b[1] = buf_ld_bf8(rh, o0 + 64u);
        fr.b[2] = buf_ld_bf8(rh, o1); fr.b[3] = buf_ld_bf8(rh, o1 + 64u);
    };
    auto fetch_rows = [&](const Unit& u, CondRow& r) {
        const int fc = u.f < a.Tf - 1 ? u.f : a.Tf - 1;
        const unsigned off = (unsigned)((u.b * a.Tf + fc) * a.N + l * 128) * 4u + lane_c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {              // q: channel group (q>>1)*32 + 8g + 4*(q&1)
            r.cz[q] = buf_ld_f4(rc, off + (unsigned)(32 * (q >> 1) + 4 * (q & 1)) * 4u);
            r.cc[q] = buf_ld_f4(rc, off + (unsigned)(H + 32 * (q >> 1) + 4 * (q & 1)) * 4u);
        }
    };

    Unit cu = unit_of(j0), nu = unit_of(j0 + js);
    CondRow cur;
    fetch_rows(cu, cur);
    BFrag ring[NCH];
#pragma unroll
    for (int i = 0; i < NCH - 1; ++i) fetch_b(cu, i, ring[i]);
    f32x4 acc[2][8];
    mfma_first(acc[0], A, ring[0].b[0], kbd);
    mfma_next<1, false, true>(acc[0], A, ring[0].b[1]);
    mfma_next<2, false, false>(acc[0], A, ring[0].b[2]);
    mfma_next<3, true, false>(acc[0], A, ring[0].b[3]);

    // the body of one unit; the first unit is peeled (called once ahead of the loop) so that the loop header
    // sees the steady-state queue of outstanding loads and the compiler's vmcnt waits do not drain the ring
    auto unit_body = [&](const int j) __attribute__((always_inline)) {
        CondRow nxt;
        fetch_rows(nu, nxt);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            if (i == 0) fetch_b(cu, NCH - 1, ring[NCH - 1]);         // slot freed by the previous unit's last chunk
            else fetch_b(nu, i - 1, ring[i - 1]);                    // slot freed by this unit's chunk i-1
            const int t0 = cu.ts + 16 * i, t = t0 + n;
            const float wu = wus[cu.jj0 + 16 * i + n];
            const float wuz = wu * K_SIG, wuc = wu * K_TANH;
            f32x4 (&ac)[8] = acc[i & 1];
            f32x4 (&an)[8] = acc[(i + 1) & 1];
            const BFrag& cb = ring[i];
            const BFrag& nb = ring[(i + 1) % NCH];                   // i == NCH-1: slot 0 already holds (next unit, 0)
            unsigned hw[8];                                          // 16 finished channels, two bf16 per word
            auto epi = [&](const int m) __attribute__((always_inline)) {
                // two channels per instruction (v_pk_fma/mul/add_f32); exp2 / rcp stay per element
                const v2f cz2[2] = {{cur.cz[m].x, cur.cz[m].y}, {cur.cz[m].z, cur.cz[m].w}};
                const v2f cc2[2] = {{cur.cc[m].x, cur.cc[m].y}, {cur.cc[m].z, cur.cc[m].w}};
                const v2f bz2[2] = {{kbxz[m].x, kbxz[m].y}, {kbxz[m].z, kbxz[m].w}};
                const v2f bc2[2] = {{kbxc[m].x, kbxc[m].y}, {kbxc[m].z, kbxc[m].w}};
                const v2f wuz2 = {wuz, wuz}, wuc2 = {wuc, wuc}, one = {1.f, 1.f}, mtwo = {-2.f, -2.f};
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const v2f az = {ac[m][2 * p], ac[m][2 * p + 1]}, acd = {ac[4 + m][2 * p], ac[4 + m][2 * p + 1]};
                    const v2f pz = (wuz2 * cz2[p] + bz2[p]) * az, pc = (wuc2 * cc2[p] + bc2[p]) * acd;
                    const v2f dz = (v2f){__builtin_amdgcn_exp2f(pz.x), __builtin_amdgcn_exp2f(pz.y)} + one;
                    const v2f dc = (v2f){__builtin_amdgcn_exp2f(pc.x), __builtin_amdgcn_exp2f(pc.y)} + one;
                    const v2f z = {__builtin_amdgcn_rcpf(dz.x), __builtin_amdgcn_rcpf(dz.y)};
                    const v2f q = {__builtin_amdgcn_rcpf(dc.x), __builtin_amdgcn_rcpf(dc.y)};
                    const v2f cd = mtwo * q + one;
                    const unsigned hpw = __builtin_bit_cast(u32x4, cb.b[2 + (m >> 1)])[(m & 1) * 2 + p];   // own tap-1 fragment
                    const v2f hp = {__builtin_bit_cast(float, hpw << 16), __builtin_bit_cast(float, hpw & 0xffff0000u)};
                    const v2f o = z * (hp - cd) + cd;                                  // (1-z) c + z h
                    hw[m * 2 + p] = __builtin_bit_cast(unsigned, __builtin_convertvector(o, bf16x2));    // one v_cvt_pk_bf16_f32
                }
            };
            mfma_first(an, A, nb.b[0], kbd);                epi(0);
            mfma_next<1, false, true>(an, A, nb.b[1]);      epi(1);
            mfma_next<2, false, false>(an, A, nb.b[2]);     epi(2);
            mfma_next<3, true, false>(an, A, nb.b[3]);      epi(3);
            uint4 o0, o1;
            o0.x = hw[0]; o0.y = hw[1]; o0.z = hw[2]; o0.w = hw[3];
            o1.x = hw[4]; o1.y = hw[5]; o1.z = hw[6]; o1.w = hw[7];
            const unsigned so = t < cu.te ? (unsigned)(cu.b * a.Tp + t0) * (H * 2u) + lane_h : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rn, so, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rn, so + 64u, 0, 0);
        }
        if (NCH & 1) {                                               // odd chunk count: realign the accumulator parity
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) acc[0][mt] = acc[1][mt];
        }
        cur = nxt; cu = nu; nu = unit_of(j + 2 * js);
    };
    unit_body(j0);
    for (int j = j0 + js; j < j1; j += js) unit_body(j);
}

// ---- 4-tile MFMA helpers of the depth-fused kernel below (a wave owns the 64 rows of one channel half) --------------------
// EVERY operand in architectural VGPRs: a kernel that names no AccVGPR gets the whole register budget of its occupancy as
// VGPRs (with any "a" operand hipcc splits it half and half: 84 + 84 at three waves per SIMD).  K-step-major order as in
// mfma_first / mfma_next above; the tail waits for the MFMA -> VALU hazard.
#define SWN_MF "v_mfma_f32_16x16x32_bf16 "
__device__ __forceinline__ void mfma4v_first(f32x4 (&acc)[4], const bf16x8 (&A)[4][4], const bf16x8& b, const f32x4 (&c)[4]) {
    asm("s_nop 3\n\t" SWN_MF "%0, %4, %8, %9\n\t" SWN_MF "%1, %5, %8, %10\n\t" SWN_MF "%2, %6, %8, %11\n\t" SWN_MF "%3, %7, %8, %12"
        : "=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3])
        : "v"(A[0][0]), "v"(A[1][0]), "v"(A[2][0]), "v"(A[3][0]), "v"(b), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]));
}
template <int KS, bool TAIL>
__device__ __forceinline__ void mfma4v_next(f32x4 (&acc)[4], const bf16x8 (&A)[4][4], const bf16x8& b) {
#define SWN_MF4 SWN_MF "%0, %4, %8, %0\n\t" SWN_MF "%1, %5, %8, %1\n\t" SWN_MF "%2, %6, %8, %2\n\t" SWN_MF "%3, %7, %8, %3"
    if (TAIL) asm("s_nop 3\n\t" SWN_MF4 "\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                  : "v"(A[0][KS]), "v"(A[1][KS]), "v"(A[2][KS]), "v"(A[3][KS]), "v"(b));
    else asm("s_nop 3\n\t" SWN_MF4 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
             : "v"(A[0][KS]), "v"(A[1][KS]), "v"(A[2][KS]), "v"(A[3][KS]), "v"(b));
#undef SWN_MF4
}
#undef SWN_MF

// ---- depth-fused gated stack: all six layers of a position range in ONE launch ---------------------------------------------
// Per-layer launches move every hidden state through HBM twice (written by layer l, read by layer l+1) and pay a launch, a
// 32 KB weight fetch per wave and a drain per layer - at BASELINE cfg4's own 8 x 16 500 that fixed part is a third of a 16 us
// launch.  Here a 768-thread workgroup (one per CU) walks a CONTIGUOUS range of conditioning frames; wave (l, hh) owns layer l
// and the channels [32 hh, 32 hh + 32) (the M-tiles {2hh, 2hh+1} gate / {4+2hh, 4+2hh+1} candidate of the permuted A image:
// 64 resident registers), and the twelve waves step through the chunk sequence in lock-step, layer l one chunk behind layer l-1:
//   * h0 arrives by LDS-DMA (`buffer_load_dwordx4 ... lds`, 8 chunks ahead, each 1-KiB piece laid down in MFMA B-fragment
//     order: lane (n, g) fetches position n, channels 8g.. of its half - the piece IS the fragment image);
//   * a layer's output chunk-half is ONE ds_write_b128 per lane into the next level's 8-chunk LDS ring - the same fragment
//     image, because a lane finishes exactly the 16 bytes of its own tap-1 fragment - and one 16-byte store to HBM (the head
//     and the backward read every level); the next layer reads tap 1 lane-linearly and tap 0 (position t - dil) at a rotated
//     lane slot of an earlier chunk (conflict-free: a rotation inside 16-lane groups).  Hidden states are READ from HBM once
//     (h0) instead of six times;
//   * one workgroup barrier per step; LDS-DMA data is ordered for its readers by the issuer's counted vmcnt + that barrier
//     (cdna_hip_programming.md 5 "Pipelining across barriers"; the DMA statements are asm, so hipcc adds no waits of its own);
//   * a range starts a few chunks early (the stack reaches 63 positions back); halo chunks are computed, not stored.
// Chunks are frame-aligned (NCF = ceil(U / 16) per conditioning frame, the last one ragged), so the hoisted in_x row is
// uniform per chunk.  Frames 0 and 1 of an utterance (zero padding in front, frame 0 shorter by `coff`) change two scalars.
// Measured: the six layers 96 -> 77 us at 8 x 16 500, 451 -> 460 us at 64 x 16 500.  What was tried on top and
// measured no better: two half-steps per chunk with consecutive layers in opposite phases (+8 %), MFMAs of chunk q+1 between
// the epilogue quarters of chunk q inside a wave (+7..13 %: six more LDS reads per step), a branch-free step loop (+-0);
// removing the barrier changes nothing, removing the MFMAs or the epilogue a quarter each: a step is ~3 500 cycles of twelve
// waves' vector-instruction issue (146 VALU + 70 SALU per wave and step), not a wait.
constexpr int FZ_L0_SLOTS = 16, FZ_L0_AHEAD = 8, FZ_LV_SLOTS = 8, FZ_COND_SLOTS = 4, FZ_NL = 6;
constexpr int FZ_O_LV = FZ_L0_SLOTS * 2048;                                  // levels 1..5 behind level 0
constexpr int FZ_O_COND = FZ_O_LV + FZ_NL * FZ_LV_SLOTS * 2048;              // [layer][frame & 3][1 KB piece]
constexpr int FZ_O_CST = FZ_O_COND + FZ_NL * FZ_COND_SLOTS * 1024;           // [layer][bd 128 | prescaled bx 128] floats
constexpr int FZ_O_WUS = FZ_O_CST + FZ_NL * 256 * 4;                         // upsampler taps, 128 floats
constexpr int FZ_O_ZERO = FZ_O_WUS + 128 * 4;                                // 16 zero bytes (tap-0 reads before t = 0)
constexpr int FZ_LDS_BYTES = FZ_O_ZERO + 16;
typedef int v4i __attribute__((ext_vector_type(4)));

// one 1-KiB LDS-DMA piece: lane i's 16 bytes land at lds_base + 16 i; voff (per lane; the range check is on it) + soff =
// byte offset into the buffer, out of range: zeros.  M0 carries the LDS base and is written in the statement that reads it.
__device__ __forceinline__ void dma_piece(const v4i rsrc, unsigned voff, unsigned soff, unsigned lds_base) {
    // (the two scalar operands are wave-uniform by construction; readfirstlane makes that visible to the register allocator)
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane((int)soff)),
                    "s"(__builtin_amdgcn_readfirstlane((int)lds_base)) : "memory");
}
__device__ __forceinline__ v4i raw_rsrc(const void* p, size_t bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)(a & 0xffffffffull));
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffull));
    r.z = __builtin_amdgcn_readfirstlane((int)(unsigned)bytes);
    r.w = 0x00020000;
    return r;
}

struct FzPos { int j, c, b, f; };                  // wave-uniform walker: global frame, chunk in frame, utterance, frame in utterance

__global__ __launch_bounds__(768) void bf16_stack_fused_kernel(const BfArgs a, const int n_frames_all, const int Fu) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];      // ALL LDS of the kernel is this one block
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = w >> 1, hh = w & 1;                                // layer, channel half
    const int n = lane & 15, g = lane >> 4;
    const int U = a.U, coff = a.coff, NCF = (U + 15) >> 4, dil = 1 << l;     // BL6 class: K = 2, dilations 1 .. 32
    float* cst = reinterpret_cast<float*>(lds + FZ_O_CST);
    for (int e = tid; e < FZ_NL * 256; e += 768) {
        const int ll = e >> 8, r = e & 255;
        cst[e] = r < 128 ? a.P[a.y.bd + (size_t)ll * 128 + r] : a.P[a.y.bx + (size_t)ll * 128 + r - 128] * (r < 192 ? K_SIG : K_TANH);
    }
    if (tid < 128) reinterpret_cast<float*>(lds + FZ_O_WUS)[tid] = tid < U ? a.P[a.y.wup + tid] : 0.f;
    if (tid < 4) reinterpret_cast<unsigned*>(lds + FZ_O_ZERO)[tid] = 0u;
    // the rings start as zeros: the first chunks read tap-0 slots no producer has written (nothing real depends on them)
    for (int e = tid; e < FZ_O_CST / 16; e += 768) reinterpret_cast<uint4*>(lds)[e] = make_uint4(0u, 0u, 0u, 0u);
    bf16x8 A[4][4];
    {
        const bf16x8* src = reinterpret_cast<const bf16x8*>(a.wbf + a.off_wd) + (size_t)l * 8 * 4 * 64;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int mt = (t >> 1) * 4 + 2 * hh + (t & 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kk = ks < 2 ? ks : (ks == 2 ? 2 + hh : 3 - hh);       // k-steps: tap0 lo, tap0 hi, tap1 own, tap1 other
                A[t][ks] = src[(mt * 4 + kk) * 64 + lane];
            }
        }
    }
    __syncthreads();
    const unsigned lc = (unsigned)(32 * hh + 8 * g) * 4u;            // this lane's 8 channels inside a 128-float row
    f32x4 kbd[4];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        kbd[m] = *reinterpret_cast<const f32x4*>(lds + FZ_O_CST + l * 1024 + lc + 16 * m);
        kbd[2 + m] = *reinterpret_cast<const f32x4*>(lds + FZ_O_CST + l * 1024 + H * 4 + lc + 16 * m);
    }
    const size_t lstride = (size_t)a.B * a.Tp * H;
    const v4i rh0 = raw_rsrc(a.hs, lstride * 2);
    const v4i rcd = raw_rsrc(a.cond, (size_t)a.B * a.Tf * a.N * 4);
    const __amdgpu_buffer_rsrc_t rn = make_rsrc(a.hs + (size_t)(l + 1) * lstride, lstride * 2);
    const unsigned lane_row = (unsigned)(n * H + 8 * g) * 2u + 64u * (unsigned)hh;     // own 16 bytes of position n's row
    const unsigned lane_lds = (unsigned)lane * 16u;
    const unsigned cond_voff = lane < 32 ? (unsigned)lane * 16u : OOB;
    const unsigned in_base = l == 0 ? 0u : (unsigned)(FZ_O_LV + (l - 1) * FZ_LV_SLOTS * 2048);   // ring of this layer's input level
    const unsigned in_mask = l == 0 ? FZ_L0_SLOTS - 1 : FZ_LV_SLOTS - 1;
    const unsigned out_base = (unsigned)(FZ_O_LV + l * FZ_LV_SLOTS * 2048);                       // (layer 5: a spare ring)

    // this workgroup's contiguous range of frames, entered `halo` chunks early
    const int G = gridDim.x, k = blockIdx.x;
    const int jb = (int)((long)n_frames_all * k / G), je = (int)((long)n_frames_all * (k + 1) / G);
    if (jb >= je) return;
    // halo: the chunks in front of frame jb that hold the 63 positions the six layers reach back (the last chunk of a frame
    // may hold as little as one position)
    int halo = 0;
    for (int pos = 0, c = NCF - 1; pos < 64; ++halo) { pos += (U - 16 * c) < 16 ? (U - 16 * c) : 16; c = c == 0 ? NCF - 1 : c - 1; }
    const int hf = (halo + NCF - 1) / NCF;                           // frames the halo reaches back
    const int NQ = halo + (je - jb) * NCF;                           // chunks every layer walks; chunk number 0 is (jb - hf, hf NCF - halo)
    auto pos_at = [&](int q) -> FzPos {                              // scalar, with divisions: seeds a walker
        FzPos p;
        const int qq = q + hf * NCF - halo;                          // chunks since the start of frame jb - hf
        p.j = jb - hf + qq / NCF; p.c = qq - (qq / NCF) * NCF;
        const int jc = p.j < 0 ? 0 : p.j;
        p.b = jc / Fu; p.f = jc - p.b * Fu;
        return p;
    };
    auto advance = [&](FzPos& p) {
        if (++p.c == NCF) { p.c = 0; ++p.j; if (p.j > 0 && ++p.f == Fu) { p.f = 0; ++p.b; } }
    };
    auto first_pos = [&](const FzPos& p) -> int { const int s = p.f * U - coff; return s > 0 ? s : 0; };
    auto real = [&](const FzPos& p) -> bool { return p.j >= 0 && p.j < n_frames_all; };
    // level-0 piece of chunk q (waves of layer 0: half hh each); lanes past the frame's end fetch positions nobody uses
    auto issue_h0 = [&](const FzPos& p, int q) {
        const bool ok = real(p) && q < NQ;
        dma_piece(rh0, ok ? lane_row : OOB, ok ? (unsigned)(p.b * a.Tp + first_pos(p) + 16 * p.c) * (H * 2u) : 0u,
                  ((unsigned)q & (FZ_L0_SLOTS - 1)) * 2048u + hh * 1024u);
    };
    // conditioning row of this layer for frame (j, b, f) (waves hh == 0)
    auto issue_cond = [&](int j, int b, int f) {
        const bool ok = j >= 0 && j < n_frames_all;
        const int fc = f < a.Tf - 1 ? f : a.Tf - 1;
        dma_piece(rcd, ok ? cond_voff : OOB, ok ? (unsigned)((b * a.Tf + fc) * a.N + l * 128) * 4u : 0u,
                  (unsigned)(FZ_O_COND + (l * FZ_COND_SLOTS + (j & (FZ_COND_SLOTS - 1))) * 1024));
    };

    FzPos cp = pos_at(0);                                            // the chunk this wave processes next
    FzPos dp = cp;                                                   // layer-0 waves: the chunk whose piece is issued next
    if (l == 0) {
        for (int q = 0; q < FZ_L0_AHEAD; ++q) { issue_h0(dp, q); advance(dp); }
    }
    if (hh == 0) {                                                   // rows of the first three frames: the loop issues frame j + 2
        issue_cond(cp.j, cp.b, cp.f);                                // when it enters frame j, and it may enter its first frame mid-way
        FzPos np = cp; np.c = NCF - 1; advance(np);                  // first chunk of the next frame
        issue_cond(np.j, np.b, np.f);
        np.c = NCF - 1; advance(np);
        issue_cond(np.j, np.b, np.f);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // The step loop is kept free of scalar branches (a taken s_cbranch costs a wave more than the four MFMAs of a k-step):
    // every wave runs the whole body in every step - before its first and after its last chunk on whatever the rings hold,
    // with the stores switched off (nothing real reads what such a step writes: the slot it writes is rewritten by the real
    // producer before its reader arrives, or its readers are done) - and everything that depends on the frame only is a
    // scalar refreshed when the walker enters a frame.
    int fp = 0, fe = 0, uprev = U, gpos = 0, cslot = 0;              // first / end position in the utterance, previous frame's image length, b*Tp + fp
    bool zero_front = false, st_frame = false;
    auto enter_frame = [&]() {
        fp = cp.f * U - coff; fp = fp > 0 ? fp : 0;
        fe = (cp.f + 1) * U - coff; fe = fe < a.Tp ? fe : a.Tp;
        uprev = cp.f == 1 ? U - coff : U;                            // frame 0's image starts at frame offset coff
        zero_front = cp.f == 0;                                      // zeros in front of the utterance
        gpos = cp.b * a.Tp + fp;
        cslot = cp.j & (FZ_COND_SLOTS - 1);
        st_frame = cp.j >= jb && cp.j >= 0 && cp.j < n_frames_all;   // halo chunks are computed, not stored
    };
    enter_frame();
    const unsigned voob = OOB;
    const int S = NQ + FZ_NL - 1;
    for (int s = 0; s < S; ++s) {
        const int q = s - l;                                         // this wave's chunk number in this step
        const bool live = q >= 0 && q < NQ;
        if (l == 0) { issue_h0(dp, s + FZ_L0_AHEAD); advance(dp); }  // always one piece per step: the vmcnt below counts on it
        if (hh == 0 && cp.c == 0 && live) {                          // two frames ahead: >= 2 NCF - 1 >= 5 younger operations by the
            FzPos np = cp; np.c = NCF - 1; advance(np);              // time it is read, more than any wait below leaves
            np.c = NCF - 1; advance(np);
            issue_cond(np.j, np.b, np.f);
        }
        // ---- fragments of chunk q out of the input level's ring
        const unsigned sb = in_base + ((unsigned)q & in_mask) * 2048u;
        bf16x8 fb[4];
        fb[2] = *reinterpret_cast<const bf16x8*>(lds + sb + hh * 1024 + lane_lds);
        fb[3] = *reinterpret_cast<const bf16x8*>(lds + sb + (1 - hh) * 1024 + lane_lds);
        const int r0 = 16 * cp.c + n - dil;                          // offset in the frame image of the tap-0 position
        const bool neg = r0 < 0;
        const int r = neg ? r0 + uprev : r0;                         // ... in the previous frame's image
        const int qs = q - cp.c + (r >> 4) - (neg ? NCF : 0);        // its chunk number
        const bool zf = neg && zero_front;
        const unsigned a0 = zf ? (unsigned)FZ_O_ZERO
                               : in_base + ((unsigned)qs & in_mask) * 2048u + (unsigned)(r & 15) * 16u + (unsigned)(16 * g) * 16u;
        fb[0] = *reinterpret_cast<const bf16x8*>(lds + a0);
        fb[1] = *reinterpret_cast<const bf16x8*>(lds + (zf ? a0 : a0 + 1024u));
        // ---- epilogue operands
        const unsigned char* cr = lds + FZ_O_COND + (l * FZ_COND_SLOTS + cslot) * 1024 + lc;
        const unsigned char* kx = lds + FZ_O_CST + l * 1024 + 128 * 4 + lc;
        float4 czq[2], ccq[2], bzq[2], bcq[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            czq[m] = *reinterpret_cast<const float4*>(cr + 16 * m); ccq[m] = *reinterpret_cast<const float4*>(cr + H * 4 + 16 * m);
            bzq[m] = *reinterpret_cast<const float4*>(kx + 16 * m); bcq[m] = *reinterpret_cast<const float4*>(kx + H * 4 + 16 * m);
        }
        int jo = 16 * cp.c + n + (zero_front ? coff : 0); jo = jo < 127 ? jo : 127;
        const float wu = reinterpret_cast<const float*>(lds + FZ_O_WUS)[jo];
        const float wz = wu * K_SIG, wc = wu * K_TANH;
        // ---- D[64 rows of this half][16 pos] = bd + Wd . [h(t-dil) ; h(t)]
        f32x4 ac[4];
        mfma4v_first(ac, A, fb[0], kbd);
        mfma4v_next<1, false>(ac, A, fb[1]);
        mfma4v_next<2, false>(ac, A, fb[2]);
        mfma4v_next<3, true>(ac, A, fb[3]);
        const u32x4 hpq = __builtin_bit_cast(u32x4, fb[2]);          // own tap-1 fragment: the highway input of these channels
        unsigned hw[4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const v2f cz2 = p ? (v2f){czq[m].z, czq[m].w} : (v2f){czq[m].x, czq[m].y};
                const v2f cc2 = p ? (v2f){ccq[m].z, ccq[m].w} : (v2f){ccq[m].x, ccq[m].y};
                const v2f bz2 = p ? (v2f){bzq[m].z, bzq[m].w} : (v2f){bzq[m].x, bzq[m].y};
                const v2f bc2 = p ? (v2f){bcq[m].z, bcq[m].w} : (v2f){bcq[m].x, bcq[m].y};
                const v2f wuz2 = {wz, wz}, wuc2 = {wc, wc}, one = {1.f, 1.f}, mtwo = {-2.f, -2.f};
                const v2f az = {ac[m][2 * p], ac[m][2 * p + 1]}, acd = {ac[2 + m][2 * p], ac[2 + m][2 * p + 1]};
                const v2f pz = (wuz2 * cz2 + bz2) * az, pc = (wuc2 * cc2 + bc2) * acd;
                const v2f dz = (v2f){__builtin_amdgcn_exp2f(pz.x), __builtin_amdgcn_exp2f(pz.y)} + one;
                const v2f dc = (v2f){__builtin_amdgcn_exp2f(pc.x), __builtin_amdgcn_exp2f(pc.y)} + one;
                const v2f z = {__builtin_amdgcn_rcpf(dz.x), __builtin_amdgcn_rcpf(dz.y)};
                const v2f qq = {__builtin_amdgcn_rcpf(dc.x), __builtin_amdgcn_rcpf(dc.y)};
                const v2f cd = mtwo * qq + one;
                const unsigned hpw = hpq[m * 2 + p];
                const v2f hp = {__builtin_bit_cast(float, hpw << 16), __builtin_bit_cast(float, hpw & 0xffff0000u)};
                const v2f o = z * (hp - cd) + cd;                                      // (1-z) c + z h
                hw[m * 2 + p] = __builtin_bit_cast(unsigned, __builtin_convertvector(o, bf16x2));
            }
        uint4 o0;
        o0.x = hw[0]; o0.y = hw[1]; o0.z = hw[2]; o0.w = hw[3];
        // next level's ring (the fragment image of chunk q; the last layer's writes land in a spare ring nobody reads) and HBM
        *reinterpret_cast<uint4*>(lds + out_base + ((unsigned)q & (FZ_LV_SLOTS - 1)) * 2048u + hh * 1024u + lane_lds) = o0;
        const int t = fp + 16 * cp.c + n;                            // this lane's position in its utterance
        const bool st = st_frame && live;
        const unsigned vo = (st && t < fe) ? lane_row : voob;        // an out-of-range store is dropped, but vmcnt counts it
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rn, vo, st ? (unsigned)(gpos + 16 * cp.c) * (H * 2u) : 0u, 0);
        if (live) {
            const int jprev = cp.j;
            advance(cp);
            if (cp.j != jprev) enter_frame();
        }
        // what a wave issued two (layer 0: six) operations ago has landed / left; then publish this step's LDS writes
        if (l == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");                               // no LDS access of the next step may move above the barrier
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // pieces past the range may still be landing
}

// ---- head: skip (K = L*64) -> relu -> out_1 (128x128) -> relu -> out_2 (<=16 x 128) ----------------
template <int LL>
__global__ __launch_bounds__(256, 2) void bf16_head_kernel(const BfArgs a, const int n_tiles) {
    constexpr int ROW2 = 256 + 16;               // [pos][128 ch] bf16 tile pitch
    __shared__ __attribute__((aligned(16))) unsigned char t1[TN * ROW2];
    // TWO workgroups per CU (65 KB of LDS and <= 256 registers each): the three GEMM stages of a tile are separated by
    // barriers and each is a short latency chain (LDS reads -> MFMAs -> epilogue -> LDS writes), so a second workgroup's
    // stages fill the gaps.  The out_1 tile t2 lives in the first 17 KB of the fragment buffer `bt`, which is dead once
    // the skip GEMM has consumed it (barrier #2) and is only re-published after barrier #4.
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
    __amdgpu_buffer_rsrc_t rl[LL];                                   // hidden states of layers 1..L
#pragma unroll
    for (int q = 0; q < LL; ++q) rl[q] = make_rsrc(a.hs + (size_t)(1 + q) * lstride, lstride * 2);

    // One workgroup per CU.  Every wave needs all 48 B fragments of a 64-position tile (its 32 skip rows times all
    // positions), so letting each wave load them made the tile cross the L2->L1 path four times (measured: the
    // kernel ran at the speed of its skip-GEMM loads alone, 2.4 TB/s of HBM but 9.5 TB/s of L2 traffic).  Now wave w
    // fetches only column w (its 16 positions x 6 layers, 12 fragments, one tile AHEAD in registers) with branch-free
    // buffer loads (a position past the end reads zeros), publishes it in LDS in fragment order, and all four
    // waves read the whole tile from LDS: the tile leaves L2 once.  (One buffer is enough: a tile is published
    // after the previous tile's last barrier, when nobody reads the buffer any more.)
    extern __shared__ __attribute__((aligned(16))) unsigned char bt[];       // [4 nt][KS1][64 lanes][16 B] = 48 KB
    unsigned char* t2 = bt;                                                  // aliases bt (see above)
    static_assert(TN * ROW2 <= 4 * KS1 * 64 * 16, "t2 must fit inside bt");
    bf16x8 mycol[KS1];
    auto fetch_col = [&](int tix) {
        const int tc = tix < n_tiles ? tix : n_tiles - 1;
        const int b = tc / tiles_per_b, t = (tc - b * tiles_per_b) * TN + 16 * w + n;
        const unsigned off = (tix < n_tiles && t < a.Tp) ? (unsigned)(b * a.Tp + t) * (H * 2u) + (unsigned)(8 * g) * 2u : OOB;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) mycol[ks] = buf_ld_bf8(rl[ks / 2], off + 64u * (ks & 1));
    };
    fetch_col(blockIdx.x);

    for (int tix = blockIdx.x; tix < n_tiles; tix += gridDim.x) {
        const int b = tix / tiles_per_b, t0 = (tix - b * tiles_per_b) * TN;
        unsigned char* btc = bt;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
            *reinterpret_cast<bf16x8*>(btc + ((w * KS1 + ks) * 64 + lane) * 16) = mycol[ks];
        fetch_col(tix + gridDim.x);                     // next tile's column flies under this tile's three GEMMs
        __syncthreads();
        // ---- skip = Wsk . [h_1 .. h_L]
        f32x4 acc[2][4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            acc[0][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[1][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(btc + ((nt * KS1 + ks) * 64 + lane) * 16);
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

// ---- dropout mode (aux_drop at sample rate) ---------------------------------------------------------------------------
// xm16[b][u][c] = bf16(drop_x[b][c][u] * (C[b][c][f] * w_up[j] + b_up)),  u + coff = f*U + j   (cswnv_shift1.py:193-195):
// the masked conditioning as time-major rows, the B operand of the in_x GEMM.  Workgroup = 64 positions x 64 channels: the
// mask is read along its contiguous axis (positions), transposed through LDS and leaves as 128-byte row segments.
__global__ __launch_bounds__(256) void xm16_kernel(const float* __restrict__ C, const float* __restrict__ P, size_t wup, size_t bup,
                                                   const float* __restrict__ drop_x, unsigned short* __restrict__ xm,
                                                   int A0, int A0x, int Tf, int U, int coff, int Tx) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64 * 72];       // [position][64 channels + 8 pad]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int u0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
    const int u = u0 + lane;
    if (u < Tx) {
        const int tt = u + coff, f = tt / U, j = tt - f * U;
        const float wu = P[wup + j], bu = P[bup];
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int c = c0 + 16 * w + i;
            float v = 0.f;
            if (c < A0) v = drop_x[((size_t)b * A0 + c) * Tx + u] * fmaf(C[((size_t)b * A0 + c) * Tf + f], wu, bu);
            tile[lane * 72 + 16 * w + i] = f2bf(v);
        }
    }
    __syncthreads();
    const int r = tid >> 2, q = tid & 3;                           // position, 16-channel quarter
    if (u0 + r < Tx && c0 + 16 * q < A0x) {
        const uint4 v0 = *reinterpret_cast<const uint4*>(tile + r * 72 + 16 * q);
        const uint4 v1 = *reinterpret_cast<const uint4*>(tile + r * 72 + 16 * q + 8);
        uint4* dst = reinterpret_cast<uint4*>(xm + ((size_t)b * Tx + u0 + r) * A0x + c0 + 16 * q);
        dst[0] = v0; dst[1] = v1;
    }
}
// fp32 [rows][ld] (first `cols` of each row) -> bf16 [rows][dcols], zero columns cols..dcols
__global__ void rows_to_bf16_kernel(const float* __restrict__ src, int ld, int rows, int cols, int dcols, unsigned short* __restrict__ dst) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)rows * dcols) return;
    const int r = (int)(e / dcols), c = (int)(e - (size_t)r * dcols);
    dst[e] = c < cols ? f2bf(src[(size_t)r * ld + c]) : (unsigned short)0;
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

// large geometries (H a multiple of 64 beyond the BL6 class): tiled bf16 GEMM stack, csrc/swn_stack_bf16g.hip
int swn_bf16g_geom(const swn_net_desc* d, SwnGeom* g);
size_t swn_bf16g_weight_bytes(const SwnGeom& g);
int swn_bf16g_pack(const SwnGeom& g, const float* packed, void* wbf, hipStream_t st);
size_t swn_bf16g_work_bytes(const SwnGeom& g, int batch, long Tp);
int swn_bf16g_expand(const SwnGeom& g, const void* work, int batch, long Tp, float* fwd_work, bool hs_only, hipStream_t st);
int swn_train_head_acts(const SwnGeom& g, const float* packed, float* work, int batch, long Tp, hipStream_t st);
int swn_bf16g_forward(const SwnGeom& g, const float* packed, const void* wbf, const float* cond, const void* audio,
                      int batch, int n_frames, void* work, float* out, hipStream_t st, float* a_keep = nullptr,
                      const float* gx = nullptr, const float* const* drop_h = nullptr, unsigned short* hm16 = nullptr);
size_t swn_bf16g_keep_floats(const SwnGeom& g, int batch, long Tp);

int swn_bf16g_plain(const unsigned short* A, int M, const unsigned short* src, size_t blk_stride, size_t src_bytes, int KB, int nblk,
                    int Tp, int B, const float* bias, unsigned short* out_bf, int out_ld, float* out_f, int NO, hipStream_t st);
bool swn_bl6_bwd_supported(const SwnGeom& g, int B, long Tp, int n_frames);                    // csrc/swn_bwd_bl6.hip

namespace {
size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
void pack_bl6_images(const SwnGeom& g, const SwnLayout& y, const float* packed, unsigned short* wbf, hipStream_t st) {
    const BfOffsets o = bf_offsets(g);
    hipLaunchKernelGGL(pack_wd_kernel, dim3((g.L * 8 * 4 * 512 + 255) / 256), dim3(256), 0, st, packed + y.wd, g.L, wbf + o.wd);
    hipLaunchKernelGGL(pack_frag_kernel, dim3((8 * g.L * 2 * 512 + 255) / 256), dim3(256), 0, st,
                       packed + y.wsk, g.L * 64, 128, g.L * 64, 8, g.L * 2, wbf + o.wsk);
    hipLaunchKernelGGL(pack_frag_kernel, dim3((8 * 4 * 512 + 255) / 256), dim3(256), 0, st, packed + y.w1, g.Sp, 128, 128, 8, 4, wbf + o.w1);
    hipLaunchKernelGGL(pack_frag_kernel, dim3((4 * 512 + 255) / 256), dim3(256), 0, st, packed + y.w2, g.O1p, g.NO, 128, 1, 4, wbf + o.w2);
}
}  // namespace

// ---- dropout mode of the BL6 class in the mixed-precision mode (SwnBl6DropLayout, csrc/swn_geom.hpp) -------------------------
// With dilation_repeat == 1 the reference's hidden-state dropout lands on the last layer's output only
// (cswnv_shift1.py:211-217: (l + 1) % dilation_depth == 0 <=> l = L - 1), which nothing reads: the step is the plain one with
// the hoisted conditioning replaced by sample-rate in_x products of the masked conditioning.
bool swn_bl6_drop_supported(const SwnGeom& g, int B, long Tp, int n_frames, const float* const* drop_h) {
    if (!g.bl6 || g.kind != SWN_KIND_LAPLACE || g.S != 128 || g.NO > 16 || g.L != 6 || !swn_bl6_bwd_supported(g, B, Tp, n_frames)) return false;
    if (!drop_h) return false;
    for (int l = 0; l + 1 < g.L; ++l) if (drop_h[l]) return false;           // a mask between layers: the generic chain
    const size_t npos = (size_t)B * Tp;                                       // seg == 1: Tx == Tp
    return npos * g.L * 256 < (1ull << 31) && npos * swn_a0x(&g) * 2 < (1ull << 31);
}
SwnBl6DropLayout swn_bl6_drop_layout(const SwnGeom& g, int B, long Tp) {
    SwnBl6DropLayout o;
    const size_t npos = (size_t)B * Tp;
    o.hs16 = 0;
    o.wbf = o.hs16 + al256((size_t)(g.L + 1) * npos * H * 2);
    o.wx16 = o.wbf + al256(bf_offsets(g).total * 2);
    o.xm16 = o.wx16 + al256((size_t)g.L * 128 * swn_a0x(&g) * 2);
    o.gx16 = o.xm16 + al256(npos * swn_a0x(&g) * 2);
    o.total = o.gx16 + al256(npos * g.L * 256);
    return o;
}
// C: the last conv_aux activation (B, A0, Tf) inside swn_frontend's work buffer; work: SwnBl6DropLayout.total bytes
int swn_bl6_drop_forward(const SwnGeom& g, const float* packed, const float* C, const float* audio, const float* drop_x,
                         int batch, int n_frames, void* work, float* out, hipStream_t st) {
    const long Tp = (long)n_frames * g.U - 2 * g.seg + 1;
    const int Tx = (int)Tp, A0x = swn_a0x(&g);
    const SwnBl6DropLayout lo = swn_bl6_drop_layout(g, batch, Tp);
    unsigned char* wb = reinterpret_cast<unsigned char*>(work);
    unsigned short* wx16 = reinterpret_cast<unsigned short*>(wb + lo.wx16);
    unsigned short* xm16 = reinterpret_cast<unsigned short*>(wb + lo.xm16);
    unsigned short* gx16 = reinterpret_cast<unsigned short*>(wb + lo.gx16);
    BfArgs a;
    swn_make_layout(&g, &a.y);
    const BfOffsets o = bf_offsets(g);
    a.P = packed; a.wbf = reinterpret_cast<const unsigned short*>(wb + lo.wbf); a.cond = nullptr; a.audio = audio;
    a.hs = reinterpret_cast<unsigned short*>(wb + lo.hs16); a.out = out;
    a.B = batch; a.Tf = n_frames; a.Tp = (int)Tp; a.U = g.U; a.N = g.N; a.L = g.L; a.seg = g.seg; a.NO = g.NO; a.coff = g.seg;
    a.off_wd = o.wd; a.off_wsk = o.wsk; a.off_w1 = o.w1; a.off_w2 = o.w2; a.gx16 = gx16;
    (void)hipGetLastError();
    pack_bl6_images(g, a.y, packed, reinterpret_cast<unsigned short*>(wb + lo.wbf), st);
    hipLaunchKernelGGL(rows_to_bf16_kernel, dim3((unsigned)(((size_t)g.L * 128 * A0x + 255) / 256)), dim3(256), 0, st,
                       packed + a.y.wx, g.A0p, g.L * 128, g.A0, A0x, wx16);
    hipLaunchKernelGGL(xm16_kernel, dim3((Tx + 63) / 64, A0x / 64 + (A0x % 64 ? 1 : 0), batch), dim3(256), 0, st, C, packed, a.y.wup, a.y.bup,
                       drop_x, xm16, g.A0, A0x, n_frames, g.U, a.coff, Tx);
    // gx16[p][l*128 + o] = b_inx[l][o] + sum_c in_x[l].W[o][c] xm[p][c]
    int rc = swn_bf16g_plain(wx16, g.L * 128, xm16, 0, (size_t)batch * Tx * A0x * 2, A0x, 1, (int)Tp, batch, packed + a.y.bxr,
                             gx16, g.L * 128, nullptr, 0, st);
    if (rc < 0) return rc;
    hipLaunchKernelGGL(bf16_input_kernel, dim3((unsigned)((Tp + 255) / 256), batch), dim3(256), 0, st, a);
    const int n_chunks = batch * (int)((Tp + 15) / 16);
    const int grid = (n_chunks + 3) / 4 < 256 ? (n_chunks + 3) / 4 : 256;
    for (int l = 0; l < g.L; ++l)
        hipLaunchKernelGGL(bf16_layer_kernel<2>, dim3(grid), dim3(256), 0, st, a, l, g.dil[l], n_chunks);
    const int n_tiles = batch * (int)((Tp + TN - 1) / TN);
    hipLaunchKernelGGL(bf16_head_kernel<6>, dim3(n_tiles < 512 ? n_tiles : 512), dim3(256), 4 * 12 * 64 * 16, st, a, n_tiles);
    return swn_launch_status("swn_forward_drop");
}

extern "C" size_t swn_bf16_weight_bytes(const swn_net_desc* d) {
    SwnGeom g;
    if (bf_geom(d, &g) < 0) return swn_bf16g_geom(d, &g) < 0 ? 0 : swn_bf16g_weight_bytes(g);
    return bf_offsets(g).total * sizeof(unsigned short);
}

extern "C" int swn_pack_bf16(const swn_net_desc* d, const float* packed, void* wbf_, void* stream_) {
    SwnGeom g; int rc = bf_geom(d, &g);
    if (rc == SWN_E_UNSUPPORTED && swn_bf16g_geom(d, &g) == SWN_OK) {
        if (!packed || !wbf_) return SWN_E_BADARG;
        return swn_bf16g_pack(g, packed, wbf_, (hipStream_t)stream_);
    }
    if (rc < 0) return rc;
    if (!packed || !wbf_) return SWN_E_BADARG;
    SwnLayout y; swn_make_layout(&g, &y);
    unsigned short* wbf = reinterpret_cast<unsigned short*>(wbf_);
    hipStream_t st = (hipStream_t)stream_;
    (void)hipGetLastError();
    pack_bl6_images(g, y, packed, wbf, st);
    return swn_launch_status("swn_pack_bf16");
}

extern "C" size_t swn_forward_bf16_work_bytes(const swn_net_desc* d, int batch, int n_frames) {
    SwnGeom g;
    if (batch < 1 || n_frames < 1) return 0;
    if (bf_geom(d, &g) < 0) {
        if (swn_bf16g_geom(d, &g) < 0) return 0;
        const long Tpg = (long)n_frames * g.U - 2 * g.seg + 1;
        return Tpg < 1 ? 0 : swn_bf16g_work_bytes(g, batch, Tpg);
    }
    const long Tp = (long)n_frames * g.U - 2 * g.seg + 1;
    if (Tp < 1) return 0;
    return (size_t)(g.L + 1) * batch * Tp * H * sizeof(unsigned short);
}

extern "C" int swn_forward_bf16(const swn_net_desc* d, const float* packed, const void* wbf, const float* cond,
                                const void* audio_, int batch, int n_frames, void* work, float* out, void* stream_) {
    const float* audio = reinterpret_cast<const float*>(audio_);      // BL6 class: Laplace, float waveform
    SwnGeom g; int rc = bf_geom(d, &g);
    if (rc == SWN_E_UNSUPPORTED && swn_bf16g_geom(d, &g) == SWN_OK) {
        if (!packed || !wbf || !cond || !audio || !work || !out || batch < 1 || batch > 65535 || n_frames < 1) return SWN_E_BADARG;
        if ((long)n_frames * g.U - 2 * g.seg + 1 < 1) return SWN_E_BADARG;
        return swn_bf16g_forward(g, packed, wbf, cond, audio_, batch, n_frames, work, out, (hipStream_t)stream_);
    }
    if (rc < 0) return rc;
    if (!packed || !wbf || !cond || !audio || !work || !out || batch < 1 || batch > 65535 || n_frames < 1) return SWN_E_BADARG;
    const long Tp = (long)n_frames * g.U - 2 * g.seg + 1;
    if (Tp < 1) return SWN_E_BADARG;
    // 32-bit buffer offsets in the layer kernel: one layer of hidden states and the conditioning must stay < 2 GiB
    if ((size_t)batch * Tp * H * 2 >= (1ull << 31) || (size_t)batch * n_frames * g.N * 4 >= (1ull << 31)) return SWN_E_UNSUPPORTED;
    BfArgs a;
    swn_make_layout(&g, &a.y);
    const BfOffsets o = bf_offsets(g);
    a.P = packed; a.wbf = reinterpret_cast<const unsigned short*>(wbf); a.cond = cond; a.audio = audio;
    a.hs = reinterpret_cast<unsigned short*>(work); a.out = out;
    a.B = batch; a.Tf = n_frames; a.Tp = (int)Tp; a.U = g.U; a.N = g.N; a.L = g.L; a.seg = g.seg; a.NO = g.NO;
    a.coff = g.seg;
    a.off_wd = o.wd; a.off_wsk = o.wsk; a.off_w1 = o.w1; a.off_w2 = o.w2; a.gx16 = nullptr;
    hipStream_t st = (hipStream_t)stream_;
    (void)hipGetLastError();
    hipLaunchKernelGGL(bf16_input_kernel, dim3((unsigned)((Tp + 255) / 256), batch), dim3(256), 0, st, a);
    const int n_tiles = batch * (int)((Tp + TN - 1) / TN);
    const int n_chunks = batch * (int)((Tp + 15) / 16);
    const int grid = (n_chunks + 3) / 4 < 256 ? (n_chunks + 3) / 4 : 256;      // persistent: one workgroup (4 waves x 512 VGPRs) per CU
    const int nch = (g.U + 15) / 16;
#if !defined(SWN_OLD_UNITS)           // diagnostic builds only: the per-layer launches for A/B timing
    // all six gated layers in one launch (bf16_stack_fused_kernel): one workgroup per CU walks a contiguous range of frames.
    // Up to ~12 frames per workgroup (cfg4's own 8 x 150: 4.7) it beats six per-layer launches (4 x 150: 0.091 / 0.104 ms,
    // 8 x 150: 0.122 / 0.141, 16 x 150: 0.198 / 0.221); from 32 x 150 on the two tie (0.369 / 0.363, 64 x 150: 0.713 / 0.707) and
    // the per-layer kernels, which need no halo, keep the large sizes
    if (g.seg == 1 && g.U >= 33 && nch <= 7 && g.L == FZ_NL && batch * ((int)((Tp - 1 + a.coff) / g.U) + 1) <= 3072) {
        const int Fu = (int)((Tp - 1 + a.coff) / g.U) + 1;           // frames per utterance
        const int n_fr = batch * Fu;
        const int ug = n_fr < 256 ? n_fr : 256;
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(bf16_stack_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    FZ_LDS_BYTES) != hipSuccess) return SWN_E_LAUNCH;
            attr_set = true;
        }
        hipLaunchKernelGGL(bf16_stack_fused_kernel, dim3(ug), dim3(768), FZ_LDS_BYTES, st, a, n_fr, Fu);
    } else
#endif
    if (g.seg == 1 && g.U >= 16 && nch <= 7) {
        const int Fu = (int)((Tp - 1 + a.coff) / g.U) + 1;           // frame units per utterance
        const int n_units = batch * Fu;
        const int ug = (n_units + 3) / 4 < 256 ? (n_units + 3) / 4 : 256;
        // few frames per wave: halve the units (4 + 3 chunks for U = 110) so that the waves finish together
        const bool halves = nch > 4 && nch <= 8 && n_units < 2 * 1024;
        const int n_half = 2 * n_units, ugh = (n_half + 3) / 4 < 256 ? (n_half + 3) / 4 : 256;
        for (int l = 0; l < g.L; ++l) {
            if (halves) hipLaunchKernelGGL(bf16_layer_units_kernel<4>, dim3(ugh), dim3(256), 0, st, a, l, g.dil[l], n_half, Fu, 2);
            else if (nch <= 4) hipLaunchKernelGGL(bf16_layer_units_kernel<4>, dim3(ug), dim3(256), 0, st, a, l, g.dil[l], n_units, Fu, 1);
            else if (nch == 5) hipLaunchKernelGGL(bf16_layer_units_kernel<5>, dim3(ug), dim3(256), 0, st, a, l, g.dil[l], n_units, Fu, 1);
            else hipLaunchKernelGGL(bf16_layer_units_kernel<7>, dim3(ug), dim3(256), 0, st, a, l, g.dil[l], n_units, Fu, 1);
        }
    } else {
        for (int l = 0; l < g.L; ++l)
            hipLaunchKernelGGL(bf16_layer_kernel<0>, dim3(grid), dim3(256), 0, st, a, l, g.dil[l], n_chunks);
    }
    const int hgrid = n_tiles < 512 ? n_tiles : 512;                            // two workgroups per CU
    hipLaunchKernelGGL(bf16_head_kernel<6>, dim3(hgrid), dim3(256), 4 * 12 * 64 * 16, st, a, n_tiles);
    return swn_launch_status("swn_forward_bf16");
}

// Training variant at the GEMM-stack geometries: the same forward, and every layer's gate pre-activations (fp32, (B, 2H, Tp)
// per layer) kept in a_keep_dev for swn_backward_keep, which then needs no recompute GEMM.  _keep_floats: 0 where the variant
// does not apply (BL6 class: swn_backward_bf16 recomputes on chip).
extern "C" size_t swn_forward_bf16_keep_floats(const swn_net_desc* d, int batch, int n_frames) {
    SwnGeom g;
    if (batch < 1 || n_frames < 1 || bf_geom(d, &g) == SWN_OK || swn_bf16g_geom(d, &g) < 0) return 0;
    const long Tp = (long)n_frames * g.U - 2 * g.seg + 1;
    return Tp < 1 ? 0 : swn_bf16g_keep_floats(g, batch, Tp);
}

extern "C" int swn_forward_bf16_keep(const swn_net_desc* d, const float* packed, const void* wbf, const float* cond,
                                     const void* audio_, int batch, int n_frames, void* work, float* out, float* a_keep,
                                     void* stream_) {
    SwnGeom g;
    if (bf_geom(d, &g) == SWN_OK) return SWN_E_UNSUPPORTED;
    const int rc = swn_bf16g_geom(d, &g);
    if (rc < 0) return rc;
    if (!packed || !wbf || !cond || !audio_ || !work || !out || !a_keep || batch < 1 || batch > 65535 || n_frames < 1) return SWN_E_BADARG;
    if ((long)n_frames * g.U - 2 * g.seg + 1 < 1) return SWN_E_BADARG;
    return swn_bf16g_forward(g, packed, wbf, cond, audio_, batch, n_frames, work, out, (hipStream_t)stream_, a_keep);
}

// After swn_forward_bf16: expand what the forward kept (bf16, time-major) into the fp32 work layout of swn_forward, so
// that swn_backward can follow a bf16 forward (mixed-precision training).  GEMM-stack class: hidden states, relu(skip)
// and relu(out_1) are all in memory.  BL6 class: the fused head keeps the two activations on chip, so the hidden
// states are expanded and the two 1x1 products are redone by the training contraction kernel, in the arithmetic mode
// given by `precision` (packed_dev needed only there).
extern "C" int swn_bf16_train_forward_supported(const swn_net_desc* d) {
    SwnGeom g;
    if (bf_geom(d, &g) == SWN_OK) return 1;
    return swn_bf16g_geom(d, &g) == SWN_OK ? 1 : 0;
}

extern "C" int swn_bf16_work_to_f32(const swn_net_desc* d, const float* packed, const void* work_bf16, int batch, int n_frames,
                                    float* fwd_work, int precision, void* stream_) {
    if (!swn_precision_ok(precision)) return SWN_E_BADARG;
    SwnModeScope mode(precision);
    SwnGeom g;
    const bool small = bf_geom(d, &g) == SWN_OK;
    if (!small) { const int rc = swn_bf16g_geom(d, &g); if (rc < 0) return rc; }
    if (!work_bf16 || !fwd_work || (small && !packed) || batch < 1 || batch > 65535 || n_frames < 1) return SWN_E_BADARG;
    const long Tp = (long)n_frames * g.U - 2 * g.seg + 1;
    if (Tp < 1) return SWN_E_BADARG;
    const int rc = swn_bf16g_expand(g, work_bf16, batch, Tp, fwd_work, small, (hipStream_t)stream_);
    if (rc < 0 || !small) return rc;
    return swn_train_head_acts(g, packed, fwd_work, batch, Tp, (hipStream_t)stream_);
}
