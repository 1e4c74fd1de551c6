// Fused backward of the BL6-class stack (H=64, K=2, seg=1, Laplace) in the mixed-precision training mode: the gradient
// side of CSWNV.forward (cswnv_shift1.py:191-278; loss.backward() of train_cswnv_laplace-stftcmplx_shift1.py:868-874) -
// head, gated layers, input layer - reading the bf16 time-major hidden states the bf16 forward (csrc/swn_stack_bf16.hip)
// kept.  Entry: swn_bl6_bwd_stack (called by swn_backward_bf16, csrc/swn_train.hip).
//
// The generic chain of csrc/swn_train.hip runs five launches per layer (recompute GEMM, gate, weight gradient, data
// gradient, conditioning) that hand 2H x T fp32 tensors to each other through HBM: ~750 MB per layer at BASELINE cfg4,
// which is what bounds it - plus an fp32 expansion of the bf16 activations and six launches for the head.  Here:
//
//   bl6_head_bwd_kernel    one launch: recompute relu(skip), relu(out_1); d out_1, d skip; g out_2 in registers;
//                          relu(skip), d out_1, d skip leave as bf16 [t][128] rows
//   bl6_layer_bwd_kernel   one launch per gated layer, per 16-position chunk (one wave, no barriers in the loop):
//        dh   = E_{l+1}(t) + [Wd_{l+1}^T | Wsk_l^T] (*) [da_{l+1} ; dskip]
//                                                          the data gradient of the layer ABOVE from the da it stored (so
//                                                          no halo is recomputed and nothing is atomically scattered),
//                                                          the skip path's share from dskip, and the highway carry E
//        a    = bd + Wd_l (*) h_l                          gate pre-activations, recomputed (bf16 MFMA, as the forward)
//        z, c, dz, dc, da = d a, dgx = d in_x-product      fp32
//        E_l  = dh * z                                     highway carry, fp32 TIME-MAJOR [t][64] (ping-pong buffers)
//        da_l -> bf16 time-major [t][128]                  operand of the next launch's data gradient and of the
//                                                          weight-gradient kernel
//        dcond[b][f][l] = sum_{t in f} w_up . dgx          work unit = one conditioning frame (or one channel half of it):
//                                                          plain stores
//        g b_inx, g w_up                                   lane / LDS accumulators, a few atomics per workgroup
//      and a last launch (MODE 2) that assembles d h_0 and does the input layer from its accumulators
//   bl6_wgrad_kernel       one launch, ten 128 x 128 jobs (six dil_h, three column blocks of out_skip, out_1) + their
//                          biases: time is the reduction axis, so both operands are staged time-major in LDS exactly as
//                          they lie in HBM and read back transposed by ds_read_b64_tr_b16.
// Every stream is time-major, i.e. whole 128/256-byte rows per position (a first version kept the carries in the generic
// chain's channel-major fp32 layout: 64-byte pieces of 64 different rows per chunk, 2.5 TB/s at best).
// Algorithmic HBM bytes per position: layer launch h 128 + E_{l+1} 256 + da_{l+1} 256 + dskip 256 + E_l 256 + da_l 256
// = 1.4 KB (the chain: ~5.7 KB, plus the skip path's data gradient as a separate 384 x 128 GEMM and a memset of the
// carries); head 1.8 KB; weight-gradient jobs 512 B each.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <hip/hip_bf16.h>
#include "swn_geom.hpp"

namespace {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s16x4;
typedef __attribute__((__vector_size__(8 * sizeof(short)))) short s16x8;

constexpr int H = 64;
constexpr unsigned OOB = 0x80000000u;       // host guarantees every buffer is smaller than 2 GiB
constexpr float K_SIG = -1.44269504f;       // sigmoid(x) = 1 / (1 + 2^(K_SIG x))
constexpr float K_TANH = 2.88539008f;       // tanh(x)    = 1 - 2 / (1 + 2^(K_TANH x))

__device__ __forceinline__ unsigned short f2bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }

// accumulator (tile m, lane group g, register r) <-> channel, the permutation of the forward's layer kernel: the 16
// channels a lane finishes are the ones its own tap-1 B fragments hold, and they are two runs of 8 consecutive channels
__device__ __host__ constexpr int chan_of(int m, int g, int r) { return 32 * (m >> 1) + 8 * g + 4 * (m & 1) + r; }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 ld_bf8(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ u32x4 ld_u4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}
__device__ __forceinline__ float4 ld_f4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float ld_f1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

// sum over the 16 lanes of a DPP row (lanes 16g .. 16g+15), result in every lane of the row
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float v) {
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);     // row_half_mirror
    v = dpp_add<0x140>(v);     // row_mirror
    return v;
}

struct BwArgs {
    const float* P; SwnLayout y;
    const float* cond;               // (B, Tf, N)
    const float* audio;              // (B, Tp) waveform input of the causal layer (seg == 1)
    const unsigned short* hs;        // [L+1][B][Tp][64] bf16: the forward's hidden states
    float* E[2];                     // [B][Tp][64] fp32 highway carries, ping-pong: layer l writes E[l & 1], reads E[(l+1) & 1]
    const unsigned short* dsk;       // [B][Tp][128] bf16: d relu(skip) pre-activation, time-major
    unsigned short* da;              // [L][B][Tp][128] bf16
    const unsigned short* wrec;      // [L][8 mt][4 ks][64 lanes][8]  Wd, rows permuted (chan_of)
    const unsigned short* wdg;       // [L+1][4 mt][12 ks][64 lanes][8]  image c: [Wd_c^T over (tap, o2) | Wsk_{c-1}^T], rows permuted
    float* dcond;                    // (B, Tf, N)
    const unsigned short* gx16;      // dropout mode (GX kernels): sample-rate in_x products [B][Tp][L*128] bf16 in place of cond ...
    unsigned short* dgx16;           // ... and their gradients [L][B][Tp][128] bf16 in place of dcond / g w_up
    float* gP;                       // gradient of the packed parameters
    int B, Tf, Tp, U, N, L, coff;
    int dil[SWN_MAXL];
};

// ---- fragment images ------------------------------------------------------------------------------------------
// recompute: A[row][k] = Wd[l][row'][k], k = tap*64 + i      (same image as the forward's pack_wd_kernel)
__global__ void pack_wrec_kernel(const float* __restrict__ wd, int L, unsigned short* __restrict__ dst) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= L * 8 * 4 * 512) return;
    const int j = e & 7, lane = (e >> 3) & 63, ks = (e >> 9) & 3, mt = (e >> 11) & 7, l = e >> 14;
    const int g = (lane & 15) >> 2, r = lane & 3;
    const int row = (mt >= 4 ? H : 0) + chan_of(mt & 3, g, r);
    const int k = 32 * ks + 8 * (lane >> 4) + j;
    dst[e] = f2bf(wd[((size_t)l * 128 + row) * 128 + k]);
}
// data gradient of consumer c (= the d h_c a layer kernel assembles), c = 0..L:
//   k-steps 0-3: Wd_c[o2][tap 0][i] <-> da_c(t + dil_c)   4-7: Wd_c[o2][tap 1][i] <-> da_c(t)      (zero for c = L)
//   k-steps 8-11: Wsk[s][(c-1)*64 + i] <-> dskip(t)                                                (zero for c = 0)
__global__ void pack_wdg_kernel(const float* __restrict__ wd, const float* __restrict__ wsk, int L, unsigned short* __restrict__ dst) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= (L + 1) * 4 * 12 * 512) return;
    const int j = e & 7, lane = (e >> 3) & 63, rest = e >> 9, ks = rest % 12, m = (rest / 12) & 3, c = rest / 48;
    const int rho = lane & 15, i = chan_of(m, rho >> 2, rho & 3);
    const int kk = 32 * (ks & 3) + 8 * (lane >> 4) + j;                 // o2 or s inside the 128-wide block
    float v = 0.f;
    if (ks < 8) { if (c < L) v = wd[(((size_t)c * 128 + kk) * 2 + (ks >> 2)) * H + i]; }
    else if (c >= 1) v = wsk[(size_t)kk * (L * H) + (c - 1) * H + i];
    dst[e] = f2bf(v);
}

// ---- one gated layer ----------------------------------------------------------------------------------------------
// MODE 0: top layer l = L-1 (d h_L = skip share only)   1: inner layer   2: l = -1: d h_0 = E_0 + Wd_0^T (*) da_0 and,
// straight from the accumulators, the input layer's backward (h_0 = softsign(cb + lifted causal taps), cswnv_shift1.py:
// 201-209 as packed: g cb, g cv, g cc) - d h_0 never goes to memory.
constexpr int LDS_REC = 32768;                                 // 128 x 128 bf16 fragment image (recompute)
constexpr int LDS_DG = 49152;                                  // 64 x 384 bf16 fragment image (data gradient | skip)
constexpr int BW_LDS = LDS_REC + LDS_DG + 4 * (256 + 128 + 128 + 128);

// HALF: a wave owns one HALF of the channels (32 of 64: accumulator tiles 2hh, 2hh+1) of its frame, two waves per frame.
// The work of a chunk and every per-lane accumulator halve, so twelve waves fit a CU (<= 170 registers): at BASELINE cfg4 the
// 1 200 frames become 2 400 units for 3 072 wave slots (whole frames: 1 200 units on 150 of the 256 CUs, 71.5 us per layer;
// halves: 68 us, against 45 at the HBM roof).  Both waves fetch all B fragments (the K axis is not split); at large batches,
// where the launch is HBM-bound, the whole-frame form is kept - except for the last launch, whose 64 extra accumulators for
// the input layer spill in the whole-frame form (390 -> 244 us at 64 x 16 500).
template <bool HALF> struct BwShape { static constexpr int NM = HALF ? 2 : 4, THREADS = HALF ? 768 : 512; };

// GX (dropout mode, aux_drop at sample rate: cswnv_shift1.py:194-195): the in_x products of a position are an operand (one
// bf16 row of a.gx16 per position) instead of w_up[j] * cond[f] + bx, and their gradient leaves as bf16 rows (a.dgx16) for
// the two in_x GEMMs of the caller instead of being reduced to d cond / g w_up here.
template <int MODE, bool HALF, bool GX = false>
__global__ __launch_bounds__(BwShape<HALF>::THREADS, 1) void bl6_layer_bwd_kernel(const BwArgs a, const int l, const int dil,
                                                                                 const int dil_up, const int n_units, const int Fu) {
    constexpr int NM = BwShape<HALF>::NM, THREADS = BwShape<HALF>::THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_rec = smem;
    unsigned char* s_dg = smem + LDS_REC;
    float* cst = reinterpret_cast<float*>(smem + LDS_REC + LDS_DG);   // bd[128] | bx[128]
    float* wus = cst + 256;                                    // upsampler taps [128] (U <= 112)
    float* gwl = wus + 128;                                    // g w_up of this workgroup
    float* gbl = gwl + 128;                                    // g bx of this workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    constexpr int KS0 = MODE == 0 ? 8 : 0, KS1 = MODE == 2 ? 8 : 12;   // k-steps of the d h GEMM this mode runs
    {
        const uint4* s0 = reinterpret_cast<const uint4*>(a.wrec) + (size_t)(MODE == 2 ? 0 : l) * (LDS_REC / 16);
        const uint4* s1 = reinterpret_cast<const uint4*>(a.wdg) + (size_t)(l + 1) * (LDS_DG / 16);
        if (MODE != 2)
            for (int e = tid; e < LDS_REC / 16; e += THREADS) reinterpret_cast<uint4*>(s_rec)[e] = s0[e];
        for (int e = tid; e < LDS_DG / 16; e += THREADS) reinterpret_cast<uint4*>(s_dg)[e] = s1[e];
        if (MODE != 2 && tid < 256)
            cst[tid] = tid < 128 ? a.P[a.y.bd + (size_t)l * 128 + tid] : a.P[a.y.bx + (size_t)l * 128 + tid - 128];
        if (MODE == 2 && tid < 320)                // cb | cv tap 0 | cv tap 1 | cc tap 0 | cc tap 1  (overlays wus, unused here)
            cst[tid] = tid < 64 ? a.P[a.y.cb + tid] : tid < 192 ? a.P[a.y.cv + tid - 64] : a.P[a.y.cc + tid - 192];
        if (tid >= 384 && tid < 512) { const int i = tid - 384; if (MODE != 2) wus[i] = i < a.U ? a.P[a.y.wup + i] : 0.f; gwl[i] = 0.f; gbl[i] = 0.f; }
    }
    __syncthreads();
    const size_t lstride = (size_t)a.B * a.Tp * H;                       // one layer of hidden states, elements
    const __amdgpu_buffer_rsrc_t rh = make_rsrc(a.hs + (size_t)(MODE == 2 ? 0 : l) * lstride, lstride * 2);
    const __amdgpu_buffer_rsrc_t rdu = make_rsrc(a.da + (size_t)(MODE == 0 ? 0 : l + 1) * lstride * 2, lstride * 4);
    const __amdgpu_buffer_rsrc_t rdo = make_rsrc(a.da + (size_t)(MODE == 2 ? 0 : l) * lstride * 2, lstride * 4);
    const __amdgpu_buffer_rsrc_t rsk = make_rsrc(a.dsk, lstride * 4);
    const __amdgpu_buffer_rsrc_t rei = make_rsrc(a.E[(l + 1) & 1], lstride * 4);       // E_{l+1}: read
    const __amdgpu_buffer_rsrc_t reo = make_rsrc(a.E[l & 1], lstride * 4);             // E_l: written
    const __amdgpu_buffer_rsrc_t rc = make_rsrc(a.cond, (size_t)a.B * a.Tf * a.N * 4);
    const __amdgpu_buffer_rsrc_t rau = make_rsrc(a.audio, (size_t)a.B * a.Tp * 4);
    const unsigned gx_row = (unsigned)a.L * 256u;                        // bytes of one position's in_x products
    const __amdgpu_buffer_rsrc_t rgx = make_rsrc(a.gx16, GX ? (size_t)a.B * a.Tp * gx_row : 0);
    const __amdgpu_buffer_rsrc_t rgo = make_rsrc(GX && MODE != 2 ? a.dgx16 + (size_t)l * lstride * 2 : nullptr, GX && MODE != 2 ? lstride * 4 : 0);
    const unsigned lane_h = (unsigned)(n * H + 8 * g) * 2u, lane_d = (unsigned)(n * 128 + 8 * g) * 2u;
    const unsigned lane_e = (unsigned)(n * H + 8 * g) * 4u;              // fp32 [t][64]: channels 8g.. of position n
    const unsigned dil_bytes = (unsigned)dil * H * 2u, dilu_bytes = (unsigned)dil_up * 256u;

    float gbx[2][NM][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) gbx[q][m][r] = 0.f;
    float inl[4][NM][4];                           // MODE 2: g cb | g cv tap 0 | g cv tap 1 | g cc tap 0 of this lane's channels
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) inl[q][m][r] = 0.f;

    const int Wn = gridDim.x * (THREADS / 64);
    const int n_jobs = HALF ? 2 * n_units : n_units;
    for (int jj_ = blockIdx.x * (THREADS / 64) + w; jj_ < n_jobs; jj_ += Wn) {
        const int j = HALF ? jj_ >> 1 : jj_;
        const int mb = HALF ? 2 * (jj_ & 1) : 0;                 // first accumulator tile of this wave: channels chan_of(mb + m, g, r)
        const unsigned chb = (unsigned)(32 * (mb >> 1));         // = 32 hh: first channel of the wave's range (tile pairs are 32 channels)
        const int b = j / Fu, f = j - b * Fu;
        const int s = f * a.U - a.coff;
        const int ts = s > 0 ? s : 0;
        int te = s + a.U; te = te < a.Tp ? te : a.Tp;
        const int jj0 = ts - s;
        float4 cz[NM] = {}, cc[NM] = {};
        if (MODE != 2 && !GX) {
            const int fc = f < a.Tf - 1 ? f : a.Tf - 1;
            const unsigned off = (unsigned)((b * a.Tf + fc) * a.N + l * 128 + 8 * g) * 4u + chb * 4u;
#pragma unroll
            for (int q = 0; q < NM; ++q) {         // tile mb + q: channels chb' + 8g + 4 (q & 1) ..
                const unsigned co = (unsigned)(32 * (q >> 1) + 4 * (q & 1)) * 4u;
                cz[q] = ld_f4(rc, off + co);
                cc[q] = ld_f4(rc, off + co + H * 4u);
            }
        }
        float dca[2][NM][4];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) dca[q][m][r] = 0.f;

        // every load of a chunk is issued together.  (Issuing the NEXT chunk's loads ahead of the current chunk's work - eight
        // waves, the freed registers holding them, pinned by a scheduling barrier - was slower, 84-86 us per layer: what
        // separates a frame's dependent chunks is not memory latency but the chunk's own chain: the compiler's schedule of the
        // LDS-fed MFMAs is read, read, wait, MFMA - one LDS latency per MFMA - and a sched_group_barrier pipeline made it
        // read, wait(0), MFMA.  More waves hide it better than anything tried inside one wave.)
        struct Loads { f32x4 D[NM]; bf16x8 du[12]; bf16x8 x[4]; float au0, au1; };
        auto fetch = [&](const int t0, Loads& q) {
            const int t = t0 + n;
            const bool ok = t < te;
            const unsigned pos = (unsigned)(b * a.Tp + t0);
            if (MODE != 0) {                       // highway carry of the layer above: channels chan_of(mb + m, g, 0..3)
                const unsigned eo = ok ? pos * 256u + lane_e + chb * 4u : OOB;
#pragma unroll
                for (int m = 0; m < NM; ++m)
                    q.D[m] = __builtin_bit_cast(f32x4, ld_u4(rei, eo + (unsigned)(32 * (m >> 1) + 4 * (m & 1)) * 4u));
            } else {
#pragma unroll
                for (int m = 0; m < NM; ++m) q.D[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            {
                const unsigned o1 = ok ? pos * 256u + lane_d : OOB;
                if (MODE != 0) {
                    const unsigned o0 = (ok && t + dil_up < a.Tp) ? o1 + dilu_bytes : OOB;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) { q.du[ks] = ld_bf8(rdu, o0 + 64u * ks); q.du[4 + ks] = ld_bf8(rdu, o1 + 64u * ks); }
                }
                if (MODE != 2) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) q.du[8 + ks] = ld_bf8(rsk, o1 + 64u * ks);
                }
            }
            q.au0 = 0.f; q.au1 = 0.f;
            if (MODE == 2) {
                q.au1 = ld_f1(rau, ok ? (pos + n) * 4u : OOB);
                q.au0 = ld_f1(rau, (ok && t >= 1) ? (pos + n - 1) * 4u : OOB);
            }
            if (MODE != 2) {
                const unsigned base = pos * (H * 2u) + lane_h;
                const unsigned o1 = ok ? base : OOB;
                const unsigned o0 = (ok && t >= dil) ? base - dil_bytes : OOB;
                q.x[0] = ld_bf8(rh, o0); q.x[1] = ld_bf8(rh, o0 + 64u); q.x[2] = ld_bf8(rh, o1); q.x[3] = ld_bf8(rh, o1 + 64u);
            }
        };
        for (int t0 = ts; t0 < te; t0 += 16) {
            const int t = t0 + n;
            const bool ok = t < te;
            const unsigned pos = (unsigned)(b * a.Tp + t0);
            Loads cur;
            fetch(t0, cur);
            // keep the loads HERE: left alone, the scheduler sinks each one next to its use to save registers, and the chunk
            // pays a memory round trip per k-step instead of one (whole-frame form at 64 x 16 500: 364 -> 329-347 us per layer,
            // 4.7-4.96 TB/s).  Not in the twelve-wave form: at 170 registers the pinned loads spill (68 -> 72 us).
            if (!HALF || MODE == 2) __builtin_amdgcn_sched_barrier(0);
            f32x4 (&D)[NM] = cur.D;
            bf16x8 (&du)[12] = cur.du;
            bf16x8 (&x)[4] = cur.x;
            const float au0 = cur.au0, au1 = cur.au1;
            // ---- d h_{l+1}: highway carry + data gradient of the layer above + the skip path's share
            {
                const unsigned char* ab = s_dg + (mb * 12 * 64 + lane) * 16;
#pragma unroll
                for (int ks = KS0; ks < KS1; ++ks)
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        const bf16x8 af = *reinterpret_cast<const bf16x8*>(ab + ((m * 12 + ks) * 64) * 16);
                        D[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, du[ks], D[m], 0, 0, 0);
                    }
            }
            // dropout mode: the in_x products of position t (runs of 8 channels, gate | candidate) are requested here, where the
            // twelve B fragments of the d h GEMM have just died: with the chunk's other loads they do not fit 168 registers
            u32x4 gqz[NM / 2], gqc[NM / 2];
            if (GX && MODE != 2) {
                const unsigned og = ok ? (pos + n) * gx_row + (unsigned)l * 256u + (chb + 8u * g) * 2u : OOB;
#pragma unroll
                for (int hlf = 0; hlf < NM / 2; ++hlf) { gqz[hlf] = ld_u4(rgx, og + 64u * hlf); gqc[hlf] = ld_u4(rgx, og + 128u + 64u * hlf); }
            }
            if (MODE == 2) {                       // input layer: h_0 = softsign(pre), pre = cb + [t >= 1](cv0 x(t-1) + cc0) + cv1 x(t) + cc1
                const float m0 = t >= 1 ? 1.f : 0.f;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const float* cp = cst + chb + 8 * g + 32 * (m >> 1) + 4 * (m & 1);
                    const f32x4 kb = *reinterpret_cast<const f32x4*>(cp), v0 = *reinterpret_cast<const f32x4*>(cp + 64),
                                v1 = *reinterpret_cast<const f32x4*>(cp + 128), c0 = *reinterpret_cast<const f32x4*>(cp + 192),
                                c1 = *reinterpret_cast<const f32x4*>(cp + 256);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pre = kb[r] + m0 * fmaf(v0[r], au0, c0[r]) + fmaf(v1[r], au1, c1[r]);
                        const float rd = __builtin_amdgcn_rcpf(1.f + fabsf(pre));
                        const float d = D[m][r] * rd * rd;                 // a position past the unit: D = 0
                        inl[0][m][r] += d;
                        inl[1][m][r] = fmaf(d, au0, inl[1][m][r]);         // au0 = 0 at t = 0
                        inl[2][m][r] = fmaf(d, au1, inl[2][m][r]);
                        inl[3][m][r] = fmaf(d, m0, inl[3][m][r]);
                    }
                }
                continue;
            }
            // ---- gate pre-activations: rows of the wave's channels, gate (tiles mb + m) and candidate (4 + mb + m)
            f32x4 acc[2 * NM];
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const float* cp = cst + chb + 8 * g + 32 * (m >> 1) + 4 * (m & 1);
                acc[m] = *reinterpret_cast<const f32x4*>(cp);
                acc[NM + m] = *reinterpret_cast<const f32x4*>(cp + H);
            }
            {
                const unsigned char* az = s_rec + (mb * 4 * 64 + lane) * 16;
                const unsigned char* ac = s_rec + ((4 + mb) * 4 * 64 + lane) * 16;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        const bf16x8 fz = *reinterpret_cast<const bf16x8*>(az + ((m * 4 + ks) * 64) * 16);
                        const bf16x8 fc = *reinterpret_cast<const bf16x8*>(ac + ((m * 4 + ks) * 64) * 16);
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fz, x[ks], acc[m], 0, 0, 0);
                        acc[NM + m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc, x[ks], acc[NM + m], 0, 0, 0);
                    }
            }
            // ---- gates and their derivatives (fp32)
            const float wu = GX ? 0.f : wus[jj0 + (t0 - ts) + n];
            float pw = 0.f;
            unsigned dav[2][2 * NM], dgv[2][2 * NM];
            f32x4 eout[NM];
            // the wave's own channels of h(t): tap-1 fragment 2 + (tile >> 1)
            const u32x4 hwa = __builtin_bit_cast(u32x4, x[2]), hwb = __builtin_bit_cast(u32x4, x[3]);
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const float* cp = cst + 128 + chb + 8 * g + 32 * (m >> 1) + 4 * (m & 1);
                const float4 bz4 = *reinterpret_cast<const float4*>(cp);
                const float4 bc4 = *reinterpret_cast<const float4*>(cp + H);
                const float czv[4] = {cz[m].x, cz[m].y, cz[m].z, cz[m].w}, ccv[4] = {cc[m].x, cc[m].y, cc[m].z, cc[m].w};
                const float bzv[4] = {bz4.x, bz4.y, bz4.z, bz4.w}, bcv[4] = {bc4.x, bc4.y, bc4.z, bc4.w};
                u32x4 hw;
                if (HALF) hw = mb ? hwb : hwa; else hw = (m >> 1) ? hwb : hwa;
                float daz[4], dac[4], dxz[4], dxc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float gz, gc;
                    if (GX) {
                        const unsigned uz = gqz[m >> 1][(m & 1) * 2 + (r >> 1)], uc = gqc[m >> 1][(m & 1) * 2 + (r >> 1)];
                        gz = __builtin_bit_cast(float, (r & 1) ? (uz & 0xffff0000u) : (uz << 16));
                        gc = __builtin_bit_cast(float, (r & 1) ? (uc & 0xffff0000u) : (uc << 16));
                    } else { gz = fmaf(wu, czv[r], bzv[r]); gc = fmaf(wu, ccv[r], bcv[r]); }
                    const float az = acc[m][r], ac = acc[NM + m][r];
                    const float z = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(K_SIG * gz * az));
                    const float q = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(K_TANH * gc * ac));
                    const float c = fmaf(-2.f, q, 1.f);
                    const unsigned hword = hw[(m & 1) * 2 + (r >> 1)];
                    const float hp = __builtin_bit_cast(float, (r & 1) ? (hword & 0xffff0000u) : (hword << 16));
                    const float dh = D[m][r];
                    const float omz = 1.f - z;
                    const float dz = dh * (hp - c) * z * omz;             // d / d(gz*az)
                    const float dc = dh * omz * fmaf(-c, c, 1.f);         // d / d(gc*ac)
                    daz[r] = dz * gz; dac[r] = dc * gc;
                    const float gxz = dz * az, gxc = dc * ac;            // d in_x products
                    eout[m][r] = dh * z;                                  // highway carry
                    gbx[0][m][r] += gxz; gbx[1][m][r] += gxc;
                    if (GX) { dxz[r] = gxz; dxc[r] = gxc; }
                    else {
                        dca[0][m][r] = fmaf(wu, gxz, dca[0][m][r]); dca[1][m][r] = fmaf(wu, gxc, dca[1][m][r]);
                        pw = fmaf(gxz, czv[r], pw); pw = fmaf(gxc, ccv[r], pw);
                    }
                }
                dav[0][m * 2] = pack2(daz[0], daz[1]); dav[0][m * 2 + 1] = pack2(daz[2], daz[3]);
                dav[1][m * 2] = pack2(dac[0], dac[1]); dav[1][m * 2 + 1] = pack2(dac[2], dac[3]);
                if (GX) {
                    dgv[0][m * 2] = pack2(dxz[0], dxz[1]); dgv[0][m * 2 + 1] = pack2(dxz[2], dxz[3]);
                    dgv[1][m * 2] = pack2(dxc[0], dxc[1]); dgv[1][m * 2 + 1] = pack2(dxc[2], dxc[3]);
                }
            }
            // ---- stores: da (bf16, [t][128]) and E_l (fp32, [t][64])
            {
                const unsigned so = ok ? pos * 256u + lane_d + chb * 2u : OOB;
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int hlf = 0; hlf < NM / 2; ++hlf) {
                        const u32x4 v = {dav[q][4 * hlf], dav[q][4 * hlf + 1], dav[q][4 * hlf + 2], dav[q][4 * hlf + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(v, rdo, so + 128u * q + 64u * hlf, 0, 0);
                        if (GX) {
                            const u32x4 vg = {dgv[q][4 * hlf], dgv[q][4 * hlf + 1], dgv[q][4 * hlf + 2], dgv[q][4 * hlf + 3]};
                            __builtin_amdgcn_raw_buffer_store_b128(vg, rgo, so + 128u * q + 64u * hlf, 0, 0);
                        }
                    }
                const unsigned eo = ok ? pos * 256u + lane_e + chb * 4u : OOB;
#pragma unroll
                for (int m = 0; m < NM; ++m)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, eout[m]), reo,
                                                           eo + (unsigned)(32 * (m >> 1) + 4 * (m & 1)) * 4u, 0, 0);
            }
            // g w_up[jj] += sum_o2 dgx[o2][t] * cond[f][o2]: finish the sum over the four lane groups (and, through the LDS
            // accumulator, over the two waves of a frame)
            // (without the LDS atomic below nothing separates two chunks for the scheduler: it then hoists the next chunk's loads
            //  over this chunk's gates and the half-frame form spills 97 registers)
            if (GX) __builtin_amdgcn_sched_barrier(0);
            if (!GX) {
                pw += __shfl_xor(pw, 16);
                pw += __shfl_xor(pw, 32);
                if (g == 0 && ok) atomicAdd(gwl + jj0 + (t0 - ts) + n, pw);
            }
        }
        if (MODE != 2 && !GX) {
            // dcond[b][f][l*128 + o2] = sum over the frame's positions
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dca[q][m][r] = row_sum16(dca[q][m][r]);
            if (n == 0) {
                float* dst = a.dcond + ((size_t)b * a.Tf + f) * a.N + (size_t)l * 128 + chb + 8 * g;
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int m = 0; m < NM; ++m)
                        *reinterpret_cast<float4*>(dst + q * H + 32 * (m >> 1) + 4 * (m & 1)) =
                            make_float4(dca[q][m][0], dca[q][m][1], dca[q][m][2], dca[q][m][3]);
            }
        }
        if (HALF) {                                // the wave's next unit may be the other half: flush its channel sums now
#pragma unroll
            for (int q = 0; q < (MODE == 2 ? 4 : 2); ++q)
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float& src = MODE == 2 ? inl[q][m][r] : gbx[q & 1][m][r];
                        const float v = row_sum16(src);
                        src = 0.f;
                        float* dstl = MODE == 2 ? gwl : gbl;               // MODE 2: gwl | gbl are 256 contiguous floats
                        if (n == 0) atomicAdd(dstl + q * H + chb + 8 * g + 4 * m + r, v);
                    }
        }
    }
    if (!HALF) {
#pragma unroll
        for (int q = 0; q < (MODE == 2 ? 4 : 2); ++q)
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = row_sum16(MODE == 2 ? inl[q][m][r] : gbx[q & 1][m][r]);
                    float* dstl = MODE == 2 ? gwl : gbl;
                    if (n == 0) atomicAdd(dstl + q * H + chan_of(m, g, r), v);
                }
    }
    __syncthreads();
    if (MODE == 2) {                               // g cb = g cc tap 1 | g cv tap 0 | g cv tap 1 | g cc tap 0
        if (tid < 256) {
            const float v = gwl[tid];
            const int o = tid & 63;
            if (tid < 64) { atomicAdd(a.gP + a.y.cb + o, v); atomicAdd(a.gP + a.y.cc + H + o, v); }
            else if (tid < 192) atomicAdd(a.gP + a.y.cv + tid - 64, v);
            else atomicAdd(a.gP + a.y.cc + o, v);
        }
        return;
    }
    // (dropout mode: in_x runs at sample rate on a conditioning that already holds b_up, its bias gradient is that of the
    //  raw bias - section bxr, csrc/swn_geom.hpp - and g w_up comes from the caller's xm backward)
    if (tid < 128) atomicAdd(a.gP + (GX ? a.y.bxr : a.y.bx) + (size_t)l * 128 + tid, gbl[tid]);
    else if (!GX && tid - 128 < a.U) atomicAdd(a.gP + a.y.wup + tid - 128, gwl[tid - 128]);
}

// ---- weight gradients, all of them in one launch --------------------------------------------------------------------------
// G[m][k] += sum_t A[t][m] * X[t][k]  (128 x 128 per job) and bias[m] += sum_t A[t][m], time the reduction axis:
//   dil_h of layer l    A = da_l            X(t) = [h_l(t - d_l) ; h_l(t)]
//   out_skip, 3 jobs    A = dskip           X(t) = [h_{2q+1}(t) ; h_{2q+2}(t)]        (columns 128q.. of the 128 x 384 matrix)
//   out_1               A = d out_1         X(t) = relu(skip)(t)
//   in_x of layer l, dropout mode (second launch): A = d gx_l, X(t) = 128 channels of the masked conditioning xm16(t)
// Workgroup = (time split, job); per step a 32-position tile of both operands is staged in LDS in the layout it has in
// HBM ([t][128] bf16, 16-byte chunks XOR-swizzled against bank conflicts) and the MFMA fragments - which want the
// reduction axis t inside a lane - are read with the transposing ds_read_b64_tr_b16.  Wave w owns rows 32w..32w+31 of the
// 128 x 128 result (16 accumulator tiles); the bias comes from an all-ones B fragment.
__device__ __forceinline__ unsigned tile_off(int row, int ch) { return 256u * row + 16u * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

struct WJob {
    const unsigned short* A;           // [B][Tp][128]
    const unsigned short* X0; const unsigned short* X1;     // left / right 64 columns of X: rows of rs0 / rs1 elements
    int rs0, rs1, sh0, sh1;            // row strides (elements) and time shifts (X0 is read at t - sh0, zero before the start)
    float* out; int ld; float* bias;   // G (row stride ld), bias (may be null)
    int ncols;                         // columns of the job that exist in G (128, less in the last column block of in_x)
};
constexpr int WG_MAXJOBS = SWN_MAXL + 4;
struct WgArgs { WJob job[WG_MAXJOBS]; int B, Tp; };

__global__ __launch_bounds__(256) void bl6_wgrad_kernel(const WgArgs a, const int tiles_per_b, const int n_tiles) {
    __shared__ __attribute__((aligned(16))) unsigned char img[2][2][32 * 256];     // [buffer][A | X][32 rows x 256 B]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const WJob& jb = a.job[blockIdx.y];
    const size_t npos = (size_t)a.B * a.Tp;
    const int lrow = tid >> 4, lch = tid & 15;
    const bool left = lch < 8;
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(jb.A, npos * 256);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(left ? jb.X0 : jb.X1, npos * (size_t)(left ? jb.rs0 : jb.rs1) * 2);
    const int rs = left ? jb.rs0 : jb.rs1, sh = left ? jb.sh0 : jb.sh1;
    u32x4 vd[2], vx[2];
    auto fetch = [&](int tix) {
        const bool tok = tix < n_tiles;
        const int tc = tok ? tix : n_tiles - 1;
        const int b = tc / tiles_per_b, t0 = (tc - b * tiles_per_b) * 32;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int t = t0 + lrow + 16 * i;
            const bool ok = tok && t < a.Tp;
            vd[i] = ld_u4(rd, ok ? (unsigned)((b * a.Tp + t) * 128 + 8 * lch) * 2u : OOB);
            const int tt = t - sh;
            vx[i] = ld_u4(rx, (ok && tt >= 0) ? (unsigned)((b * a.Tp + tt) * rs + 8 * (lch & 7)) * 2u : OOB);
        }
    };
    f32x4 acc[2][8], accb[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        accb[mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) acc[mi][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const s16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    // transposed-read addresses: lane 4q+p of group kg supplies row 8kg + 4h + q, 16-byte chunk 2*tile + (p>>1), half p&1
    const int q = n >> 2, p = n & 3;
    unsigned roff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) roff[h] = (unsigned)(8 * g + 4 * h + q);
    fetch(blockIdx.x);
    int cur = 0;
    for (int tix = blockIdx.x; tix < n_tiles; tix += gridDim.x) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned o = tile_off(lrow + 16 * i, lch);
            *reinterpret_cast<u32x4*>(&img[cur][0][o]) = vd[i];
            *reinterpret_cast<u32x4*>(&img[cur][1][o]) = vx[i];
        }
        fetch(tix + gridDim.x);
        __syncthreads();
        auto frag = [&](const unsigned char* im, int tile) -> bf16x8 {
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(im + tile_off((int)roff[0], 2 * tile + (p >> 1)) + 8 * (p & 1)));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(im + tile_off((int)roff[1], 2 * tile + (p >> 1)) + 8 * (p & 1)));
            const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            return __builtin_bit_cast(bf16x8, v);
        };
        bf16x8 af[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            af[mi] = frag(&img[cur][0][0], 2 * w + mi);
            accb[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], __builtin_bit_cast(bf16x8, ones), accb[mi], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            const bf16x8 bfr = frag(&img[cur][1][0], nt);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) acc[mi][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mi], bfr, acc[mi][nt], 0, 0, 0);
        }
        cur ^= 1;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * (2 * w + mi) + 4 * g + r;
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
                if (16 * nt + n < jb.ncols) atomicAdd(jb.out + (size_t)row * jb.ld + 16 * nt + n, acc[mi][nt][r]);
            if (n == 0 && jb.bias) atomicAdd(jb.bias + row, accb[mi][r]);
        }
}

// ---- head: out_2 / out_1 / skip backward ------------------------------------------------------------------------------------
// Per 64-position tile (8 waves, wave w owns rows 16w..16w+15 of every 128-row product, all four 16-position column tiles):
//   recompute  s1 = relu(bsk + Wsk . [h_1..h_L]),  r1 = relu(b1 + W1 . s1)                 (what the forward's head kernel kept on chip)
//   d out_1 = [r1 > 0] . W2^T dY          d skip = [s1 > 0] . W1^T d out_1
//   g W2 += dY . r1^T, g b2 += sum dY     (the only products with NO <= 16 rows: accumulated here, one column tile per wave)
//   s1, d out_1, d skip -> bf16 time-major [t][128]: operands of the weight-gradient launch; d skip also feeds the layer kernels.
// Fragment images (pack_frag / pack_fragT below): A[row][k] row-major in k, [mt][ks][lane][8].
__global__ void pack_frag_kernel(const float* __restrict__ src, int ld, int rows, int cols, int MT, int KS, unsigned short* __restrict__ dst) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= MT * KS * 512) return;
    const int j = e & 7, lane = (e >> 3) & 63, ks = (e >> 9) % KS, mt = (e >> 9) / KS;
    const int r = 16 * mt + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + j;
    dst[e] = (r < rows && k < cols) ? f2bf(src[(size_t)r * ld + k]) : (unsigned short)0;
}
// transposed source: A[row][k] = W[k][row]
__global__ void pack_fragT_kernel(const float* __restrict__ src, int ld, int krows, int rcols, int MT, int KS, unsigned short* __restrict__ dst) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= MT * KS * 512) return;
    const int j = e & 7, lane = (e >> 3) & 63, ks = (e >> 9) % KS, mt = (e >> 9) / KS;
    const int r = 16 * mt + (lane & 15), k = 32 * ks + 8 * (lane >> 4) + j;
    dst[e] = (r < rcols && k < krows) ? f2bf(src[(size_t)k * ld + r]) : (unsigned short)0;
}

constexpr int HB_THREADS = 512, HB_TN = 64, HB_ROW = 272;            // LDS tile pitch in bytes of a [pos][128 bf16] tile
constexpr int HB_BT = 4 * 12 * 64 * 16;                              // published B fragments of [h_1..h_6] for 64 positions
constexpr int HB_LDS = HB_BT + 4 * HB_TN * HB_ROW;
struct HbArgs {
    const float* P; SwnLayout y;
    const unsigned short* hs;          // [L+1][B][Tp][64]
    const float* dY;                   // (B, NO, Tp) fp32: gradient wrt the raw out_2 outputs
    const unsigned short *wsk, *w1, *w1t, *w2t;        // fragment images [8][12] / [8][4] / [8][4] / [8][1]
    unsigned short *s1, *do1, *dsk;    // [B][Tp][128] bf16 outputs
    float* gP;
    int B, Tp, NO, O1p;
};

__global__ __launch_bounds__(HB_THREADS, 1) void bl6_head_bwd_kernel(const HbArgs a, const int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    unsigned char* bt = sm;
    unsigned char* t1 = sm + HB_BT;                     // relu(skip)
    unsigned char* t2 = t1 + HB_TN * HB_ROW;            // relu(out_1)
    unsigned char* t3 = t2 + HB_TN * HB_ROW;            // d out_1
    unsigned char* t4 = t3 + HB_TN * HB_ROW;            // d skip
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    bf16x8 Ask[12], A1[4], A1T[4], A2T;
    {
        const bf16x8* i0 = reinterpret_cast<const bf16x8*>(a.wsk);
        const bf16x8* i1 = reinterpret_cast<const bf16x8*>(a.w1);
        const bf16x8* i2 = reinterpret_cast<const bf16x8*>(a.w1t);
        const bf16x8* i3 = reinterpret_cast<const bf16x8*>(a.w2t);
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) Ask[ks] = i0[(w * 12 + ks) * 64 + lane];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { A1[ks] = i1[(w * 4 + ks) * 64 + lane]; A1T[ks] = i2[(w * 4 + ks) * 64 + lane]; }
        A2T = i3[w * 64 + lane];
    }
    float bsk[4], b1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { bsk[r] = a.P[a.y.bsk + 16 * w + 4 * g + r]; b1[r] = a.P[a.y.b1 + 16 * w + 4 * g + r]; }
    const size_t lstride = (size_t)a.B * a.Tp * H;
    const int tiles_per_b = (a.Tp + HB_TN - 1) / HB_TN;
    __amdgpu_buffer_rsrc_t rl[6];
#pragma unroll
    for (int qq = 0; qq < 6; ++qq) rl[qq] = make_rsrc(a.hs + (size_t)(1 + qq) * lstride, lstride * 2);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.dY, (size_t)a.B * a.NO * a.Tp * 4);
    const __amdgpu_buffer_rsrc_t ro[3] = {make_rsrc(a.s1, lstride * 4), make_rsrc(a.do1, lstride * 4), make_rsrc(a.dsk, lstride * 4)};
    // wave w fetches column tile w & 3, k-steps 6 (w >> 2) .. + 5 of the skip GEMM's B operand, one tile ahead
    const int cnt = w & 3, ck0 = 6 * (w >> 2);
    bf16x8 mycol[6];
    auto fetch_col = [&](int tix) {
        const int tc = tix < n_tiles ? tix : n_tiles - 1;
        const int b = tc / tiles_per_b, t = (tc - b * tiles_per_b) * HB_TN + 16 * cnt + n;
        const unsigned off = (tix < n_tiles && t < a.Tp) ? (unsigned)(b * a.Tp + t) * (H * 2u) + (unsigned)(8 * g) * 2u : OOB;
#pragma unroll
        for (int u = 0; u < 6; ++u) { const int ks = ck0 + u; mycol[u] = ld_bf8(rl[ks >> 1], off + 64u * (ks & 1)); }
    };
    fetch_col(blockIdx.x);
    f32x4 acc_w2 = {0.f, 0.f, 0.f, 0.f}, acc_b2 = {0.f, 0.f, 0.f, 0.f};
    const s16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const int q = n >> 2, p = n & 3;

    for (int tix = blockIdx.x; tix < n_tiles; tix += gridDim.x) {
        const int b = tix / tiles_per_b, t0 = (tix - b * tiles_per_b) * HB_TN;
#pragma unroll
        for (int u = 0; u < 6; ++u) *reinterpret_cast<bf16x8*>(bt + ((cnt * 12 + ck0 + u) * 64 + lane) * 16) = mycol[u];
        fetch_col(tix + gridDim.x);
        // dY in both orientations (tiny: NO <= 16 rows), issued ahead of the GEMMs
        bf16x8 dyb[4], dya[2];
        {
            float v[4][8], u2[2][8];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)             // B operand of W2^T . dY: lane (position, k group): k = output row 8g + j
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int no = 8 * g + j, t = t0 + 16 * nt + n;
                    v[nt][j] = ld_f1(ry, (no < a.NO && t < a.Tp) ? (unsigned)((b * a.NO + no) * a.Tp + t) * 4u : OOB);
                }
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)             // A operand of dY . r1^T: lane (row = output row n, k group): 8 consecutive positions
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = t0 + 32 * k2 + 8 * g + j;
                    u2[k2][j] = ld_f1(ry, (n < a.NO && t < a.Tp) ? (unsigned)((b * a.NO + n) * a.Tp + t) * 4u : OOB);
                }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const u32x4 pk = {pack2(v[nt][0], v[nt][1]), pack2(v[nt][2], v[nt][3]), pack2(v[nt][4], v[nt][5]), pack2(v[nt][6], v[nt][7])};
                dyb[nt] = __builtin_bit_cast(bf16x8, pk);
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const u32x4 pk = {pack2(u2[k2][0], u2[k2][1]), pack2(u2[k2][2], u2[k2][3]), pack2(u2[k2][4], u2[k2][5]), pack2(u2[k2][6], u2[k2][7])};
                dya[k2] = __builtin_bit_cast(bf16x8, pk);
            }
        }
        __syncthreads();                                                   // (A) bt published
        // ---- skip = Wsk . [h_1 .. h_L]
        // k-step outer, the four column tiles' fragments read as one batch: four LDS reads in flight per four MFMAs (column
        // tile outer, the schedule was read, read, wait, MFMA: one LDS latency per MFMA)
        f32x4 s1v[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) s1v[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
            bf16x8 bf[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bf[nt] = *reinterpret_cast<const bf16x8*>(bt + ((nt * 12 + ks) * 64 + lane) * 16);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) s1v[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Ask[ks], bf[nt], s1v[nt], 0, 0, 0);
        }
        auto put = [&](unsigned char* tl, const int nt, const f32x4& v) {   // rows 16w + 4g .. + 3 of position 16nt + n
            uint2 pk; pk.x = pack2(v[0], v[1]); pk.y = pack2(v[2], v[3]);
            *reinterpret_cast<uint2*>(tl + (16 * nt + n) * HB_ROW + (16 * w + 4 * g) * 2) = pk;
        };
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s1v[nt][r] = fmaxf(s1v[nt][r] + bsk[r], 0.f);
            put(t1, nt, s1v[nt]);
        }
        __syncthreads();                                                   // (B) t1 complete
        // ---- out_1 and d out_1
        f32x4 r1v[4], dv[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) r1v[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 bf[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bf[nt] = *reinterpret_cast<const bf16x8*>(t1 + (16 * nt + n) * HB_ROW + (32 * ks + 8 * g) * 2);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) r1v[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1[ks], bf[nt], r1v[nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            dv[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2T, dyb[nt], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                r1v[nt][r] = fmaxf(r1v[nt][r] + b1[r], 0.f);
                dv[nt][r] = r1v[nt][r] > 0.f ? dv[nt][r] : 0.f;
            }
            put(t2, nt, r1v[nt]);
            put(t3, nt, dv[nt]);
        }
        __syncthreads();                                                   // (C) t2, t3 complete
        // ---- g W2 column tile w (+ g b2 on wave 0): B = r1 read transposed from t2, rows = positions
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const unsigned char* base = t2 + (32 * k2 + 8 * g + q) * HB_ROW + (16 * w + 4 * p) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * HB_ROW));
            const s16x8 vv = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            acc_w2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dya[k2], __builtin_bit_cast(bf16x8, vv), acc_w2, 0, 0, 0);
            if (w == 0) acc_b2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dya[k2], __builtin_bit_cast(bf16x8, ones), acc_b2, 0, 0, 0);
        }
        // ---- d skip = [s1 > 0] . W1^T d out_1
        {
            f32x4 kv[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) kv[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 bf[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) bf[nt] = *reinterpret_cast<const bf16x8*>(t3 + (16 * nt + n) * HB_ROW + (32 * ks + 8 * g) * 2);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) kv[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1T[ks], bf[nt], kv[nt], 0, 0, 0);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) kv[nt][r] = s1v[nt][r] > 0.f ? kv[nt][r] : 0.f;
                put(t4, nt, kv[nt]);
            }
        }
        __syncthreads();                                                   // (D) t4 complete
        // ---- s1, d out_1, d skip -> HBM, whole 256-byte rows
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int which = i >> 1, e = tid + HB_THREADS * (i & 1), pos = e >> 4, ch = e & 15;     // 1024 chunks per tile
            const unsigned char* tl = which == 0 ? t1 : which == 1 ? t3 : t4;
            const u32x4 v = *reinterpret_cast<const u32x4*>(tl + pos * HB_ROW + ch * 16);
            const int t = t0 + pos;
            __builtin_amdgcn_raw_buffer_store_b128(v, ro[which], t < a.Tp ? (unsigned)((b * a.Tp + t) * 128 + 8 * ch) * 2u : OOB, 0, 0);
        }
        __syncthreads();                                                   // (E) tiles free
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int no = 4 * g + r;
        if (no < a.NO) {
            atomicAdd(a.gP + a.y.w2 + (size_t)no * a.O1p + 16 * w + n, acc_w2[r]);
            if (w == 0 && n == 0) atomicAdd(a.gP + a.y.b2 + no, acc_b2[r]);
        }
    }
}

template <int MODE, bool HALF, bool GX>
int launch_layer_h(const BwArgs& a, int l, int dil, int dil_up, int n_units, int Fu, hipStream_t st) {
    auto kern = bl6_layer_bwd_kernel<MODE, HALF, GX>;
    static bool attr_done = false;                                // per instantiation; the attribute is per function
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, BW_LDS) != hipSuccess)
            return SWN_E_LAUNCH;
        attr_done = true;
    }
    constexpr int wpw = BwShape<HALF>::THREADS / 64;
    const int jobs = HALF ? 2 * n_units : n_units;
    const int grid = (jobs + wpw - 1) / wpw < 256 ? (jobs + wpw - 1) / wpw : 256;     // one workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BwShape<HALF>::THREADS), BW_LDS, st, a, l, dil, dil_up, n_units, Fu);
    return SWN_OK;
}
// half-frame units while the frames alone cannot fill the chip's wave slots twice over (see BwShape)
template <int MODE, bool GX = false>
int launch_layer(const BwArgs& a, int l, int dil, int dil_up, int n_units, int Fu, hipStream_t st) {
    // (the last launch carries 64 more accumulators per lane for the input layer: always in halves, or it spills)
    return (MODE == 2 || n_units < 2 * 2048) ? launch_layer_h<MODE, true, GX>(a, l, dil, dil_up, n_units, Fu, st)
                                             : launch_layer_h<MODE, false, GX>(a, l, dil, dil_up, n_units, Fu, st);
}
// fp32 [rows][ld] -> transposed bf16 [cols (padded to dcols rows... )]: dst[c][r] = src[r][c] for c < cols, zero rows c >= cols
__global__ void rowsT_to_bf16_kernel(const float* __restrict__ src, int ld, int rows, int cols, int drows, unsigned short* __restrict__ dst) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)drows * rows) return;
    const int c = (int)(e / rows), r = (int)(e - (size_t)c * rows);
    dst[e] = c < cols ? f2bf(src[(size_t)r * ld + c]) : (unsigned short)0;
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

// ---- interface to csrc/swn_train.hip ------------------------------------------------------------------------------------
bool swn_bl6_bwd_supported(const SwnGeom& g, int B, long Tp, int n_frames) {
    if (!g.bl6 || g.kind != SWN_KIND_LAPLACE || g.seg != 1 || g.Hp != g.H || g.S != 128 || g.U < 16 || g.U > 112) return false;
    // 32-bit buffer offsets
    if ((size_t)B * (g.L + 1) * g.H * Tp * 4 >= (1ull << 31) || (size_t)B * Tp * 256 >= (1ull << 31) ||
        (size_t)B * n_frames * g.N * 4 >= (1ull << 31)) return false;
    return true;
}

constexpr size_t IMG_WSK = 8 * 12 * 1024, IMG_W1 = 8 * 4 * 1024, IMG_W2T = 8 * 1024;     // fragment images, bytes

size_t swn_bl6_bwd_scratch_bytes(const SwnGeom& g, int B, long Tp) {
    const size_t row = al256((size_t)B * Tp * 256);              // one [B][Tp][128] bf16 or [B][Tp][64] fp32 buffer
    return al256((size_t)g.L * LDS_REC) + al256((size_t)(g.L + 1) * LDS_DG) + IMG_WSK + 2 * IMG_W1 + IMG_W2T +
           (size_t)(g.L + 5) * row;
}
// dropout mode: + d gx of every layer [L][B][Tp][128] bf16 and the transposed in_x image [A0x][L*128] bf16
size_t swn_bl6_bwd_drop_scratch_bytes(const SwnGeom& g, int B, long Tp) {
    return swn_bl6_bwd_scratch_bytes(g, B, Tp) + al256((size_t)g.L * B * Tp * 256) + al256((size_t)swn_a0x(&g) * g.L * 128 * 2);
}
int swn_bf16g_plain(const unsigned short* A, int M, const unsigned short* src, size_t blk_stride, size_t src_bytes, int KB, int nblk,
                    int Tp, int B, const float* bias, unsigned short* out_bf, int out_ld, float* out_f, int NO, hipStream_t st);

// The whole stack backward behind d raw: head (out_2, out_1, skip), the gated layers, the input layer.  Fills dcond and every
// sample-rate section of gpacked (zeroed by the caller): w2 b2 w1 b1 wsk bsk wd bd bx wup cb cv cc.
// Dropout mode (gx16 != null: the forward was swn_bl6_drop_forward; cond and dcond are not used, scratch holds
// swn_bl6_bwd_drop_scratch_bytes): fills wx and bxr instead of bx / wup, and dxm16 [B][Tx][A0x] bf16 = in_x^T d gx, the
// gradient wrt the masked conditioning (the caller's xm backward turns it into d C, g w_up, g b_up).
int swn_bl6_bwd_stack(const SwnGeom& g, const SwnLayout& y, const float* packed, const float* cond, const float* audio,
                      const void* hs_bf16, const float* grad_out, float* dcond, float* gpacked, void* scratch, int B, int n_frames,
                      long Tp, hipStream_t st, const unsigned short* gx16, const unsigned short* xm16, unsigned short* dxm16) {
    BwArgs a;
    a.P = packed; a.y = y; a.cond = cond; a.audio = audio; a.hs = reinterpret_cast<const unsigned short*>(hs_bf16);
    unsigned char* p = reinterpret_cast<unsigned char*>(scratch);
    const size_t row = al256((size_t)B * Tp * 256);
    unsigned short* wrec = reinterpret_cast<unsigned short*>(p); p += al256((size_t)g.L * LDS_REC);
    unsigned short* wdg = reinterpret_cast<unsigned short*>(p);  p += al256((size_t)(g.L + 1) * LDS_DG);
    unsigned short* iwsk = reinterpret_cast<unsigned short*>(p); p += IMG_WSK;
    unsigned short* iw1 = reinterpret_cast<unsigned short*>(p);  p += IMG_W1;
    unsigned short* iw1t = reinterpret_cast<unsigned short*>(p); p += IMG_W1;
    unsigned short* iw2t = reinterpret_cast<unsigned short*>(p); p += IMG_W2T;
    a.E[0] = reinterpret_cast<float*>(p); p += row;
    a.E[1] = reinterpret_cast<float*>(p); p += row;
    unsigned short* dsk = reinterpret_cast<unsigned short*>(p); p += row;
    unsigned short* s1 = reinterpret_cast<unsigned short*>(p);  p += row;
    unsigned short* do1 = reinterpret_cast<unsigned short*>(p); p += row;
    a.dsk = dsk;
    a.da = reinterpret_cast<unsigned short*>(p);                 // L buffers of B*Tp*256 bytes, contiguous (no padding)
    p = reinterpret_cast<unsigned char*>(scratch) + swn_bl6_bwd_scratch_bytes(g, B, Tp);
    a.gx16 = gx16; a.dgx16 = gx16 ? reinterpret_cast<unsigned short*>(p) : nullptr;
    unsigned short* wxt = reinterpret_cast<unsigned short*>(p + al256((size_t)g.L * B * Tp * 256));      // dropout mode only
    a.wrec = wrec; a.wdg = wdg;
    a.dcond = dcond; a.gP = gpacked;
    a.B = B; a.Tf = n_frames; a.Tp = (int)Tp; a.U = g.U; a.N = g.N; a.L = g.L; a.coff = g.seg;
    for (int l = 0; l < SWN_MAXL; ++l) a.dil[l] = l < g.L ? g.dil[l] : 1;
    (void)hipGetLastError();
    hipLaunchKernelGGL(pack_wrec_kernel, dim3((g.L * 8 * 4 * 512 + 255) / 256), dim3(256), 0, st, packed + y.wd, g.L, wrec);
    hipLaunchKernelGGL(pack_wdg_kernel, dim3(((g.L + 1) * 4 * 12 * 512 + 255) / 256), dim3(256), 0, st, packed + y.wd,
                       packed + y.wsk, g.L, wdg);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(8 * 12 * 2), dim3(256), 0, st, packed + y.wsk, g.L * g.Hp, g.S, g.L * g.H, 8, 12, iwsk);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(8 * 4 * 2), dim3(256), 0, st, packed + y.w1, g.Sp, g.O1, g.S, 8, 4, iw1);
    hipLaunchKernelGGL(pack_fragT_kernel, dim3(8 * 4 * 2), dim3(256), 0, st, packed + y.w1, g.Sp, g.O1, g.S, 8, 4, iw1t);
    hipLaunchKernelGGL(pack_fragT_kernel, dim3(8 * 2), dim3(256), 0, st, packed + y.w2, g.O1p, g.NO, g.O1, 8, 1, iw2t);
    // ---- head
    {
        HbArgs h;
        h.P = packed; h.y = y; h.hs = a.hs; h.dY = grad_out;
        h.wsk = iwsk; h.w1 = iw1; h.w1t = iw1t; h.w2t = iw2t;
        h.s1 = s1; h.do1 = do1; h.dsk = dsk; h.gP = gpacked;
        h.B = B; h.Tp = (int)Tp; h.NO = g.NO; h.O1p = g.O1p;
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(bl6_head_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    HB_LDS) != hipSuccess) return SWN_E_LAUNCH;
            attr_done = true;
        }
        const int n_tiles = B * (int)((Tp + HB_TN - 1) / HB_TN);
        hipLaunchKernelGGL(bl6_head_bwd_kernel, dim3(n_tiles < 256 ? n_tiles : 256), dim3(HB_THREADS), HB_LDS, st, h, n_tiles);
    }
    // ---- gated layers, input layer
    const int Fu = (int)((Tp - 1 + a.coff) / g.U) + 1;               // frame units per utterance
    const int n_units = B * Fu;
    int rc = SWN_OK;
    for (int l = g.L - 1; l >= 0 && rc == SWN_OK; --l) {
        if (gx16) {
            if (l == g.L - 1) rc = launch_layer<0, true>(a, l, g.dil[l], 1, n_units, Fu, st);
            else rc = launch_layer<1, true>(a, l, g.dil[l], g.dil[l + 1], n_units, Fu, st);
        } else if (l == g.L - 1) rc = launch_layer<0>(a, l, g.dil[l], 1, n_units, Fu, st);
        else rc = launch_layer<1>(a, l, g.dil[l], g.dil[l + 1], n_units, Fu, st);
    }
    if (rc == SWN_OK) rc = launch_layer<2>(a, -1, 1, g.dil[0], n_units, Fu, st);
    if (rc != SWN_OK) return rc;
    // ---- weight gradients: L dil_h jobs, L/2 out_skip column blocks, out_1
    {
        WgArgs wa;
        wa.B = B; wa.Tp = (int)Tp;
        const size_t lstride = (size_t)B * Tp * H;
        int nj = 0;
        for (int l = 0; l < g.L; ++l) {
            const unsigned short* hl = a.hs + (size_t)l * lstride;
            wa.job[nj++] = WJob{a.da + (size_t)l * lstride * 2, hl, hl, H, H, g.dil[l], 0, gpacked + y.wd + (size_t)l * 128 * 128, 128,
                                gpacked + y.bd + (size_t)l * 128, 128};
        }
        for (int qq = 0; qq < g.L / 2; ++qq)
            wa.job[nj++] = WJob{dsk, a.hs + (size_t)(2 * qq + 1) * lstride, a.hs + (size_t)(2 * qq + 2) * lstride, H, H, 0, 0,
                                gpacked + y.wsk + 128 * qq, g.L * g.Hp, qq == 0 ? gpacked + y.bsk : nullptr, 128};
        wa.job[nj++] = WJob{do1, s1, s1 + H, 128, 128, 0, 0, gpacked + y.w1, g.Sp, gpacked + y.b1, 128};
        const int tiles_per_b = (int)((Tp + 31) / 32), n_tiles = B * tiles_per_b;
        const int split = n_tiles < 96 ? n_tiles : 96;
        hipLaunchKernelGGL(bl6_wgrad_kernel, dim3(split, nj), dim3(256), 0, st, wa, tiles_per_b, n_tiles);
        if (gx16) {
            // g in_x.W[l][o][c] = sum_t d gx_l[t][o] xm[t][c] in column blocks of 128 conditioning channels (g b_inx: the layer kernels)
            const int A0x = swn_a0x(&g), ncb = (g.A0 + 127) / 128;
            nj = 0;
            for (int l = 0; l < g.L; ++l)
                for (int cb = 0; cb < ncb; ++cb) {
                    // (a last block of 32 or 64 columns past A0x reads the neighbouring row's first channels: finite values, never stored)
                    wa.job[nj++] = WJob{a.dgx16 + (size_t)l * lstride * 2, xm16 + 128 * cb, xm16 + 128 * cb + 64, A0x, A0x, 0, 0,
                                        gpacked + y.wx + (size_t)l * 128 * g.A0p + 128 * cb, g.A0p, nullptr,
                                        g.A0 - 128 * cb < 128 ? g.A0 - 128 * cb : 128};
                    if (nj == WG_MAXJOBS || (l == g.L - 1 && cb == ncb - 1)) {
                        const int split2 = n_tiles < 48 ? n_tiles : 48;      // 2.71 / 2.63 / 2.63 / 2.84 ms per step at 96 / 48 / 32 / 160
                        hipLaunchKernelGGL(bl6_wgrad_kernel, dim3(split2, nj), dim3(256), 0, st, wa, tiles_per_b, n_tiles);
                        nj = 0;
                    }
                }
        }
    }
    if (gx16) {
        // dxm16[b][u][c] = sum_{l,o} in_x[l].W[o][c] d gx_l[b][u][o]: A = the transposed matrices [c][l*128 + o] (zero rows
        // A0..A0x), k = L blocks of 128
        const int A0x = swn_a0x(&g);
        hipLaunchKernelGGL(rowsT_to_bf16_kernel, dim3((unsigned)(((size_t)A0x * g.L * 128 + 255) / 256)), dim3(256), 0, st,
                           packed + y.wx, g.A0p, g.L * 128, g.A0, A0x, wxt);
        const size_t lstride2 = (size_t)B * Tp * 128;
        const int rcg = swn_bf16g_plain(wxt, A0x, a.dgx16, lstride2, (size_t)g.L * lstride2 * 2, 128, g.L, (int)Tp, B, nullptr,
                                        dxm16, A0x, nullptr, 0, st);
        if (rcg < 0) return rcg;
    }
    return swn_launch_status("swn_backward_bf16");
}
