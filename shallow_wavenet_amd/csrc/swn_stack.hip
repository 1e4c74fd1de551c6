// Teacher-forced dilated stack on gfx950 (CSWNV.forward cswnv_shift1.py:191-267,
// DSWNV.forward dswnv.py:250-276): all T positions in parallel, layer by layer.
//
// fp32 reference-parity path (this file):
//   input layer   h0 = softsign(causal(lift(audio)))        fused wav_conv + causal taps / gather table
//   layer l       a = Wd_l (*) h_{l-1}  (dilated causal conv as a GEMM over K shifted copies of h)
//                 g = in_x_l(x) (.) a ; z = sigmoid(g[:H]) ; h_l = (1-z) tanh(g[H:]) + z h_{l-1}
//                 -- gate, highway and the hoisted conditioning fused into the GEMM epilogue --
//   head          relu(Wsk . [h_1 .. h_L] + bsk) -> relu(W1 . + b1) -> W2 . + b2
//                 (the six out_skip 1x1s are ONE GEMM over the concatenated hidden states)
// Hidden states are kept as (B, L+1, H, Tp) so the head reads them as one (L*H)-channel tensor and
// a backward pass can reuse them.  Every product is an fp32 fma chain in ascending-k order.
#include <hip/hip_runtime.h>
#include "swn_geom.hpp"
#include "swn_mma.hpp"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float ssign(float x) { return x / (1.f + fabsf(x)); }

struct FwdArgs {
    SwnGeom g;
    SwnLayout y;
    const float* P;
    const float* cond;
    const void* audio;
    float* hs;        // (B, L+1, H, Tp)
    int B, Tf, Tp, coff;   // coff: conditioning offset (seg for laplace, 1 for softmax)
    const float* gx;       // dropout mode: (B, L, 2H, Tp) sample-rate in_x products of the masked conditioning, or null
};

// ---- input layer -------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void tf_input_kernel(const FwdArgs a) {
    const SwnGeom& g = a.g;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.z;
    if (t >= a.Tp) return;
    const float* P = a.P;
    float* h0 = a.hs + (size_t)b * (g.L + 1) * g.H * a.Tp;
    const int K = g.K, H = g.H;
    if (KIND == SWN_KIND_LAPLACE) {
        const float* au = reinterpret_cast<const float*>(a.audio) + (size_t)b * (a.Tp + g.seg - 1);
        const int ai = t + g.seg - 1;                       // causal output index, cswnv_shift1.py:205
        for (int o = blockIdx.y; o < H; o += gridDim.y) {
            float acc = P[a.y.cb + o];
            for (int k = 0; k < K; ++k) {
                const int r = ai - (K - 1 - k);
                if (r >= 0) acc += fmaf(P[a.y.cv + (size_t)k * H + o], au[r], P[a.y.cc + (size_t)k * H + o]);
            }
            h0[(size_t)o * a.Tp + t] = ssign(acc);
        }
    } else {
        const int* au = reinterpret_cast<const int*>(a.audio) + (size_t)b * a.Tp;
        for (int o = blockIdx.y; o < H; o += gridDim.y) {
            float acc = P[a.y.cb + o];
            for (int k = 0; k < K; ++k) {
                const int r = t - (K - 1 - k);
                if (r >= 0) {
                    int idx = au[r] % g.Q; idx = idx < 0 ? idx + g.Q : idx;     // OneHot applies x % depth
                    acc += P[a.y.ct + ((size_t)k * g.Q + idx) * H + o];
                }
            }
            h0[(size_t)o * a.Tp + t] = ssign(acc);
        }
    }
}

// ---- one gated layer: 64 positions x (32 gate rows + 32 candidate rows) per workgroup ----------
// GEMM over Kd = K*H with the B operand = K dilated shifts of h_{l-1}; BK = 16; thread = 4 positions
// x (2 gate + 2 candidate rows); fused epilogue writes h_l.
template <int KIND>
__global__ __launch_bounds__(256) void tf_layer_kernel(const FwdArgs a, const int l, const float* __restrict__ in_mul) {
    __shared__ float As[16][SWN_MMA_PITCH];      // [k][tile row]: rows 0..31 gate, 32..63 candidate
    __shared__ float Bs[16][SWN_MMA_PITCH];      // [k][position]
    const SwnGeom& g = a.g;
    const int H = g.H, Hp = g.Hp, K = g.K, H2 = 2 * g.H, seg = g.seg;
    // XCD-aware order (1-D grid; workgroup id i runs on XCD i % 8): each XCD walks a contiguous range of time tiles, the
    // channel tiles of one time tile back to back - they read the same window of h_{l-1}, and neighbouring time
    // tiles share their tap halos in that XCD's L2
    const int ntt = (a.Tp + 63) / 64, mtl = (H + 31) / 32;
    const int chunk = (ntt * a.B + 7) / 8;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int ttl = idx / mtl, mtile = idx - ttl * mtl;
    const int gt = xcd * chunk + ttl;
    if (ttl >= chunk || gt >= ntt * a.B) return;
    const int b = gt / ntt;
    const int t0 = (gt - b * ntt) * 64, o0 = mtile * 32;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int dil = g.dil[l];
    const float* P = a.P;
    const float* hprev = a.hs + ((size_t)b * (g.L + 1) + l) * H * a.Tp;
    float* hnext = a.hs + ((size_t)b * (g.L + 1) + l + 1) * H * a.Tp;
    const float* W = P + a.y.wd + (size_t)l * H2 * K * Hp;            // [o2][k][i]
    swn_f32x4 acc[4] = {};          // M-tiles 0,1: gate rows of channels o0..o0+31 ; 2,3: their candidate rows
    const int Kd = K * Hp;
    // tile loads: division-free (tap, i) bookkeeping, fetched into registers one k-tile ahead of the MFMAs
    const int rr = tid >> 2, kq = (tid & 3) * 4;
    const int oa = o0 + (rr & 31);
    const float* wrow = W + (size_t)((rr < 32) ? oa : H + oa) * Kd;
    const int tt = tid & 63;
    const bool tok = t0 + tt < a.Tp;
    int tapB[4], iB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int kd = (tid >> 6) + 4 * q; tapB[q] = kd / Hp; iB[q] = kd - tapB[q] * Hp; }
    // operand loads through buffer resources, out-of-range offset = zero: no branch and no select touches a load, so
    // two k-tiles stay in flight under the MFMAs (a conditional load is an exec-masked branch + s_waitcnt vmcnt(0))
    const __amdgpu_buffer_rsrc_t rW = rsrc_of(W), rH = rsrc_of(hprev);
    const __amdgpu_buffer_rsrc_t rM = rsrc_of(in_mul ? in_mul + (size_t)b * H * a.Tp : hprev);
    const unsigned wbase = oa < H ? (unsigned)((size_t)((rr < 32) ? oa : H + oa) * Kd * 4) + (unsigned)(kq * 4) : SWN_OOB;
    swn_fl4 ra[2]; float rb[2][4], rm[2][4];
    auto fetch = [&](int k0, swn_fl4& qa, float (&qb)[4], float (&qm)[4]) {
        qa = bld4(rW, k0 + kq < Kd ? wbase + (unsigned)(k0 * 4) : SWN_OOB);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ts = t0 + tt - (K - 1 - tapB[q]) * dil;
            const bool ok = k0 + (tid >> 6) + 4 * q < Kd && iB[q] < H && tok && ts >= 0;
            const unsigned off = ok ? (unsigned)(((size_t)iB[q] * a.Tp + ts) * 4) : SWN_OOB;
            qb[q] = bld1(rH, off);
            qm[q] = in_mul ? bld1(rM, off) : 1.f;       // input = dropped output of layer l-1 (uniform branch)
            iB[q] += 16; while (iB[q] >= Hp) { iB[q] -= Hp; ++tapB[q]; }
        }
    };
    fetch(0, ra[0], rb[0], rm[0]);
    fetch(16, ra[1], rb[1], rm[1]);
    for (int k0 = 0; k0 < Kd; k0 += 32) {                 // a k-tile past Kd is all zeros
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            As[kq + 0][rr] = ra[u].x; As[kq + 1][rr] = ra[u].y; As[kq + 2][rr] = ra[u].z; As[kq + 3][rr] = ra[u].w;
#pragma unroll
            for (int q = 0; q < 4; ++q) Bs[(tid >> 6) + 4 * q][tt] = rb[u][q] * rm[u][q];
            __syncthreads();
            fetch(k0 + 16 * (u + 2), ra[u], rb[u], rm[u]);
            swn_mma_64x64x16(As, Bs, acc, lane, w);
            __syncthreads();
        }
    }
    // ---- epilogue: conditioning (hoisted in_x + rank-1 upsampler), gate, highway.  A lane holds, for its position
    //      t, the gate rows (M-tiles 0,1) and the candidate rows (M-tiles 2,3) of the same 8 channels.
    const int t = t0 + swn_mma_col(lane, w);
    if (t >= a.Tp) return;
    const float* condb = a.cond + (size_t)b * a.Tf * g.N;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = o0 + swn_mma_row(lane, mt, i);
            if (o >= H) continue;
            const float bz = P[a.y.bd + (size_t)l * H2 + o], bc = P[a.y.bd + (size_t)l * H2 + H + o];
            float gz, gc;
            if (a.gx) {      // dropout mode: in_x evaluated at sample rate on the masked conditioning
                const float* gr = a.gx + (((size_t)b * g.L + l) * H2) * a.Tp + t;
                gz = gr[(size_t)o * a.Tp] + P[a.y.bxr + (size_t)l * H2 + o];
                gc = gr[(size_t)(H + o) * a.Tp] + P[a.y.bxr + (size_t)l * H2 + H + o];
            } else {
                gz = P[a.y.bx + (size_t)l * H2 + o]; gc = P[a.y.bx + (size_t)l * H2 + H + o];
                for (int s = 0; s < seg; ++s) {
                    const int tt = t + s + a.coff;
                    int f = tt / g.U; const int jj = tt - f * g.U;
                    f = f < a.Tf ? f : a.Tf - 1;
                    const float wv = P[a.y.wup + jj];
                    const float* cr = condb + (size_t)f * g.N + (size_t)(l * seg + s) * H2;
                    gz = fmaf(wv, cr[o], gz);
                    gc = fmaf(wv, cr[H + o], gc);
                }
            }
            if (KIND == SWN_KIND_SOFTMAX && g.audio_in) {
                int idx = reinterpret_cast<const int*>(a.audio)[(size_t)b * a.Tp + t] % g.Q;
                idx = idx < 0 ? idx + g.Q : idx;
                const float* wa = P + a.y.wxa + ((size_t)l * g.Q + idx) * H2;
                gz += wa[o]; gc += wa[H + o];
            }
            const float z = sigm(gz * (acc[mt][i] + bz));
            const float c = tanhf(gc * (acc[2 + mt][i] + bc));
            const float hin = hprev[(size_t)o * a.Tp + t] * (in_mul ? in_mul[((size_t)b * H + o) * a.Tp + t] : 1.f);
            hnext[(size_t)o * a.Tp + t] = (1.f - z) * c + z * hin;
        }
}

// ---- Y[b][m][t] = act(sum_k W[m][k] X[b][k][t] + bias[m]) : 64x64 tile, BK=16, 4x4 per thread -----
__global__ __launch_bounds__(256) void gemm_wx_kernel(const float* __restrict__ W, int ldw,
                                                      const float* __restrict__ bias,
                                                      const float* __restrict__ X, size_t xstride_b,
                                                      float* __restrict__ Y, size_t ystride_b,
                                                      int M, int Kd, int T, int relu) {
    __shared__ float As[16][SWN_MMA_PITCH];
    __shared__ float Bs[16][SWN_MMA_PITCH];
    const int b = blockIdx.z, t0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* Xb = X + (size_t)b * xstride_b;
    float* Yb = Y + (size_t)b * ystride_b;
    swn_f32x4 acc[4] = {};
    // operands through buffer resources (out-of-range offset = zero), two k-tiles of loads in flight under the MFMAs
    const __amdgpu_buffer_rsrc_t rW = rsrc_of(W), rX = rsrc_of(Xb);
    const int rr = tid >> 2, kq = (tid & 3) * 4, tt = tid & 63, kb = tid >> 6;
    const unsigned wrow = m0 + rr < M ? (unsigned)((size_t)(m0 + rr) * ldw * 4) : SWN_OOB;
    const bool tok = t0 + tt < T;
    float ra[2][4], rb[2][4];
    auto fetch = [&](int k0, float (&qa)[4], float (&qb)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            qa[e] = bld1(rW, (wrow != SWN_OOB && k0 + kq + e < Kd) ? wrow + (unsigned)((k0 + kq + e) * 4) : SWN_OOB);
            const int kk = k0 + kb + 4 * e;
            qb[e] = bld1(rX, (kk < Kd && tok) ? (unsigned)(((size_t)kk * T + t0 + tt) * 4) : SWN_OOB);
        }
    };
    fetch(0, ra[0], rb[0]);
    fetch(16, ra[1], rb[1]);
    for (int k0 = 0; k0 < Kd; k0 += 32) {                 // a k-tile past Kd is all zeros
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { As[kq + e][rr] = ra[u][e]; Bs[kb + 4 * e][tt] = rb[u][e]; }
            __syncthreads();
            fetch(k0 + 16 * (u + 2), ra[u], rb[u]);
            swn_mma_64x64x16(As, Bs, acc, lane, w);
            __syncthreads();
        }
    }
    const int t = t0 + swn_mma_col(lane, w);
    if (t >= T) return;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + swn_mma_row(lane, mt, i);
            if (m >= M) continue;
            const float v = acc[mt][i] + bias[m];
            Yb[(size_t)m * T + t] = relu ? fmaxf(v, 0.f) : v;
        }
}

size_t r64(size_t x) { return (x + 63) & ~(size_t)63; }

// raw (B, NO, Tp) -> time-major mu / b / logb / a (+ clipped copies); one thread per (b, t)
__global__ __launch_bounds__(256) void laplace_head_kernel(const float* __restrict__ raw, int Tp, int seg, int lpc,
                                                           float* mu, float* bsc, float* logb, float* acf,
                                                           float* b_clip, float* logb_clip, int* below) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= Tp) return;
    const int NO = 2 * seg + lpc;
    const float* r = raw + (size_t)b * NO * Tp + t;
    const float FLOOR = -14.162084148244246758816564788835f;
    bool any = false;
    for (int j = 0; j < seg; ++j) {
        const size_t o = ((size_t)b * Tp + t) * seg + j;
        mu[o] = r[(size_t)j * Tp];
        const float y = r[(size_t)(seg + j) * Tp];
        const float lb = fminf(y, 0.f) - log1pf(expf(-fabsf(y)));      // logsigmoid
        logb[o] = lb;
        bsc[o] = expf(lb);
        any |= lb < FLOOR;
        if (logb_clip) { const float lc = fmaxf(lb, FLOOR); logb_clip[o] = lc; b_clip[o] = expf(lc); }
    }
    for (int k = 0; k < lpc; ++k) acf[((size_t)b * Tp + t) * lpc + k] = r[(size_t)(2 * seg + k) * Tp];
    if (any) atomicOr(below, 1);
}

}  // namespace

extern "C" int swn_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return SWN_E_NODEVICE;
    return n;
}

// work = hidden states (only when the caller does not pass hs) + skip activations + out_1 activations
extern "C" size_t swn_forward_work_floats(const swn_net_desc* d, int batch, int n_frames) {
    SwnGeom g; if (swn_make_geom(d, &g) < 0 || batch < 1 || n_frames < 1) return 0;
    const long T = (long)n_frames * g.U;
    const long Tp = g.kind == SWN_KIND_SOFTMAX ? T - 1 : T - 2 * g.seg + 1;
    if (Tp < 1) return 0;
    return r64((size_t)batch * (g.L + 1) * g.H * Tp) + r64((size_t)batch * g.S * Tp) + r64((size_t)batch * g.O1 * Tp);
}

namespace {

// masked, upsampled conditioning for the dropout mode: xm[b][c][u] = drop_x[b][c][u] * (C[b][c][f] * w_up[j] + b_up)
// with u + coff = f*U + j   (cswnv_shift1.py:193-195, dswnv.py:252-254)
__global__ __launch_bounds__(256) void xm_fwd_kernel(const float* __restrict__ C, const float* __restrict__ P, size_t wup, size_t bup,
                                                     const float* __restrict__ drop_x, float* __restrict__ xm,
                                                     int A0, int A0x, int Tf, int U, int coff, int Tx) {
    const int u = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;      // c < A0x; rows A0.. are zero padding
    if (u >= Tx) return;
    const size_t ox = ((size_t)b * A0x + c) * Tx + u;
    if (c >= A0) { xm[ox] = 0.f; return; }
    const int tt = u + coff, f = tt / U, j = tt - f * U;
    const size_t o = ((size_t)b * A0 + c) * Tx + u;
    xm[ox] = drop_x[o] * fmaf(C[((size_t)b * A0 + c) * Tf + f], P[wup + j], P[bup]);
}

// the same as bf16 rows of `pitch` elements (swn_drop_inx16: operand of the bf16-copy contraction kernels, csrc/swn_train.hip).
// A thread takes eight consecutive positions: two 16-byte mask loads, one 16-byte store (one element per thread - 4-byte loads,
// 2-byte stores - ran at 2.1 TB/s: 187 us for the 257 MB mask at the run.sh geometry).
__global__ __launch_bounds__(256) void xm_fwd16_kernel(const float* __restrict__ C, const float* __restrict__ P, size_t wup, size_t bup,
                                                       const float* __restrict__ drop_x, unsigned short* __restrict__ xm16,
                                                       int A0, int A0x, int Tf, int U, int coff, int Tx, long pitch) {
    const int u0 = (blockIdx.x * 256 + threadIdx.x) * 8, c = blockIdx.y, b = blockIdx.z;
    if (u0 >= Tx) return;
    unsigned short* dst = xm16 + ((size_t)b * A0x + c) * pitch + u0;          // pitch % 32 == 0: 16-byte aligned
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (c < A0) {
        // (a view that ends with the mask tensor: the last row's final piece may reach past it and must read zeros, not memory)
        const size_t left = ((size_t)(gridDim.z - b) * A0 - c) * Tx * sizeof(float);
        const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(drop_x + ((size_t)b * A0 + c) * Tx), 0,
                                                                            left < 0x7fffffffu ? (unsigned)left : 0x7fffffffu, 0x00020000);
        const swn_fl4 m0 = bld4(rM, (unsigned)(u0 * 4)), m1 = bld4(rM, u0 + 4 < Tx ? (unsigned)((u0 + 4) * 4) : SWN_OOB);
        const float m[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
        const float* Cr = C + ((size_t)b * A0 + c) * Tf;
        const float bu = P[bup];
        int f = (u0 + coff) / U, j = u0 + coff - f * U;
        float cf = Cr[f];
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = u0 + e < Tx ? m[e] * fmaf(cf, P[wup + j], bu) : 0.f;
            if (++j == U) { j = 0; ++f; cf = Cr[f < Tf ? f : Tf - 1]; }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) w[d] = swn_pack_bf16(v[2 * d], v[2 * d + 1]);
    }
    *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
}

}  // namespace

// relu(skip), relu(out_1) from the hidden states in `work` through the contraction kernels of the training mode (csrc/swn_train.hip)
int swn_train_head_acts(const SwnGeom& g, const float* packed, float* work, int batch, long Tp, hipStream_t st);
// gated layers of the dropout-mode forward in the mixed-precision mode (csrc/swn_train.hip)
int swn_train_layers_forward_drop(const SwnGeom& g, const SwnLayout& y, const float* packed, const void* audio, const float* gx,
                                  const float* const* drop_h, float* hs, float* a_scr, float* hmask, int B, int n_frames, int Tp,
                                  hipStream_t st);
// sample-rate in_x of every layer over the masked conditioning (csrc/swn_train.hip: generic time GEMM)
int swn_train_inx_forward(const SwnGeom& g, const SwnLayout& y, const float* packed, const float* xm, float* gx,
                          int B, int Tx, int Tp, hipStream_t st, unsigned short* wx16 = nullptr, bool g4 = false);
// GEMM-stack geometries, mixed-precision mode: the dropout-mode forward on the bf16 time-major stack (csrc/swn_stack_bf16g.hip)
int swn_bf16g_geom(const swn_net_desc* d, SwnGeom* g);
size_t swn_bf16g_weight_bytes(const SwnGeom& g);
int swn_bf16g_pack(const SwnGeom& g, const float* packed, void* wbf, hipStream_t st);
size_t swn_bf16g_work_bytes(const SwnGeom& g, int batch, long Tp);
int swn_bf16g_expand(const SwnGeom& g, const void* work, int batch, long Tp, float* fwd_work, bool hs_only, hipStream_t st);
int swn_bf16g_forward(const SwnGeom& g, const float* packed, const void* wbf, const float* cond, const void* audio,
                      int batch, int n_frames, void* work, float* out, hipStream_t st, float* a_keep,
                      const float* gx, const float* const* drop_h, unsigned short* hm16);
// BL6 class, mixed-precision mode, aux_drop the only mask that acts: the fused path (csrc/swn_stack_bf16.hip)
int swn_bl6_drop_forward(const SwnGeom& g, const float* packed, const float* C, const float* audio, const float* drop_x,
                         int batch, int n_frames, void* work, float* out, hipStream_t st);

namespace {

// does the dropout-mode forward of the mixed-precision mode run on the bf16 time-major GEMM stack?  (the geometry class of
// csrc/swn_stack_bf16g.hip, its 32-bit operand offsets, a sequence long enough for the bf16-copy contractions)
bool drop_g16(const swn_net_desc* d, int batch, long Tp) { return swn_drop_g16(d, batch, Tp); }

}  // namespace

bool swn_drop_g16(const swn_net_desc* d, int batch, long Tp) {
    SwnGeom g;
    if (swn_bf16g_geom(d, &g) != SWN_OK || Tp < 256) return false;
    const size_t lstride = (size_t)batch * Tp * g.H;
    return (size_t)g.L * lstride * 2 < (1ull << 31) && (size_t)batch * Tp * (g.S > g.O1 ? g.S : g.O1) * 2 < (1ull << 31);
}

namespace {

int forward_impl(const swn_net_desc* d, const float* packed, const float* cond, const float* fe_work, const void* audio,
                 int batch, int n_frames, const float* drop_x, const float* const* drop_h, float* work, float* out,
                 float* hs, void* stream_, const char* where) {
    FwdArgs a;
    int rc = swn_make_geom(d, &a.g);
    if (rc < 0) return rc;
    const SwnGeom& g = a.g;
    const bool drop = drop_x != nullptr;
    if (!packed || (!drop && !cond) || (drop && (!fe_work || !drop_h)) || !audio || !work || !out || batch < 1 ||
        batch > 65535 || n_frames < 1) return SWN_E_BADARG;
    swn_make_layout(&a.g, &a.y);
    const long T = (long)n_frames * g.U;
    const long Tp = g.kind == SWN_KIND_SOFTMAX ? T - 1 : T - 2 * g.seg + 1;
    if (Tp < 1) return SWN_E_BADARG;
    {   // the tiled kernels address one utterance's operands with 32-bit byte offsets
        size_t widest = (size_t)(g.L + 1) * g.H;
        if ((size_t)g.S > widest) widest = g.S;
        if ((size_t)g.O1 > widest) widest = g.O1;
        if (widest * Tp * sizeof(float) >= (1ull << 31)) {
            swn_set_error_detail(where, "one utterance's activations exceed 2 GiB: split the chunk");
            return SWN_E_UNSUPPORTED;
        }
    }
    hipStream_t st = (hipStream_t)stream_;
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls; only our launches are reported
    if (drop && !hs && swn_call_mode() == SWN_PRECISION_BF16 && swn_bl6_drop_supported(g, batch, Tp, n_frames, drop_h)) {
        size_t fe_off = (size_t)g.n_aux;                         // frame-rate activations: scaled | conv_aux layers
        for (int i = 0; i + 1 < g.auxl; ++i) fe_off += g.aux_cout[i];
        return swn_bl6_drop_forward(g, packed, fe_work + fe_off * (size_t)batch * n_frames, reinterpret_cast<const float*>(audio),
                                    drop_x, batch, n_frames, work, out, st);
    }
    const size_t hs_floats = r64((size_t)batch * (g.L + 1) * g.H * Tp);
    float* hbuf = hs ? hs : work;
    float* skipb = work + hs_floats;
    float* o1b = skipb + r64((size_t)batch * g.S * Tp);
    a.P = packed; a.cond = cond; a.audio = audio; a.hs = hbuf; a.B = batch; a.Tf = n_frames; a.Tp = (int)Tp;
    a.coff = g.kind == SWN_KIND_SOFTMAX ? 1 : g.seg;
    a.gx = nullptr;
    if (drop) {
        // work tail: xm (B, A0x, Tx) | gx (B, L, 2H, Tp)
        const int Tx = (int)(T - a.coff);
        float* xm = o1b + r64((size_t)batch * g.O1 * Tp);
        float* gx = xm + r64((size_t)batch * swn_a0x(&g) * Tx);
        size_t fe_off = (size_t)g.n_aux;                         // frame-rate activations: scaled | conv_aux layers
        for (int i = 0; i + 1 < g.auxl; ++i) fe_off += g.aux_cout[i];
        const float* C = fe_work + fe_off * (size_t)batch * n_frames;
        unsigned short* wx16 = nullptr;
        if (swn_call_mode() == SWN_PRECISION_BF16 && swn_drop_inx16(&g, Tp)) {
            // xm as bf16 rows in the same section (the backward of the same mode reads them there), the bf16 in_x matrix at the
            // end of the work buffer
            wx16 = reinterpret_cast<unsigned short*>(gx + r64((size_t)batch * g.L * 2 * g.H * Tp) +
                                                     (size_t)g.L * r64((size_t)batch * 2 * g.H * Tp) + r64((size_t)batch * g.H * Tp));
            hipLaunchKernelGGL(xm_fwd16_kernel, dim3((Tx + 2047) / 2048, swn_a0x(&g), batch), dim3(256), 0, st, C, packed, a.y.wup, a.y.bup,
                               drop_x, reinterpret_cast<unsigned short*>(xm), g.A0, swn_a0x(&g), n_frames, g.U, a.coff, Tx, swn_pitch16(Tx));
        } else
        hipLaunchKernelGGL(xm_fwd_kernel, dim3((Tx + 255) / 256, swn_a0x(&g), batch), dim3(256), 0, st, C, packed, a.y.wup, a.y.bup,
                           drop_x, xm, g.A0, swn_a0x(&g), n_frames, g.U, a.coff, Tx);
        // (the GEMM stack reads gx in the G4 layout, swn_geom.hpp; it needs the bf16-copy product kernel, i.e. wx16)
        const bool g16 = !hs && swn_call_mode() == SWN_PRECISION_BF16 && drop_g16(d, batch, Tp) && wx16;
        rc = swn_train_inx_forward(g, a.y, packed, xm, gx, batch, Tx, (int)Tp, st, wx16, g16);
        if (rc < 0) return rc;
        a.gx = gx;
    }
    if (drop && !hs && swn_call_mode() == SWN_PRECISION_BF16 && drop_g16(d, batch, Tp) && swn_drop_inx16(&g, Tp)) {
        // GEMM-stack geometries: input layer, gated layers (gate epilogue on the sample-rate in_x rows, a dropped level read through
        // its masked copy, pre-activations kept for the backward) and head on the bf16 time-major stack of the forward without
        // dropout; what the backward reads in fp32 (hidden states, relu(skip), relu(out_1)) is expanded from it.  Replaces one
        // fp32-operand time GEMM + one element-wise gate launch per layer (475 -> ~330 us per layer at the run.sh geometry).
        float* a_scr = const_cast<float*>(a.gx) + r64((size_t)batch * g.L * 2 * g.H * Tp);
        float* tail = a_scr + (size_t)g.L * r64((size_t)batch * 2 * g.H * Tp) + r64((size_t)batch * g.H * Tp) +
                      r64((size_t)g.L * 2 * g.H * swn_a0x(&g) / 2 + 1);
        void* work16 = tail;
        unsigned short* hm16 = reinterpret_cast<unsigned short*>(tail + r64((swn_bf16g_work_bytes(g, batch, Tp) + 3) / 4));
        void* wbf = reinterpret_cast<float*>(hm16) + r64((size_t)batch * Tp * g.H / 2 + 1);
        rc = swn_bf16g_pack(g, packed, wbf, st);
        if (rc < 0) return rc;
        rc = swn_bf16g_forward(g, packed, wbf, nullptr, audio, batch, n_frames, work16, out, st, a_scr, a.gx, drop_h, hm16);
        if (rc < 0) return rc;
        rc = swn_bf16g_expand(g, work16, batch, Tp, work, false, st);
        return rc < 0 ? rc : swn_launch_status(where);
    }
    const int tb64 = (int)((Tp + 63) / 64);
    {
        dim3 grid((unsigned)((Tp + 255) / 256), g.H < 16 ? g.H : 16, batch);
        if (g.kind == SWN_KIND_LAPLACE) hipLaunchKernelGGL(tf_input_kernel<SWN_KIND_LAPLACE>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(tf_input_kernel<SWN_KIND_SOFTMAX>, grid, dim3(256), 0, st, a);
    }
    if (drop && swn_call_mode() == SWN_PRECISION_BF16 && swn_drop_bf16_forward(&g)) {
        // mixed-precision mode: bf16-operand GEMM + element-wise gate per layer; after gx in the work buffer: the gate
        // pre-activations of every layer (B, 2H, Tp) x L - kept, the backward of the same mode reads them instead of
        // recomputing - | masked input of one layer (B, H, Tp)
        float* a_scr = const_cast<float*>(a.gx) + r64((size_t)batch * g.L * 2 * g.H * Tp);
        float* hmask = a_scr + (size_t)g.L * r64((size_t)batch * 2 * g.H * Tp);
        rc = swn_train_layers_forward_drop(g, a.y, packed, audio, a.gx, drop_h, hbuf, a_scr, hmask, batch, n_frames, (int)Tp, st);
        if (rc < 0) return rc;
    } else
    for (int l = 0; l < g.L; ++l) {
        const dim3 grid((unsigned)(8 * ((tb64 * batch + 7) / 8) * ((g.H + 31) / 32)));
        const float* in_mul = (drop && l > 0) ? drop_h[l - 1] : nullptr;      // layer l-1's output was dropped
        if (g.kind == SWN_KIND_LAPLACE) hipLaunchKernelGGL(tf_layer_kernel<SWN_KIND_LAPLACE>, grid, dim3(256), 0, st, a, l, in_mul);
        else hipLaunchKernelGGL(tf_layer_kernel<SWN_KIND_SOFTMAX>, grid, dim3(256), 0, st, a, l, in_mul);
    }
    const size_t hstride = (size_t)(g.L + 1) * g.H * Tp;
    // skip: one GEMM over the L concatenated (undropped) hidden states (requires Hp == H, i.e. H % 4 == 0)
    if (g.Hp != g.H) return SWN_E_UNSUPPORTED;
    if (drop && swn_call_mode() == SWN_PRECISION_BF16 && !hs && swn_drop_bf16_forward(&g)) {
        // mixed-precision mode: the two wide 1x1 layers as bf16-operand time GEMMs with bias + relu epilogues (0.5 ms each as
        // exact-fp32 gemm_wx at the run.sh geometry); out_2 (<= 16 rows) stays below
        rc = swn_train_head_acts(g, packed, work, batch, Tp, st);
        if (rc < 0) return rc;
    } else {
    hipLaunchKernelGGL(gemm_wx_kernel, dim3(tb64, (g.S + 63) / 64, batch), dim3(256), 0, st,
                       packed + a.y.wsk, g.L * g.Hp, packed + a.y.bsk, hbuf + (size_t)g.H * Tp, hstride,
                       skipb, (size_t)g.S * Tp, g.S, g.L * g.H, (int)Tp, 1);
    hipLaunchKernelGGL(gemm_wx_kernel, dim3(tb64, (g.O1 + 63) / 64, batch), dim3(256), 0, st,
                       packed + a.y.w1, g.Sp, packed + a.y.b1, skipb, (size_t)g.S * Tp,
                       o1b, (size_t)g.O1 * Tp, g.O1, g.S, (int)Tp, 1);
    }
    hipLaunchKernelGGL(gemm_wx_kernel, dim3(tb64, (g.NO + 63) / 64, batch), dim3(256), 0, st,
                       packed + a.y.w2, g.O1p, packed + a.y.b2, o1b, (size_t)g.O1 * Tp,
                       out, (size_t)g.NO * Tp, g.NO, g.O1, (int)Tp, 0);
    return swn_launch_status(where);
}

}  // namespace

extern "C" int swn_forward(const swn_net_desc* d, const float* packed, const float* cond, const void* audio,
                           int batch, int n_frames, float* work, float* out, float* hs, void* stream_) {
    return forward_impl(d, packed, cond, nullptr, audio, batch, n_frames, nullptr, nullptr, work, out, hs, stream_, "swn_forward");
}

extern "C" int swn_drop_fused_path(const swn_net_desc* d, int batch, int n_frames, const float* const* drop_h) {
    SwnGeom g;
    if (swn_make_geom(d, &g) < 0 || batch < 1 || n_frames < 1 || g.kind != SWN_KIND_LAPLACE) return 0;
    const long Tp = (long)n_frames * g.U - 2 * g.seg + 1;
    return Tp >= 1 && swn_bl6_drop_supported(g, batch, Tp, n_frames, drop_h) ? 1 : 0;
}

extern "C" size_t swn_forward_drop_work_floats(const swn_net_desc* d, int batch, int n_frames) {
    SwnGeom g; if (swn_make_geom(d, &g) < 0 || batch < 1 || n_frames < 1) return 0;
    const size_t base = swn_forward_work_floats(d, batch, n_frames);
    if (!base) return 0;
    const long T = (long)n_frames * g.U;
    const int coff = g.kind == SWN_KIND_SOFTMAX ? 1 : g.seg;
    const long Tp = g.kind == SWN_KIND_SOFTMAX ? T - 1 : T - 2 * g.seg + 1;
    // xm | gx | (mixed-precision forward) gate pre-activations of every layer | masked input of one layer
    const size_t chain = base + r64((size_t)batch * swn_a0x(&g) * (T - coff)) + r64((size_t)batch * g.L * 2 * g.H * Tp) +
                         (size_t)g.L * r64((size_t)batch * 2 * g.H * Tp) + r64((size_t)batch * g.H * Tp) +
                         r64((size_t)g.L * 2 * g.H * swn_a0x(&g) / 2 + 1) +       // bf16 in_x matrix (swn_drop_inx16)
                         (drop_g16(d, batch, Tp) ? r64((swn_bf16g_work_bytes(g, batch, Tp) + 3) / 4) + r64((size_t)batch * Tp * g.H / 2 + 1) +
                                                       r64((swn_bf16g_weight_bytes(g) + 3) / 4) : 0);    // bf16 stack | masked level | bf16 weights
    const size_t fused = g.bl6 ? (swn_bl6_drop_layout(g, batch, Tp).total + 3) / 4 : 0;      // the fused BL6 path's own layout
    return chain > fused ? chain : fused;
}

extern "C" int swn_forward_drop(const swn_net_desc* d, const float* packed, const float* fe_work, const void* audio,
                                int batch, int n_frames, const float* drop_x, const float* const* drop_h,
                                float* work, float* out, float* hs, int precision, void* stream_) {
    if (!drop_x || !swn_precision_ok(precision)) return SWN_E_BADARG;
    SwnModeScope mode(precision);
    return forward_impl(d, packed, nullptr, fe_work, audio, batch, n_frames, drop_x, drop_h, work, out, hs, stream_,
                        "swn_forward_drop");
}

extern "C" int swn_laplace_head(const swn_net_desc* d, const float* out, int batch, int tp, float* mu, float* b,
                                float* logb, float* a, float* b_clip, float* logb_clip, int32_t* below,
                                void* stream_) {
    SwnGeom g; int rc = swn_make_geom(d, &g);
    if (rc < 0) return rc;
    if (g.kind != SWN_KIND_LAPLACE) return SWN_E_BADDESC;
    if (!out || !mu || !b || !logb || !below || (g.lpc > 0 && !a) || batch < 1 || batch > 65535 || tp < 1 ||
        ((b_clip == nullptr) != (logb_clip == nullptr)))
        return SWN_E_BADARG;
    (void)hipGetLastError();
    hipLaunchKernelGGL(laplace_head_kernel, dim3((tp + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream_,
                       out, tp, g.seg, g.lpc, mu, b, logb, a, b_clip, logb_clip, below);
    return swn_launch_status("swn_forward");
}
