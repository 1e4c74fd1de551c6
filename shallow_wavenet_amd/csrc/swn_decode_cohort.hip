// Cohort decode: many utterances of a LARGE geometry (REF6: H=192/256, K=7) generated in lock step,
// LANES = UTTERANCES.
//
// The stepped decode (swn_decode_stepped.hip) gives every utterance its own workgroups, so B utterances read
// the 15-24 MB of weights B times per generated sample from the Infinity Cache (B=64: ~6 TB/s of it, 162 us per
// step).  Here the state of up to 64 utterances is stored structure-of-arrays with the utterance index fastest
//     state[(cohort * stride + element) * 64 + lane]
// and a wave owns one output row (pair) for all 64 utterances of a cohort: the weight of an input is ONE scalar
// (wave-uniform) operand multiplied into 64 different activations loaded with one coalesced 256-byte load.  The
// weights cross the chip once per step and cohort, there are no cross-lane reductions at all, and the gate /
// sampling epilogues are plain per-lane code.  A workgroup owns a few output rows; its 16 waves split the inputs
// (each activation load feeds all the workgroup's rows) and the partial sums meet in LDS.
// Launch chain per step (stream order = the dependency chain, as in the stepped decode):
//     co_in | L x co_layer | co_rowvec x 3 (skip, out_1, out_2) | co_tail            = L + 5 launches
// Math, ring layout and sampling are those of the stepped / generic kernels (cswnv_shift1.py:348-402,
// dswnv.py:338-374); only the order of the fp32 sums over the inputs differs (16 contiguous slices).
#include <hip/hip_runtime.h>
#include "swn_geom.hpp"
#include "swn_noise.hpp"
#include <type_traits>

namespace {

constexpr int CO = 64;          // utterances per cohort = lanes of a wave
typedef float cf4 __attribute__((ext_vector_type(4)));

// branch-free loads (see swn_decode_stepped.hip): out-of-range offset = zero, no exec-masked branch, no drained queue
constexpr unsigned CO_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t co_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float co_ld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}

struct CoArgs {
    SwnGeom g;
    SwnLayout y;
    const float* P; const float* cond; SwnNoise nz; const void* forced; const void* seed;
    float* state; void* out; float* heads;
    int B, Tf, n_steps, n_pro, WN;
    int ring_off[SWN_MAXL], ring_len[SWN_MAXL];
    int o_hcat, o_skip, o_o1, o_o2, o_hist, stride;             // element offsets inside a cohort
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ int pmod(int r, int m) { int t = r % m; return t < 0 ? t + m : t; }

struct Iter { bool gen; int i, np, q0; };
__device__ __forceinline__ Iter iter_of(const CoArgs& a, int it) {
    Iter r; r.gen = it >= a.n_pro; r.i = it - a.n_pro; r.np = r.gen ? a.g.seg : 1;
    r.q0 = r.gen ? a.g.rf + 1 - a.g.seg + r.i * a.g.seg : it;
    return r;
}

// ---- input layer: one wave per channel, lane = utterance ----------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void co_in_kernel(const CoArgs a, const int it) {
    const SwnGeom& g = a.g;
    const int lane = threadIdx.x & 63, c = blockIdx.y;
    const int o = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (o >= g.H) return;
    const bool live = c * CO + lane < a.B;
    float* st = a.state + (size_t)c * a.stride * CO + lane;
    const float* P = a.P;
    const Iter r = iter_of(a, it);
    const int H = g.H, K = g.K, seg = g.seg, WN = a.WN;
    for (int j = 0; j < r.np; ++j) {
        const int q = r.q0 + j;
        float acc = P[a.y.cb + o];
        for (int k = 0; k < K; ++k) {
            const int rr = q - (K - 1 - k);
            if (KIND == SWN_KIND_LAPLACE) {
                if (rr >= -(seg - 1)) {
                    const int qe = r.gen ? g.rf + r.i * seg : g.rf;
                    const float sv = r.gen ? st[(size_t)(a.o_hist + rr - qe + WN - 1) * CO] : 0.f;
                    acc += fmaf(P[a.y.cv + (size_t)k * H + o], sv, P[a.y.cc + (size_t)k * H + o]);
                }
            } else if (rr >= 0) {
                const int qe = r.gen ? g.rf + r.i : g.rf;
                const int idx = r.gen ? __builtin_bit_cast(int, st[(size_t)(a.o_hist + rr - qe + WN - 1) * CO]) : g.Q / 2;
                acc += P[a.y.ct + ((size_t)k * g.Q + idx) * H + o];
            }
        }
        if (live) st[(size_t)(a.ring_off[0] + pmod(q, a.ring_len[0]) * g.Hp + o) * CO] = acc / (1.f + fabsf(acc));
    }
}

// ---- one gated layer.  Workgroup = PW channel pairs (gate row o, candidate row H+o), 16 waves; wave s owns the
//      input-channel slice [s*Hp/16, (s+1)*Hp/16) of every tap for ALL PW pairs: an activation (one coalesced
//      256-byte load for the 64 utterances) feeds 2*PW FMAs whose weights are wave-uniform scalars, so neither
//      the activations (re-read once per workgroup) nor the weights (once per cohort) are streamed redundantly.
constexpr int PW = 4;           // channel pairs per workgroup
constexpr int KS = 16;          // input slices = waves per workgroup

// stage `rows` weight rows (each `len` floats, global row stride `ld`) into LDS as wsh[row][len], all 1024 threads:
// every thread issues ALL its (at most NSTG) 16-byte loads before the first LDS store
constexpr int NSTG = 5;         // 1024 threads x 5 x 4 floats = 20 480 floats >= 8 rows x 2 304 (the longest row: K*H of H=256... = 1 792)
__device__ __forceinline__ void stage_rows(float* wsh, const float* src, const int* rowidx, int rows, int len, int ld) {
    const __amdgpu_buffer_rsrc_t rs = co_rsrc(src);
    const int n4 = len >> 2, tot = rows * n4;
    float4 v[NSTG];
#pragma unroll
    for (int u = 0; u < NSTG; ++u) {
        const int e = threadIdx.x + 1024 * u;
        const int ec = e < tot ? e : 0;
        const int r = ec / n4, k4 = ec - r * n4;
        v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
            rs, e < tot ? (unsigned)(((size_t)rowidx[r] * ld + 4 * k4) * 4) : CO_OOB, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < NSTG; ++u) {
        const int e = threadIdx.x + 1024 * u;
        if (e < tot) { const int r = e / n4, k4 = e - r * n4; *reinterpret_cast<float4*>(wsh + (size_t)r * len + 4 * k4) = v[u]; }
    }
}

// The kernels are latency chains if written naively (a memory round trip per group of inputs), so each one issues
// ALL its loads up front: the workgroup's weight rows go to LDS with one cooperative copy, every wave fetches the
// activations of its whole input slice into registers, and only then the FMAs run (weights read back from LDS as
// broadcasts).  Two memory latencies per launch instead of ~28.
template <int KIND, int IPWC, int TP>      // IPWC: inputs per slice and tap, TP: taps whose activations are in flight at once
__global__ __launch_bounds__(1024) void co_layer_kernel(const CoArgs a, const int l, const int it) {
    extern __shared__ __attribute__((aligned(16))) float lsh[];     // weights [2*PW][K*Hp], later the partial sums
    __shared__ int rowidx[2 * PW];
    const SwnGeom& g = a.g;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int o0 = blockIdx.x * PW, c = blockIdx.y;
    const int H = g.H, Hp = g.Hp, K = g.K, H2 = 2 * g.H, seg = g.seg, KH = K * Hp;
    const int b = c * CO + lane;
    const bool live = b < a.B;
    float* st = a.state + (size_t)c * a.stride * CO + lane;
    const __amdgpu_buffer_rsrc_t rSt = co_rsrc(a.state + (size_t)c * a.stride * CO);      // one cohort: < 2 GiB
    const float* P = a.P;
    const int i0 = w * IPWC;
    const int dil = g.dil[l], R = a.ring_len[l];
    const Iter r = iter_of(a, it);
    if (threadIdx.x < 2 * PW) {
        const int u = threadIdx.x, oo = o0 + (u & (PW - 1));
        rowidx[u] = oo < H ? (u < PW ? 0 : H) + oo : 0;
    }
    __syncthreads();
    stage_rows(lsh, P + a.y.wd + (size_t)l * H2 * KH, rowidx, 2 * PW, KH, KH);
    for (int j = 0; j < r.np; ++j) {
        const int q = r.q0 + j;
        // epilogue operands of the finishing waves first: their (gathered) loads fly under everything else
        const int o = o0 + (w < PW ? w : 0);
        const bool fin = w < PW && o < H;
        float gz = 0.f, gc = 0.f, hp = 0.f;
        if (fin) {
            gz = P[a.y.bx + (size_t)l * H2 + o]; gc = P[a.y.bx + (size_t)l * H2 + H + o];
            hp = st[(size_t)(a.ring_off[l] + pmod(q, R) * Hp + o) * CO];
            if (live) {
                const float* condb = a.cond + (size_t)b * a.Tf * g.N;
                for (int s = 0; s < seg; ++s) {
                    int tt = q + s - g.rf; tt = tt < 0 ? 0 : tt;
                    int f = tt / g.U; const int jj = tt - f * g.U;
                    f = f < a.Tf ? f : a.Tf - 1;
                    const float wu = P[a.y.wup + jj];
                    const float* cr = condb + (size_t)f * g.N + (size_t)(l * seg + s) * H2;
                    gz = fmaf(wu, cr[o], gz); gc = fmaf(wu, cr[H + o], gc);
                }
                if (KIND == SWN_KIND_SOFTMAX && g.audio_in) {
                    const int qe = r.gen ? g.rf + r.i : g.rf;
                    const int idx = r.gen ? __builtin_bit_cast(int, st[(size_t)(a.o_hist + q - qe + a.WN - 1) * CO]) : g.Q / 2;
                    const float* wa = P + a.y.wxa + ((size_t)l * g.Q + idx) * H2;
                    gz += wa[o]; gc += wa[H + o];
                }
            }
        }
        float acc[2 * PW];
#pragma unroll
        for (int u = 0; u < 2 * PW; ++u) acc[u] = 0.f;
        for (int t0 = 0; t0 < K; t0 += TP) {
            float x[TP][IPWC];
#pragma unroll
            for (int tt = 0; tt < TP; ++tt) {
                const int tap = t0 + tt;
                const unsigned xb = (unsigned)(((size_t)(a.ring_off[l] + pmod(q - (K - 1 - (tap < K ? tap : 0)) * dil, R) * Hp + i0) * CO + lane) * 4);
#pragma unroll
                for (int e = 0; e < IPWC; ++e) x[tt][e] = co_ld1(rSt, (tap < K && i0 + e < Hp) ? xb + (unsigned)(e * CO * 4) : CO_OOB);
            }
            if (t0 == 0 && j == 0) __syncthreads();              // the staged weights are in LDS
#pragma unroll
            for (int tt = 0; tt < TP; ++tt) {
                const int tap = t0 + tt;
                if (tap < K && i0 < Hp) {
#pragma unroll
                    for (int e = 0; e < IPWC; e += 4)
#pragma unroll
                        for (int u = 0; u < 2 * PW; ++u) {
                            const float4 wv = *reinterpret_cast<const float4*>(__builtin_assume_aligned(lsh + (size_t)u * KH + tap * Hp + ((i0 + e < Hp) ? i0 + e : 0), 16));
                            acc[u] = fmaf(wv.x, x[tt][e], acc[u]); acc[u] = fmaf(wv.y, x[tt][e + 1], acc[u]);
                            acc[u] = fmaf(wv.z, x[tt][e + 2], acc[u]); acc[u] = fmaf(wv.w, x[tt][e + 3], acc[u]);
                        }
                }
            }
        }
        // partial sums live behind the weights (which later positions still need)
        float* part = lsh + (size_t)2 * PW * KH;
#pragma unroll
        for (int u = 0; u < 2 * PW; ++u) part[((size_t)w * 2 * PW + u) * CO + lane] = acc[u];
        __syncthreads();
        if (fin) {
            float sz = 0.f, sc = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) { sz += part[((size_t)s2 * 2 * PW + w) * CO + lane]; sc += part[((size_t)s2 * 2 * PW + PW + w) * CO + lane]; }
            const float z = sigm(gz * (sz + P[a.y.bd + (size_t)l * H2 + o]));
            const float cd = tanhf(gc * (sc + P[a.y.bd + (size_t)l * H2 + H + o]));
            const float hn = (1.f - z) * cd + z * hp;
            if (l + 1 < g.L) st[(size_t)(a.ring_off[l + 1] + pmod(q, a.ring_len[l + 1]) * Hp + o) * CO] = hn;
            if (j == r.np - 1) st[(size_t)(a.o_hcat + l * Hp + o) * CO] = hn;
        }
        __syncthreads();
    }
}

// ---- y[row][b] = act(bias[row] + W[row][:] . x[:][b]): workgroup = RW rows, 16 waves split the inputs; same
//      all-loads-first structure (weights to LDS, the wave's activations to registers)
constexpr int RW = 8;
constexpr int RV_X = 10;        // activations per wave held in registers per pass (80 inputs per pass and wave)

__global__ __launch_bounds__(1024) void co_rowvec_kernel(const CoArgs a, size_t w_off, int ldw, size_t b_off, int rows, int ni,
                                                         int x_off, int y_off, int relu) {
    extern __shared__ __attribute__((aligned(16))) float lsh[];     // weights [RW][ni] | partial sums [KS][RW][CO]
    __shared__ int rowidx[RW];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = blockIdx.x * RW, c = blockIdx.y;
    float* st = a.state + (size_t)c * a.stride * CO + lane;
    if (threadIdx.x < RW) rowidx[threadIdx.x] = r0 + (int)threadIdx.x < rows ? r0 + (int)threadIdx.x : 0;
    __syncthreads();
    stage_rows(lsh, a.P + w_off, rowidx, RW, ni, ldw);
    const int ipw = ((ni + KS - 1) / KS + 3) & ~3, i0 = w * ipw, i1 = (i0 + ipw < ni) ? i0 + ipw : ni;
    const __amdgpu_buffer_rsrc_t rSt = co_rsrc(a.state + (size_t)c * a.stride * CO);
    float acc[RW];
#pragma unroll
    for (int u = 0; u < RW; ++u) acc[u] = 0.f;
    bool first = true;
    for (int ib = i0; ib < i1 || first; ib += 4 * RV_X) {
        float x[4 * RV_X];
#pragma unroll
        for (int e = 0; e < 4 * RV_X; ++e) x[e] = co_ld1(rSt, (ib + e < i1) ? (unsigned)(((size_t)(x_off + ib + e) * CO + lane) * 4) : CO_OOB);
        if (first) { __syncthreads(); first = false; }
#pragma unroll
        for (int e = 0; e < 4 * RV_X; e += 4) {
            if (ib + e < i1) {
#pragma unroll
                for (int u = 0; u < RW; ++u) {
                    const float4 wv = *reinterpret_cast<const float4*>(__builtin_assume_aligned(lsh + (size_t)u * ni + ib + e, 16));
                    acc[u] = fmaf(wv.x, x[e], acc[u]); acc[u] = fmaf(wv.y, x[e + 1], acc[u]);
                    acc[u] = fmaf(wv.z, x[e + 2], acc[u]); acc[u] = fmaf(wv.w, x[e + 3], acc[u]);
                }
            }
        }
    }
    float* part = lsh + (size_t)RW * ni;
#pragma unroll
    for (int u = 0; u < RW; ++u) part[((size_t)w * RW + u) * CO + lane] = acc[u];
    __syncthreads();
    if (w < RW && r0 + w < rows) {
        float v = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) v += part[((size_t)s2 * RW + w) * CO + lane];
        v += a.P[b_off + r0 + w];
        st[(size_t)(y_off + r0 + w) * CO] = relu ? fmaxf(v, 0.f) : v;
    }
}


// ---- exact-fp32 matrix-core form of the two contractions above (round 2) -------------------------------------------
// A layer of a 64-utterance cohort is a real GEMM, D[2H rows][64 utterances] = Wd[2H][K*H] . X[K*H][64], and the
// utterance-minor state IS its B operand: a ring slot is a contiguous [Hp][64] block.  v_mfma_f32_16x16x4_f32 multiplies
// fp32 operands like an fmaf chain (parity kept); its operands come per lane straight from memory - A: 16 bytes of a
// weight row (the four k of one 16-wide k group: the k order inside a group is permuted identically for A and B,
// which a sum does not see), B: one float of four consecutive state rows - so nothing is broadcast out of LDS, which
// is what bounded the VALU kernels above (16 waves x 168 ds_read_b128 per launch).
//   workgroup = 32 rows (gate rows o0..o0+15 and their candidate rows | 32 consecutive rows) x 16 utterances x all of K;
//   its 8 waves split K (16-wide k groups w, w+8, ...), every wave issues ALL its operand loads before its first MFMA
//   (one memory latency per launch), the eight partial tiles meet in LDS and are summed in wave order (deterministic);
//   the epilogue's operands (conditioning, biases, highway input) are requested at kernel entry and arrive meanwhile.
typedef float co_f32x4 __attribute__((ext_vector_type(4)));
constexpr int GW = 8;             // waves per workgroup = K slices
constexpr int GMAX = 11;          // 16-wide k groups per wave held in registers (K*Hp <= 16 * 8 * 11 = 1408 per pass)

struct CoGemm {
    size_t w_off; int ldw;        // A[row][k] = P[w_off + row * ldw + k]
    int rows;                     // rows of the product
    int split;                    // > 0: row group g = rows g*16.. and split + g*16.. (gate | candidate); 0: rows g*32 .. g*32+31
    int ktot;                     // K (multiple of 16)
    int ring;                     // 1: k = tap * Hp + i addresses layer l's ring; 0: k addresses x_off + k
    int x_off, y_off, relu;
    size_t b_off;
};

template <int KIND, int EPI>      // EPI 0: gated-layer epilogue, 1: bias (+ relu) rows
__global__ __launch_bounds__(64 * GW) void co_gemm_kernel(const CoArgs a, const CoGemm d, const int l, const int it) {
    __shared__ __attribute__((aligned(16))) float red[GW][2][64][4];
    const SwnGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rc = lane & 15, kq = lane >> 4;
    const int grp = blockIdx.x >> 2, cb = blockIdx.x & 3, c = blockIdx.y;
    if (c * CO + cb * 16 >= a.B) return;                       // no utterance in this column block
    const int col = 16 * cb + rc;
    const int H = g.H, Hp = g.Hp, K = g.K, H2 = 2 * g.H, seg = g.seg;
    const float* P = a.P;
    const __amdgpu_buffer_rsrc_t rP = co_rsrc(P);
    const __amdgpu_buffer_rsrc_t rSt = co_rsrc(a.state + (size_t)c * a.stride * CO);
    const Iter r = iter_of(a, it);
    const int np = EPI == 0 ? r.np : 1;
    const int lim = d.split ? d.split : d.rows;
    const int ra = (d.split ? grp * 16 : grp * 32) + rc, rb = d.split ? d.split + grp * 16 + rc : grp * 32 + 16 + rc;
    const bool oka = ra < lim, okb = d.split ? oka : rb < d.rows;
    const int ngroups = d.ktot >> 4;
    // epilogue role of this thread (threads 0..255 / 0..511): one output element each
    const int e_row = EPI == 0 ? (tid >> 4) & 15 : tid >> 4, e_col = 16 * cb + (tid & 15);
    const int e_b = c * CO + e_col;
    const bool e_on = EPI == 0 ? tid < 256 && grp * 16 + e_row < H : grp * 32 + e_row < d.rows;
    float* st = a.state + (size_t)c * a.stride * CO;
    for (int j = 0; j < np; ++j) {
        const int q = r.q0 + (EPI == 0 ? j : r.np - 1);
        // ---- epilogue operands first: their latency hides under the operand stream
        float e_gz = 0.f, e_gc = 0.f, e_bz = 0.f, e_bc = 0.f, e_hp = 0.f;
        if (e_on) {
            if (EPI == 0) {
                const int o = grp * 16 + e_row;
                e_gz = P[a.y.bx + (size_t)l * H2 + o]; e_gc = P[a.y.bx + (size_t)l * H2 + H + o];
                e_bz = P[a.y.bd + (size_t)l * H2 + o]; e_bc = P[a.y.bd + (size_t)l * H2 + H + o];
                e_hp = st[(size_t)(a.ring_off[l] + pmod(q, a.ring_len[l]) * Hp + o) * CO + e_col];
                if (e_b < a.B) {
                    const float* condb = a.cond + (size_t)e_b * a.Tf * g.N;
                    for (int s = 0; s < seg; ++s) {
                        int tt = q + s - g.rf; tt = tt < 0 ? 0 : tt;
                        int f = tt / g.U; const int jj = tt - f * g.U;
                        f = f < a.Tf ? f : a.Tf - 1;
                        const float wu = P[a.y.wup + jj];
                        const float* cr = condb + (size_t)f * g.N + (size_t)(l * seg + s) * H2;
                        e_gz = fmaf(wu, cr[o], e_gz); e_gc = fmaf(wu, cr[H + o], e_gc);
                    }
                }
            } else {
                e_bz = P[d.b_off + grp * 32 + e_row];
            }
        }
        co_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int g0 = w; g0 < ngroups; g0 += GW * GMAX) {      // REF6 Laplace: one pass
            cf4 a0[GMAX], a1[GMAX];
            float bv[GMAX][4];
#pragma unroll
            for (int u = 0; u < GMAX; ++u) {
                const int gi = g0 + u * GW;
                const bool on = gi < ngroups;
                const int k = 16 * gi + 4 * kq;                                  // this lane's four k: k .. k+3
                a0[u] = __builtin_bit_cast(cf4, __builtin_amdgcn_raw_buffer_load_b128(rP, (on && oka) ? (unsigned)((d.w_off + (size_t)ra * d.ldw + k) * 4) : CO_OOB, 0, 0));
                a1[u] = __builtin_bit_cast(cf4, __builtin_amdgcn_raw_buffer_load_b128(rP, (on && okb) ? (unsigned)((d.w_off + (size_t)rb * d.ldw + k) * 4) : CO_OOB, 0, 0));
                int xrow;                                                        // state row of input k
                if (d.ring) { const int tap = k / Hp, ii = k - tap * Hp; xrow = a.ring_off[l] + pmod(q - (K - 1 - tap) * g.dil[l], a.ring_len[l]) * Hp + ii; }
                else xrow = d.x_off + k;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) bv[u][jj] = co_ld1(rSt, on ? (unsigned)((((size_t)xrow + jj) * CO + col) * 4) : CO_OOB);
            }
#pragma unroll
            for (int u = 0; u < GMAX; ++u)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u][jj], bv[u][jj], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u][jj], bv[u][jj], acc1, 0, 0, 0);
                }
        }
        *reinterpret_cast<co_f32x4*>(&red[w][0][lane][0]) = acc0;
        *reinterpret_cast<co_f32x4*>(&red[w][1][lane][0]) = acc1;
        __syncthreads();
        if (e_on) {
            // element (row, column) of a 16 x 16 tile lives in lane (row / 4) * 16 + column, component row % 4
            const int t = EPI == 0 ? 0 : e_row >> 4, rr = e_row & 15;
            const int sl = (rr >> 2) * 16 + (tid & 15), si = rr & 3;
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int ww = 0; ww < GW; ++ww) { v0 += red[ww][t][sl][si]; if (EPI == 0) v1 += red[ww][1][sl][si]; }
            if (EPI == 0) {
                const int o = grp * 16 + e_row;
                const float z = sigm(e_gz * (v0 + e_bz));
                const float cd = tanhf(e_gc * (v1 + e_bc));
                const float hn = (1.f - z) * cd + z * e_hp;
                if (l + 1 < g.L) st[(size_t)(a.ring_off[l + 1] + pmod(q, a.ring_len[l + 1]) * Hp + o) * CO + e_col] = hn;
                if (j == np - 1) st[(size_t)(a.o_hcat + l * Hp + o) * CO + e_col] = hn;
            } else {
                const float v = v0 + e_bz;
                st[(size_t)(d.y_off + grp * 32 + e_row) * CO + e_col] = d.relu ? fmaxf(v, 0.f) : v;
            }
        }
        __syncthreads();                                         // red is reused by the next position
    }
}

// ---- sampling + history update, lane = utterance.  Laplace: one wave; softmax: 4 waves split the classes ------
template <int KIND>
__global__ __launch_bounds__(256) void co_tail_kernel(const CoArgs a, const int it) {
    __shared__ float redf[4][CO];
    __shared__ int redi[4][CO];
    const SwnGeom& g = a.g;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), c = blockIdx.x;
    const int b = c * CO + lane;
    const bool live = b < a.B;
    float* st = a.state + (size_t)c * a.stride * CO + lane;
    const float* o2 = st + (size_t)a.o_o2 * CO;
    float* shist = st + (size_t)a.o_hist * CO;
    const int i = it - a.n_pro, seg = g.seg, WN = a.WN;
    if (a.heads && live && w == 0)
        for (int e = 0; e < g.NO; ++e) a.heads[((size_t)b * a.n_steps + i) * g.NO + e] = o2[(size_t)e * CO];
    if (KIND == SWN_KIND_LAPLACE) {
        if (w != 0 || !live) return;
        {
#pragma clang fp contract(off)
        // Laplace head, cswnv_shift1.py:368-391
        const float* forced = reinterpret_cast<const float*>(a.forced);
        float* outp = reinterpret_cast<float*>(a.out) + (size_t)b * a.n_steps * seg + (size_t)i * seg;
        float lp[16], fed[16];
        const int lpc = g.lpc;
        for (int k = 0; k < lpc; ++k) lp[k] = shist[(size_t)(WN - lpc + k) * CO];
        for (int j = 0; j < seg; ++j) {
            const float mu = o2[(size_t)j * CO], yv = o2[(size_t)(seg + j) * CO];
            const float bsc = expf(fminf(yv, 0.f) - log1pf(expf(-fabsf(yv))));
            float lpv = 0.f;
            for (int k = 0; k < lpc; ++k) lpv += o2[(size_t)(2 * seg + lpc - 1 - k) * CO] * lp[k];
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
        for (int k = 0; k + seg < WN; ++k) shist[(size_t)k * CO] = shist[(size_t)(k + seg) * CO];
        for (int j = 0; j < seg; ++j) shist[(size_t)(WN - seg + j) * CO] = fed[j];
        }
    } else {
        // softmax head, dswnv.py:361-369: p = softmax(logits); p /= sum(p); index = argmax(p / q), q ~ Exp(1)
        const int Q = g.Q, e0 = w * ((Q + 3) / 4), e1 = (e0 + (Q + 3) / 4 < Q) ? e0 + (Q + 3) / 4 : Q;
        const int bq = live ? b : 0;
        float m = -INFINITY;
        for (int e = e0; e < e1; ++e) m = fmaxf(m, o2[(size_t)e * CO]);
        redf[w][lane] = m; __syncthreads();
        m = fmaxf(fmaxf(redf[0][lane], redf[1][lane]), fmaxf(redf[2][lane], redf[3][lane])); __syncthreads();
        float sum = 0.f;
        for (int e = e0; e < e1; ++e) sum += expf(o2[(size_t)e * CO] - m);
        redf[w][lane] = sum; __syncthreads();
        sum = (redf[0][lane] + redf[1][lane]) + (redf[2][lane] + redf[3][lane]); __syncthreads();
        float sum2 = 0.f;
        for (int e = e0; e < e1; ++e) sum2 += expf(o2[(size_t)e * CO] - m) / sum;
        redf[w][lane] = sum2; __syncthreads();
        sum2 = (redf[0][lane] + redf[1][lane]) + (redf[2][lane] + redf[3][lane]); __syncthreads();
        float best = -1.f; int bi = 0x7fffffff;
        for (int e = e0; e < e1; ++e) {
            const float rr = ((expf(o2[(size_t)e * CO] - m) / sum) / sum2) / swn_noise_exp1(a.nz, bq, i, e, a.n_steps, Q);
            if (rr > best) { best = rr; bi = e; }
        }
        redf[w][lane] = best; redi[w][lane] = bi; __syncthreads();
        if (w == 0 && live) {
            for (int u = 1; u < 4; ++u) {          // ascending class ranges: a tie keeps the lower index
                const float ob = redf[u][lane]; const int oi = redi[u][lane];
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            const int* forced = reinterpret_cast<const int*>(a.forced);
            reinterpret_cast<int*>(a.out)[(size_t)b * a.n_steps + i] = bi;
            const int fd = forced ? forced[(size_t)b * a.n_steps + i] : bi;
            for (int k = 0; k + 1 < WN; ++k) shist[(size_t)k * CO] = shist[(size_t)(k + 1) * CO];
            shist[(size_t)(WN - 1) * CO] = __builtin_bit_cast(float, fd);
        }
    }
}

// sample window after the state was zeroed: softmax = padding class Q/2 with the caller's seed class newest;
// Laplace = the caller's seed samples in the newest seg slots
__global__ void co_seed_kernel(const CoArgs a) {
    const int c = blockIdx.x, lane = threadIdx.x, b = c * CO + lane;
    float* st = a.state + (size_t)c * a.stride * CO + lane;
    const bool live = b < a.B;
    if (a.g.kind == SWN_KIND_SOFTMAX) {
        const int sc = (a.seed && live) ? reinterpret_cast<const int*>(a.seed)[b] : a.g.Q / 2;
        for (int k = 0; k < a.WN; ++k) st[(size_t)(a.o_hist + k) * CO] = __builtin_bit_cast(float, k == a.WN - 1 ? sc : a.g.Q / 2);
    } else if (a.seed && live) {
        for (int j = 0; j < a.g.seg; ++j)
            st[(size_t)(a.o_hist + a.WN - a.g.seg + j) * CO] = reinterpret_cast<const float*>(a.seed)[(size_t)b * a.g.seg + j];
    }
}

int plan(CoArgs& a) {
    const SwnGeom& g = a.g;
    int o = 0;
    for (int l = 0; l < g.L; ++l) { a.ring_off[l] = o; a.ring_len[l] = g.pad[l] + g.seg; o += a.ring_len[l] * g.Hp; }
    a.WN = (g.K - 1 > g.lpc ? g.K - 1 : g.lpc) + g.seg;
    a.o_hcat = o; o += g.L * g.Hp;
    a.o_skip = o; o += g.Sp;
    a.o_o1 = o; o += g.O1p;
    a.o_o2 = o; o += swn_round4(g.NO);
    a.o_hist = o; o += swn_round4(a.WN);
    a.stride = (o + 63) & ~63;
    return a.stride;
}

}  // namespace

// geometries the matrix-core kernels take: every K a multiple of 16 (16 * GW * GMAX = 1408 inputs per pass: REF6 Laplace
// layers in one pass, the H = 256 softmax layers in two)
bool co_mfma_ok(const SwnGeom& g) {
    return (g.Hp % 16 == 0) && (g.Sp % 16 == 0) && (g.O1p % 16 == 0) && !g.audio_in && g.Hp == g.H;
}

extern "C" size_t swn_decode_cohort_state_floats(const swn_net_desc* d, int batch) {
    CoArgs a;
    if (swn_make_geom(d, &a.g) < 0 || batch < 1) return 0;
    return (size_t)plan(a) * CO * ((batch + CO - 1) / CO);
}

extern "C" int swn_decode_cohort(const swn_net_desc* d, const float* packed, const float* cond, int batch, int n_frames,
                                 int n_steps, const SwnNoise* nz, const void* forced, const void* seed, float* state,
                                 void* out, float* heads, void* stream_) {
    CoArgs a;
    int rc = swn_make_geom(d, &a.g);
    if (rc < 0) return rc;
    const SwnGeom& g = a.g;
    if (g.seg > 16 || g.lpc > 16 || g.Hp != g.H) return SWN_E_UNSUPPORTED;
    if (!state) return SWN_E_BADARG;
    swn_make_layout(&a.g, &a.y);
    plan(a);
    a.P = packed; a.cond = cond; a.nz = *nz; a.forced = forced; a.seed = seed; a.state = state; a.out = out; a.heads = heads;
    a.B = batch; a.Tf = n_frames; a.n_steps = n_steps; a.n_pro = g.rf - g.seg + 1;
    hipStream_t st = (hipStream_t)stream_;
    const unsigned nco = (unsigned)((batch + CO - 1) / CO);
    if (hipMemsetAsync(state, 0, sizeof(float) * (size_t)a.stride * CO * nco, st) != hipSuccess) return SWN_E_LAUNCH;
    const bool lap = g.kind == SWN_KIND_LAPLACE;
    const bool mfma = co_mfma_ok(g);
    auto gemm = [&](int epi, size_t w_off, int ldw, int rows, int split, int ktot, int ring, int x_off,
                    size_t b_off, int y_off, int relu, int l, int it) {
        CoGemm d2;
        d2.w_off = w_off; d2.ldw = ldw; d2.rows = rows; d2.split = split; d2.ktot = ktot; d2.ring = ring;
        d2.x_off = x_off; d2.y_off = y_off; d2.relu = relu; d2.b_off = b_off;
        const int groups = split ? (split + 15) / 16 : (rows + 31) / 32;
        const dim3 gg((unsigned)(groups * 4), nco);             // x = row group * 4 + block of 16 utterances
        if (epi == 0) {
            if (lap) hipLaunchKernelGGL((co_gemm_kernel<SWN_KIND_LAPLACE, 0>), gg, dim3(64 * GW), 0, st, a, d2, l, it);
            else hipLaunchKernelGGL((co_gemm_kernel<SWN_KIND_SOFTMAX, 0>), gg, dim3(64 * GW), 0, st, a, d2, l, it);
        } else {
            hipLaunchKernelGGL((co_gemm_kernel<SWN_KIND_LAPLACE, 1>), gg, dim3(64 * GW), 0, st, a, d2, l, it);
        }
    };
    if (!lap || seed) hipLaunchKernelGGL(co_seed_kernel, dim3(nco), dim3(CO), 0, st, a);
    // slice width per wave and tap (multiple of 4); the instantiated widths cover Hp <= 256 and K <= 8
    const int ipw = ((g.Hp + KS - 1) / KS + 3) & ~3;
    if ((ipw != 4 && ipw != 12 && ipw != 16) || g.K > 8) return SWN_E_UNSUPPORTED;
    {   // stage_rows covers rows * len <= 1024 * NSTG * 4 floats
        const size_t cap = (size_t)1024 * NSTG * 4;
        if ((size_t)2 * PW * g.K * g.Hp > cap || (size_t)RW * g.L * g.Hp > cap || (size_t)RW * g.Sp > cap || (size_t)RW * g.O1p > cap)
            return SWN_E_UNSUPPORTED;
    }
    const size_t lds_layer = ((size_t)2 * PW * g.K * g.Hp + (size_t)KS * 2 * PW * CO) * sizeof(float);
    auto lds_rv = [&](int ni) { return ((size_t)RW * ni + (size_t)KS * RW * CO) * sizeof(float); };
    const size_t lds_max = 150 * 1024;             // gfx950: 160 KB of LDS per CU, one 1024-thread workgroup each
    if (lds_layer > lds_max || lds_rv(g.L * g.Hp) > lds_max || lds_rv(g.Sp) > lds_max || lds_rv(g.O1p) > lds_max)
        return SWN_E_UNSUPPORTED;
    {   // more than the default 64 KB of dynamic LDS has to be enabled per kernel
        const int big = (int)lds_max;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&co_rowvec_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, big);
#define SWN_CO_ATTR(KN) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&KN), hipFuncAttributeMaxDynamicSharedMemorySize, big)
        SWN_CO_ATTR((co_layer_kernel<SWN_KIND_LAPLACE, 4, 8>));  SWN_CO_ATTR((co_layer_kernel<SWN_KIND_SOFTMAX, 4, 8>));
        SWN_CO_ATTR((co_layer_kernel<SWN_KIND_LAPLACE, 12, 3>)); SWN_CO_ATTR((co_layer_kernel<SWN_KIND_SOFTMAX, 12, 3>));
        SWN_CO_ATTR((co_layer_kernel<SWN_KIND_LAPLACE, 16, 2>)); SWN_CO_ATTR((co_layer_kernel<SWN_KIND_SOFTMAX, 16, 2>));
#undef SWN_CO_ATTR
        if (e != hipSuccess) { swn_set_error_detail("swn_decode(cohort): hipFuncSetAttribute", hipGetErrorString(e)); return SWN_E_LAUNCH; }
    }
    const int total = a.n_pro + n_steps;
    for (int it = 0; it < total; ++it) {
        if (lap) hipLaunchKernelGGL(co_in_kernel<SWN_KIND_LAPLACE>, dim3((g.H + 3) / 4, nco), dim3(256), 0, st, a, it);
        else hipLaunchKernelGGL(co_in_kernel<SWN_KIND_SOFTMAX>, dim3((g.H + 3) / 4, nco), dim3(256), 0, st, a, it);
        for (int l = 0; l < g.L; ++l) {
            if (mfma) {     // D[2H][64] = Wd_l[2H][K*Hp] . X: one workgroup per (16 channel pairs, tap)
                gemm(0, a.y.wd + (size_t)l * 2 * g.H * g.K * g.Hp, g.K * g.Hp, 2 * g.H, g.H, g.K * g.Hp, 1, 0, 0, 0, 0, l, it);
                continue;
            }
            const dim3 lg((g.H + PW - 1) / PW, nco);
#define SWN_CO_LAYER(IP, TP_)                                                                                       \
            do {                                                                                                     \
                if (lap) hipLaunchKernelGGL((co_layer_kernel<SWN_KIND_LAPLACE, IP, TP_>), lg, dim3(1024), lds_layer, st, a, l, it); \
                else hipLaunchKernelGGL((co_layer_kernel<SWN_KIND_SOFTMAX, IP, TP_>), lg, dim3(1024), lds_layer, st, a, l, it);     \
            } while (0)
            if (ipw == 4) SWN_CO_LAYER(4, 8);
            else if (ipw == 12) SWN_CO_LAYER(12, 3);
            else SWN_CO_LAYER(16, 2);
#undef SWN_CO_LAYER
        }
        if (it < a.n_pro) continue;
        // row lengths are the padded ones (multiples of 4; the padding of weights and activations is zero)
        if (mfma) {
            gemm(1, a.y.wsk, g.L * g.Hp, g.S, 0, g.L * g.Hp, 0, a.o_hcat, a.y.bsk, a.o_skip, 1, 0, it);
            gemm(1, a.y.w1, g.Sp, g.O1, 0, g.Sp, 0, a.o_skip, a.y.b1, a.o_o1, 1, 0, it);
        } else {
            hipLaunchKernelGGL(co_rowvec_kernel, dim3((g.S + RW - 1) / RW, nco), dim3(1024), lds_rv(g.L * g.Hp), st, a, a.y.wsk, g.L * g.Hp, a.y.bsk,
                               g.S, g.L * g.Hp, a.o_hcat, a.o_skip, 1);
            hipLaunchKernelGGL(co_rowvec_kernel, dim3((g.O1 + RW - 1) / RW, nco), dim3(1024), lds_rv(g.Sp), st, a, a.y.w1, g.Sp, a.y.b1,
                               g.O1, g.Sp, a.o_skip, a.o_o1, 1);
        }
        if (mfma && g.NO >= 32)
            gemm(1, a.y.w2, g.O1p, g.NO, 0, g.O1p, 0, a.o_o1, a.y.b2, a.o_o2, 0, 0, it);
        else
            hipLaunchKernelGGL(co_rowvec_kernel, dim3((g.NO + RW - 1) / RW, nco), dim3(1024), lds_rv(g.O1p), st, a, a.y.w2, g.O1p, a.y.b2,
                               g.NO, g.O1p, a.o_o1, a.o_o2, 0);
        if (lap) hipLaunchKernelGGL(co_tail_kernel<SWN_KIND_LAPLACE>, dim3(nco), dim3(256), 0, st, a, it);
        else hipLaunchKernelGGL(co_tail_kernel<SWN_KIND_SOFTMAX>, dim3(nco), dim3(256), 0, st, a, it);
    }
    return swn_launch_status("swn_decode(cohort)");
}
