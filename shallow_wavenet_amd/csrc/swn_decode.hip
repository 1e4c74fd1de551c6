// Autoregressive decode on gfx950: ONE persistent workgroup per utterance runs the prologue
// (seed positions 0..rf-seg) and every generation step inside a single launch; the ~22
// convolution launches and ~5.3k ATen calls per sample of the reference loop
// (cswnv_shift1.py:348-402, dswnv.py:338-374) become LDS/L2 traffic and s_barriers.
//
// This file holds the GENERIC kernel: any (H, S, K, dd, dr, seg, lpc, Q) at run time, history
// rings in a global scratch buffer (L2-resident), weights streamed from the packed buffer.
// The register/LDS-resident BL6-class kernel lives in swn_decode_bl6.hip.
//
// Per position q the math is (cswnv_shift1.py:281-285, :352-391):
//   h0[q]  = softsign(cb + sum_k [valid] (cv_k * S[q-(K-1-k)] + cc_k))         fused wav_conv+causal
//   a      = bd_l + Wd_l . [h_{l-1}[q-(K-1)d] .. h_{l-1}[q]]                    dilated causal conv
//   g      = in_x_l(x)[q] (.) a ;  z = sigmoid(g[:H]) ; h_l = (1-z) tanh(g[H:]) + z h_{l-1}[q]
//   head   = out_2(relu(out_1(relu(sum_l out_skip_l(h_l)))))   at the last position of a step
// The "true zero" left padding of the reference prologue is reproduced by zero-initialised
// rings: a slot that would hold a negative position has not been written yet.
#include <hip/hip_runtime.h>
#include "swn_geom.hpp"
#include "swn_noise.hpp"

namespace {

constexpr int NT = 256;

struct DecArgs {
    SwnGeom g;
    SwnLayout y;
    const float* packed;
    const float* cond;
    SwnNoise nz;
    const void* forced;
    const void* seed;
    float* state;
    void* out;
    float* heads;
    int B, Tf, n_steps;
    int ring_off[SWN_MAXL];
    int ring_len[SWN_MAXL];
    int state_stride;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softsignf_(float x) { return x / (1.f + fabsf(x)); }
__device__ __forceinline__ int pmod(int r, int m) { int t = r % m; return t < 0 ? t + m : t; }

// rows x (nseg segments of SL floats) mat-vec for NP right-hand sides.  16 lanes per row read
// 16-byte pieces of the row (256 B contiguous per row and pass), rows are reduced with
// row-local shuffles.  xf(j, s) returns the base of segment s of right-hand side j.
template <int NP, class XF>
__device__ __forceinline__ void matvec16(const float* __restrict__ W, int NR, int nseg, int SL, XF xf,
                                         float* out, int ldout, const float* __restrict__ bias,
                                         bool relu, int np) {
    const int p = threadIdx.x & 15, rs = threadIdx.x >> 4;
    const size_t rowlen = (size_t)nseg * SL;
    for (int r0 = 0; r0 < NR; r0 += NT / 16) {
        const int row = r0 + rs;
        float acc[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) acc[j] = 0.f;
        if (row < NR) {
            const float* wr = W + (size_t)row * rowlen;
            for (int s = 0; s < nseg; ++s) {
                const float* xs[NP];
#pragma unroll
                for (int j = 0; j < NP; ++j) xs[j] = (j < np) ? xf(j, s) : nullptr;
                for (int i0 = p * 4; i0 < SL; i0 += 64) {
                    const float4 w = *reinterpret_cast<const float4*>(wr + (size_t)s * SL + i0);
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        if (j < np) {
                            const float4 x = *reinterpret_cast<const float4*>(xs[j] + i0);
                            acc[j] = fmaf(w.x, x.x, acc[j]);
                            acc[j] = fmaf(w.y, x.y, acc[j]);
                            acc[j] = fmaf(w.z, x.z, acc[j]);
                            acc[j] = fmaf(w.w, x.w, acc[j]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            float v = acc[j];
            v += __shfl_xor(v, 8, 16);
            v += __shfl_xor(v, 4, 16);
            v += __shfl_xor(v, 2, 16);
            v += __shfl_xor(v, 1, 16);
            if (p == 0 && row < NR && j < np) {
                v += bias[row];
                out[j * ldout + row] = relu ? fmaxf(v, 0.f) : v;
            }
        }
    }
}

template <int SEGT, int KIND>
__global__ __launch_bounds__(NT) void decode_generic_kernel(const DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SwnGeom& g = a.g;
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const int H = g.H, Hp = g.Hp, H2 = 2 * g.H, K = g.K, L = g.L, seg = g.seg, S = g.S;
    const int WN = (K - 1 > g.lpc ? K - 1 : g.lpc) + seg;
    const float* __restrict__ P = a.packed;

    // ---- LDS carve (all offsets multiples of 4 floats)
    float* a_out = smem;                                    // [SEGT][2H]
    float* hcat = a_out + SEGT * swn_round4(H2);            // [L][Hp]
    float* skipv = hcat + L * Hp;                           // [Sp]
    float* o1v = skipv + g.Sp;                              // [O1p]
    float* o2v = o1v + g.O1p;                               // [round4(NO)]
    float* shist = o2v + swn_round4(g.NO);                  // [round4(WN)]  (int bits for softmax)
    float* tw = shist + swn_round4(WN);                     // [SEGT*SEGT]
    int* tf = reinterpret_cast<int*>(tw + SEGT * SEGT);     // [SEGT*SEGT]
    const int lds_floats = (int)((tf + SEGT * SEGT) - reinterpret_cast<int*>(smem));
    for (int e = tid; e < lds_floats; e += NT) smem[e] = 0.f;
    int* ihist = reinterpret_cast<int*>(shist);
    __syncthreads();
    if (KIND == SWN_KIND_SOFTMAX) {
        // padding class Q/2 = encode_mu_law(0), dswnv.py:308; the newest slot is the caller's seed class
        const int sc = a.seed ? reinterpret_cast<const int*>(a.seed)[b] : g.Q / 2;
        for (int e = tid; e < WN; e += NT) ihist[e] = (e == WN - 1) ? sc : g.Q / 2;
    } else if (a.seed) {
        // seed waveform `audio` (B, seg): the newest seg samples of the window (cswnv_shift1.py:300-334)
        for (int e = tid; e < seg; e += NT) shist[WN - seg + e] = reinterpret_cast<const float*>(a.seed)[(size_t)b * seg + e];
    }
    __syncthreads();

    float* st = a.state + (size_t)b * a.state_stride;
    const float* condb = a.cond + (size_t)b * a.Tf * g.N;
    const int ld_a = swn_round4(H2);
    const int rf = g.rf;
    const int n_pro = rf - seg + 1;                          // prologue positions 0..rf-seg
    const int total = n_pro + a.n_steps;

    for (int it = 0; it < total; ++it) {
        const bool gen = it >= n_pro;
        const int i = it - n_pro;                            // generation step index
        const int np = gen ? seg : 1;
        const int q0 = gen ? rf + 1 - seg + i * seg : it;    // first position handled now

        // ---- conditioning lookups for this step (frame index / upsampler tap per (j,s))
        if (tid < np * seg) {
            const int j = tid / seg, s = tid % seg;
            int tt = q0 + j + s - rf; tt = tt < 0 ? 0 : tt;
            int f = tt / g.U; const int jj = tt - f * g.U;
            f = f < a.Tf ? f : a.Tf - 1;
            tf[j * SEGT + s] = f;
            tw[j * SEGT + s] = P[a.y.wup + jj];
        }
        // ---- input layer: h0 = softsign(causal(lift(S)))  -> ring 0
        for (int e = tid; e < H * np; e += NT) {
            const int j = e / H, o = e - j * H;
            const int q = q0 + j;
            float acc = P[a.y.cb + o];
            for (int k = 0; k < K; ++k) {
                const int r = q - (K - 1 - k);               // sample position feeding tap k
                if (KIND == SWN_KIND_LAPLACE) {
                    if (r >= -(seg - 1)) {
                        // window holds S[qe-WN+1..qe], qe = last known position
                        const int qe = gen ? rf + i * seg : rf;
                        const float sv = gen ? shist[r - qe + WN - 1] : 0.f;
                        acc += fmaf(P[a.y.cv + (size_t)k * H + o], sv, P[a.y.cc + (size_t)k * H + o]);
                    }
                } else {
                    if (r >= 0) {
                        const int qe = gen ? rf + i : rf;
                        const int idx = gen ? ihist[r - qe + WN - 1] : g.Q / 2;
                        acc += P[a.y.ct + ((size_t)k * g.Q + idx) * H + o];
                    }
                }
            }
            st[a.ring_off[0] + pmod(q, a.ring_len[0]) * Hp + o] = softsignf_(acc);
        }
        __syncthreads();

        // ---- stack
        for (int l = 0; l < L; ++l) {
            const int dil = g.dil[l], R = a.ring_len[l];
            const float* ring = st + a.ring_off[l];
            auto xf = [&](int j, int k) -> const float* {
                return ring + (size_t)pmod(q0 + j - (K - 1 - k) * dil, R) * Hp;
            };
            matvec16<SEGT>(P + a.y.wd + (size_t)l * H2 * K * Hp, H2, K, Hp, xf, a_out, ld_a,
                           P + a.y.bd + (size_t)l * H2, false, np);
            __syncthreads();
            for (int e = tid; e < H * np; e += NT) {
                const int j = e / H, o = e - j * H;
                const int q = q0 + j;
                float gxz = P[a.y.bx + (size_t)l * H2 + o];
                float gxc = P[a.y.bx + (size_t)l * H2 + H + o];
                for (int s = 0; s < seg; ++s) {
                    const float* cr = condb + (size_t)tf[j * SEGT + s] * g.N + (size_t)(l * seg + s) * H2;
                    const float w = tw[j * SEGT + s];
                    gxz = fmaf(w, cr[o], gxz);
                    gxc = fmaf(w, cr[H + o], gxc);
                }
                if (KIND == SWN_KIND_SOFTMAX && g.audio_in) {
                    const int qe = gen ? rf + i : rf;
                    const int idx = gen ? ihist[q - qe + WN - 1] : g.Q / 2;
                    const float* wa = P + a.y.wxa + ((size_t)l * g.Q + idx) * H2;
                    gxz += wa[o]; gxc += wa[H + o];
                }
                const float z = sigmoidf_(gxz * a_out[j * ld_a + o]);
                const float c = tanhf(gxc * a_out[j * ld_a + H + o]);
                const float hp = ring[(size_t)pmod(q, R) * Hp + o];
                const float hn = (1.f - z) * c + z * hp;
                if (l + 1 < L) st[a.ring_off[l + 1] + pmod(q, a.ring_len[l + 1]) * Hp + o] = hn;
                if (j == np - 1) hcat[l * Hp + o] = hn;
            }
            __syncthreads();
        }
        if (!gen) continue;

        // ---- head at the last position of the step
        {
            auto x1 = [&](int, int) -> const float* { return hcat; };
            matvec16<1>(P + a.y.wsk, S, 1, L * Hp, x1, skipv, 0, P + a.y.bsk, true, 1);
            __syncthreads();
            auto x2 = [&](int, int) -> const float* { return skipv; };
            matvec16<1>(P + a.y.w1, g.O1, 1, g.Sp, x2, o1v, 0, P + a.y.b1, true, 1);
            __syncthreads();
            auto x3 = [&](int, int) -> const float* { return o1v; };
            matvec16<1>(P + a.y.w2, g.NO, 1, g.O1p, x3, o2v, 0, P + a.y.b2, false, 1);
            __syncthreads();
        }
        if (a.heads)
            for (int e = tid; e < g.NO; e += NT) a.heads[((size_t)b * a.n_steps + i) * g.NO + e] = o2v[e];

        if (KIND == SWN_KIND_LAPLACE) {
            // Laplace head, cswnv_shift1.py:368-391: b = exp(logsigmoid(.)), LP coefficients flipped,
            // one uniform draw per sample, clamp before feedback.
            if (tid == 0) {
#pragma clang fp contract(off)
                const float* forced = reinterpret_cast<const float*>(a.forced);
                float* outp = reinterpret_cast<float*>(a.out) + (size_t)b * a.n_steps * seg + (size_t)i * seg;
                float lp[16];
                const int lpc = g.lpc;
                for (int k = 0; k < lpc; ++k) lp[k] = shist[WN - lpc + k];
                float fed[SEGT];
                for (int j = 0; j < seg; ++j) {
                    const float mu = o2v[j];
                    const float yv = o2v[seg + j];
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
                for (int k = 0; k + seg < WN; ++k) shist[k] = shist[k + seg];
                for (int j = 0; j < seg; ++j) shist[WN - seg + j] = fed[j];
            }
        } else {
            // softmax head, dswnv.py:361-369: softmax -> Categorical renormalisation ->
            // multinomial(n=1) == argmax(p / q) with q ~ Exp(1) supplied by the host.
            if (tid < 64) {
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
                for (int e = tid; e < Q; e += 64) {
                    const float r = ((expf(o2v[e] - m) / sum) / sum2) / swn_noise_exp1(a.nz, b, i, e, a.n_steps, Q);
                    if (r > best) { best = r; bi = e; }
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
                    for (int k = 0; k + 1 < WN; ++k) ihist[k] = ihist[k + 1];
                    ihist[WN - 1] = fd;
                }
            }
        }
        __syncthreads();
    }
}

template <int SEGT>
int launch_generic(const DecArgs& a, size_t lds, hipStream_t st) {
    if (a.g.kind == SWN_KIND_LAPLACE)
        hipLaunchKernelGGL((decode_generic_kernel<SEGT, SWN_KIND_LAPLACE>), dim3(a.B), dim3(NT), lds, st, a);
    else
        hipLaunchKernelGGL((decode_generic_kernel<SEGT, SWN_KIND_SOFTMAX>), dim3(a.B), dim3(NT), lds, st, a);
    return swn_launch_status("swn_decode(generic)");
}

int ring_plan(const SwnGeom& g, int* off, int* len) {
    int o = 0;
    for (int l = 0; l < g.L; ++l) {
        off[l] = o; len[l] = g.pad[l] + g.seg;
        o += len[l] * g.Hp;
    }
    return (o + 63) & ~63;
}

}  // namespace

// defined in swn_decode_bl6.hip; returns SWN_E_UNSUPPORTED when the geometry is not a BL6-class one
extern "C" int swn_decode_bl6_try(const swn_net_desc* d, const float* packed, const float* cond,
                                  int batch, int n_frames, int n_steps, const SwnNoise* nz,
                                  const void* forced, const void* seed, void* out, float* heads, void* stream);

// defined in swn_decode_bl6w.hip: the wave-specialised form for the single-sample Laplace nets of that class
extern "C" int swn_decode_bl6w_try(const swn_net_desc* d, const float* packed, const float* cond,
                                   int batch, int n_frames, int n_steps, const SwnNoise* nz,
                                   const void* forced, const void* seed, void* out, float* heads, void* stream);

// defined in swn_decode_stepped.hip
extern "C" size_t swn_decode_stepped_state_floats(const swn_net_desc* d, int batch);
extern "C" int swn_decode_stepped(const swn_net_desc* d, const float* packed, const float* cond, int batch, int n_frames,
                                  int n_steps, const SwnNoise* nz, const void* forced, const void* seed, float* state,
                                  void* out, float* heads, void* stream);

extern "C" size_t swn_decode_state_floats(const swn_net_desc* d, int batch) {
    SwnGeom g; if (swn_make_geom(d, &g) < 0 || batch < 1) return 0;
    int off[SWN_MAXL], len[SWN_MAXL];
    size_t a = (size_t)ring_plan(g, off, len) * batch;
    const size_t b = swn_decode_stepped_state_floats(d, batch);
    return a > b ? a : b;                                  // large enough for every kernel variant
}

extern "C" int swn_decode(const swn_net_desc* d, const float* packed, const float* cond, int batch,
                          int n_frames, int n_steps, const swn_decode_io* io,
                          float* state, void* out, float* heads, int variant, void* stream_) {
    DecArgs a;
    int rc = swn_make_geom(d, &a.g);
    if (rc < 0) return rc;
    if (batch < 1 || n_frames < 1 || n_steps < 0 || !io) return SWN_E_BADARG;
    if (n_steps == 0) return SWN_OK;                       // nothing to generate (empty buffers may be null)
    if (!packed || !cond || !out) return SWN_E_BADARG;
    if ((long)n_steps * a.g.seg > (long)n_frames * a.g.U) return SWN_E_BADARG;   // conditioning too short
    SwnNoise nz;
    nz.ptr = io->noise_dev; nz.dump = io->noise_out_dev;
    nz.key0 = (uint32_t)(io->rng_seed & 0xffffffffu); nz.key1 = (uint32_t)(io->rng_seed >> 32); nz.utt0 = io->rng_utt0; nz.ids = io->rng_utt_ids_dev;
    const void* forced = io->forced_dev;
    const void* seed = io->seed_dev;
    hipStream_t st = (hipStream_t)stream_;
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls
    if (variant == 0 || variant == 2) {        // BL6 class: the wave-specialised kernel where it applies, else the symmetric one
        rc = swn_decode_bl6w_try(d, packed, cond, batch, n_frames, n_steps, &nz, forced, seed, out, heads, stream_);
        if (rc != SWN_E_UNSUPPORTED) return rc;
    }
    if (variant == 0 || variant == 2 || variant == 6) {     // 6 = the symmetric BL6 kernel, whatever the net (A/B and parity runs)
        rc = swn_decode_bl6_try(d, packed, cond, batch, n_frames, n_steps, &nz, forced, seed, out, heads, stream_);
        if (rc != SWN_E_UNSUPPORTED || variant != 0) return rc;
    }
    if (!state) return SWN_E_BADARG;
    // large geometries (REF6: MBs of weights per step) run one launch per phase over many CUs
    const bool big = (size_t)a.g.L * 2 * a.g.H * a.g.K * a.g.Hp >= (size_t)256 * 1024;
    if (variant < 0 || variant > 3) return SWN_E_BADARG;   // (4 and 5, the cohort and cluster experiments of ABI 2, are retired)
    if (variant == 3 || (variant == 0 && big)) {
        rc = swn_decode_stepped(d, packed, cond, batch, n_frames, n_steps, &nz, forced, seed, state, out, heads, stream_);
        if (rc != SWN_E_UNSUPPORTED || variant == 3) return rc;
    }
    swn_make_layout(&a.g, &a.y);
    a.packed = packed; a.cond = cond; a.nz = nz; a.forced = forced; a.seed = seed; a.state = state;
    a.out = out; a.heads = heads; a.B = batch; a.Tf = n_frames; a.n_steps = n_steps;
    a.state_stride = ring_plan(a.g, a.ring_off, a.ring_len);
    if (hipMemsetAsync(state, 0, sizeof(float) * (size_t)a.state_stride * batch, st) != hipSuccess)
        return SWN_E_LAUNCH;
    const SwnGeom& g = a.g;
    const int segt = g.seg <= 1 ? 1 : (g.seg <= 2 ? 2 : (g.seg <= 5 ? 5 : 10));
    const int WN = (g.K - 1 > g.lpc ? g.K - 1 : g.lpc) + g.seg;
    const size_t lds_floats = (size_t)segt * swn_round4(2 * g.H) + (size_t)g.L * g.Hp + g.Sp + g.O1p +
                              swn_round4(g.NO) + swn_round4(WN) + 2 * (size_t)segt * segt;
    const size_t lds = lds_floats * sizeof(float);
    if (lds > 160 * 1024) return SWN_E_UNSUPPORTED;
    switch (segt) {
        case 1: return launch_generic<1>(a, lds, st);
        case 2: return launch_generic<2>(a, lds, st);
        case 5: return launch_generic<5>(a, lds, st);
        default: return launch_generic<10>(a, lds, st);
    }
}
