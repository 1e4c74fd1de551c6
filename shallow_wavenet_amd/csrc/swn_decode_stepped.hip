// Stepped autoregressive decode for LARGE geometries (reference-shipped REF6: H=192/256, K=7, 15-24 MB
// of weights touched per generated sample) on gfx950.
//
// One CU cannot hold or stream that much per step, and in-kernel cross-CU hand-offs cost 1-3 us each
// (MI355X_MICROARCH.md price list) - about what a dependent kernel boundary costs (1.45 us).  So every
// phase of a step is its OWN launch, spread over many CUs that each read a slice of the weight rows from
// L2 / Infinity Cache:
//     step_in      h0 = softsign(causal(lift(S)))                    1 workgroup / utterance
//     step_layer   x L : rows of one dilated conv + fused gate        8 channel pairs / workgroup
//     rowvec       skip (all out_skip 1x1s as one mat-vec), out_1    8 rows / workgroup
//     step_tail    out_2 + sampling + history update                 1 workgroup / utterance
// L+4 launches per generated step, no spinning, no inter-workgroup protocol: the stream order is the
// dependency chain (cswnv_shift1.py:348-402).  State (history rings, hcat, skip, out_1, sample window,
// iteration counter) lives in the caller's scratch buffer; the math and the ring layout are those of
// the generic persistent kernel (swn_decode.hip), so the two are interchangeable.
#include <hip/hip_runtime.h>
#include "swn_geom.hpp"

namespace {

struct StArgs {
    SwnGeom g;
    SwnLayout y;
    const float* P; const float* cond; const float* noise; const void* forced;
    float* state; void* out; float* heads;
    int B, Tf, n_steps, n_pro, WN;
    int ring_off[SWN_MAXL], ring_len[SWN_MAXL];
    int o_hcat, o_skip, o_o1, o_o2, o_hist, o_cnt, stride;      // per-utterance float offsets
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ int pmod(int r, int m) { int t = r % m; return t < 0 ? t + m : t; }
__device__ __forceinline__ float sum32(float v) {
    v += __shfl_xor(v, 16, 32); v += __shfl_xor(v, 8, 32); v += __shfl_xor(v, 4, 32);
    v += __shfl_xor(v, 2, 32);  v += __shfl_xor(v, 1, 32);
    return v;
}

// iteration `it` of the per-utterance counter: it < n_pro is a prologue position, else generation step
struct Iter { bool gen; int i, np, q0; };
__device__ __forceinline__ Iter iter_of(const StArgs& a, int it) {
    Iter r; r.gen = it >= a.n_pro; r.i = it - a.n_pro; r.np = r.gen ? a.g.seg : 1;
    r.q0 = r.gen ? a.g.rf + 1 - a.g.seg + r.i * a.g.seg : it;
    return r;
}

// ---- step_in: advance the counter, input layer -> ring 0 ----------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void step_in_kernel(const StArgs a) {
    const SwnGeom& g = a.g;
    const int b = blockIdx.x, tid = threadIdx.x;
    float* st = a.state + (size_t)b * a.stride;
    int* cnt = reinterpret_cast<int*>(st + a.o_cnt);         // [0] next iteration, [1] current iteration
    const int it = cnt[0];
    __syncthreads();
    if (tid == 0) { cnt[1] = it; cnt[0] = it + 1; }
    const Iter r = iter_of(a, it);
    const float* P = a.P;
    const float* shist = st + a.o_hist;
    const int* ihist = reinterpret_cast<const int*>(shist);
    const int H = g.H, K = g.K, seg = g.seg, WN = a.WN;
    for (int e = tid; e < H * r.np; e += 256) {
        const int j = e / H, o = e - j * H, q = r.q0 + j;
        float acc = P[a.y.cb + o];
        for (int k = 0; k < K; ++k) {
            const int rr = q - (K - 1 - k);
            if (KIND == SWN_KIND_LAPLACE) {
                if (rr >= -(seg - 1)) {
                    const int qe = r.gen ? g.rf + r.i * seg : g.rf;
                    const float sv = r.gen ? shist[rr - qe + WN - 1] : 0.f;
                    acc += fmaf(P[a.y.cv + (size_t)k * H + o], sv, P[a.y.cc + (size_t)k * H + o]);
                }
            } else if (rr >= 0) {
                const int qe = r.gen ? g.rf + r.i : g.rf;
                const int idx = r.gen ? ihist[rr - qe + WN - 1] : g.Q / 2;
                acc += P[a.y.ct + ((size_t)k * g.Q + idx) * H + o];
            }
        }
        st[a.ring_off[0] + pmod(q, a.ring_len[0]) * g.Hp + o] = acc / (1.f + fabsf(acc));
    }
}

// ---- step_layer: 8 channel pairs per workgroup, 32 lanes per pair, weights kept in registers ------
template <int NI, int KIND>       // NI = ceil(K*Hp / 128): float4 pieces per lane and row
__global__ __launch_bounds__(256) void step_layer_kernel(const StArgs a, const int l) {
    const SwnGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 31, grp = tid >> 5;
    const int o = blockIdx.x * 8 + grp;
    const int H = g.H, Hp = g.Hp, K = g.K, H2 = 2 * g.H, seg = g.seg, KH = K * Hp;
    const bool live = o < H;
    const float* P = a.P;
    float4 wz[NI], wc[NI];
    {
        const float* rz = P + a.y.wd + ((size_t)l * H2 + (live ? o : 0)) * KH;
        const float* rc = rz + (size_t)H * KH;
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int idx = it * 128 + lane * 4;
            const bool ok = live && idx < KH;
            wz[it] = ok ? *reinterpret_cast<const float4*>(rz + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
            wc[it] = ok ? *reinterpret_cast<const float4*>(rc + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const int dil = g.dil[l], R = a.ring_len[l];
    for (int b = blockIdx.y * 8; b < a.B && b < blockIdx.y * 8 + 8; ++b) {
        float* st = a.state + (size_t)b * a.stride;
        const Iter r = iter_of(a, reinterpret_cast<const int*>(st + a.o_cnt)[1]);
        const float* ring = st + a.ring_off[l];
        for (int j = 0; j < r.np; ++j) {
            const int q = r.q0 + j;
            float az = 0.f, ac = 0.f;
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const int idx = it * 128 + lane * 4;
                if (idx < KH) {
                    const int tap = idx / Hp, i = idx - tap * Hp;
                    const float4 x = *reinterpret_cast<const float4*>(ring + (size_t)pmod(q - (K - 1 - tap) * dil, R) * Hp + i);
                    az = fmaf(wz[it].x, x.x, az); az = fmaf(wz[it].y, x.y, az); az = fmaf(wz[it].z, x.z, az); az = fmaf(wz[it].w, x.w, az);
                    ac = fmaf(wc[it].x, x.x, ac); ac = fmaf(wc[it].y, x.y, ac); ac = fmaf(wc[it].z, x.z, ac); ac = fmaf(wc[it].w, x.w, ac);
                }
            }
            az = sum32(az); ac = sum32(ac);
            if (lane == 0 && live) {
                float gz = P[a.y.bx + (size_t)l * H2 + o], gc = P[a.y.bx + (size_t)l * H2 + H + o];
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
                const float z = sigm(gz * (az + P[a.y.bd + (size_t)l * H2 + o]));
                const float c = tanhf(gc * (ac + P[a.y.bd + (size_t)l * H2 + H + o]));
                const float hn = (1.f - z) * c + z * ring[(size_t)pmod(q, R) * Hp + o];
                if (l + 1 < g.L) st[a.ring_off[l + 1] + pmod(q, a.ring_len[l + 1]) * Hp + o] = hn;
                if (j == r.np - 1) st[a.o_hcat + l * Hp + o] = hn;
            }
        }
    }
}

// ---- rowvec: y[b][row] = act(bias[row] + W[row][:] . x[b][:]), 8 rows / workgroup, 32 lanes / row --
__global__ __launch_bounds__(256) void rowvec_kernel(const StArgs a, size_t w_off, int ldw, size_t b_off, int rows,
                                                     int ni, int x_off, int y_off, int relu) {
    const int tid = threadIdx.x, lane = tid & 31, grp = tid >> 5;
    const int row = blockIdx.x * 8 + grp;
    const bool live = row < rows;
    const float* wr = a.P + w_off + (size_t)(live ? row : 0) * ldw;
    const float bias = live ? a.P[b_off + row] : 0.f;
    for (int b = blockIdx.y * 8; b < a.B && b < blockIdx.y * 8 + 8; ++b) {
        float* st = a.state + (size_t)b * a.stride;
        const float* x = st + x_off;
        float acc = 0.f;
        for (int idx = lane * 4; idx < ni; idx += 128) {
            const float4 w = live ? *reinterpret_cast<const float4*>(wr + idx) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 xv = *reinterpret_cast<const float4*>(x + idx);
            acc = fmaf(w.x, xv.x, acc); acc = fmaf(w.y, xv.y, acc); acc = fmaf(w.z, xv.z, acc); acc = fmaf(w.w, xv.w, acc);
        }
        acc = sum32(acc);
        if (lane == 0 && live) { const float v = acc + bias; st[y_off + row] = relu ? fmaxf(v, 0.f) : v; }
    }
}

// ---- step_tail: sampling from the head outputs o2v, history update (one workgroup / utterance) -----
template <int KIND>
__global__ __launch_bounds__(64) void step_tail_kernel(const StArgs a) {
    const SwnGeom& g = a.g;
    const int b = blockIdx.x, tid = threadIdx.x;
    float* st = a.state + (size_t)b * a.stride;
    const int it = reinterpret_cast<const int*>(st + a.o_cnt)[1];
    const int i = it - a.n_pro, seg = g.seg, WN = a.WN;
    const float* o2v = st + a.o_o2;
    float* shist = st + a.o_hist;
    int* ihist = reinterpret_cast<int*>(shist);
    if (a.heads) for (int e = tid; e < g.NO; e += 64) a.heads[((size_t)b * a.n_steps + i) * g.NO + e] = o2v[e];
    if (KIND == SWN_KIND_LAPLACE) {
        if (tid == 0) {
#pragma clang fp contract(off)
            // Laplace head, cswnv_shift1.py:368-391
            const float* nz = a.noise + ((size_t)b * a.n_steps + i) * seg;
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
                const float e = nz[j];
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
        // softmax head, dswnv.py:361-369
        const int Q = g.Q;
        const float* qn = a.noise + ((size_t)b * a.n_steps + i) * Q;
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
            const float r = ((expf(o2v[e] - m) / sum) / sum2) / qn[e];
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

// set the sample window seed (softmax: mu-law zero class) after the state was zeroed
__global__ void step_seed_kernel(const StArgs a) {
    const int b = blockIdx.x;
    int* ihist = reinterpret_cast<int*>(a.state + (size_t)b * a.stride + a.o_hist);
    if ((int)threadIdx.x < a.WN) ihist[threadIdx.x] = a.g.Q / 2;
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

template <int NI>
void launch_layer(const StArgs& a, int l, hipStream_t st) {
    dim3 grid((a.g.H + 7) / 8, (a.B + 7) / 8);
    if (a.g.kind == SWN_KIND_LAPLACE) hipLaunchKernelGGL((step_layer_kernel<NI, SWN_KIND_LAPLACE>), grid, dim3(256), 0, st, a, l);
    else hipLaunchKernelGGL((step_layer_kernel<NI, SWN_KIND_SOFTMAX>), grid, dim3(256), 0, st, a, l);
}

}  // namespace

extern "C" size_t swn_decode_stepped_state_floats(const swn_net_desc* d, int batch) {
    StArgs a;
    if (swn_make_geom(d, &a.g) < 0 || batch < 1) return 0;
    return (size_t)plan(a) * batch;
}

extern "C" int swn_decode_stepped(const swn_net_desc* d, const float* packed, const float* cond, int batch, int n_frames,
                                  int n_steps, const float* noise, const void* forced, float* state, void* out,
                                  float* heads, void* stream_) {
    StArgs a;
    int rc = swn_make_geom(d, &a.g);
    if (rc < 0) return rc;
    const SwnGeom& g = a.g;
    const int ni = (g.K * g.Hp + 127) / 128;
    if (ni > 14 || g.seg > 16 || g.lpc > 16) return SWN_E_UNSUPPORTED;
    swn_make_layout(&a.g, &a.y);
    plan(a);
    a.P = packed; a.cond = cond; a.noise = noise; a.forced = forced; a.state = state; a.out = out; a.heads = heads;
    a.B = batch; a.Tf = n_frames; a.n_steps = n_steps; a.n_pro = g.rf - g.seg + 1;
    hipStream_t st = (hipStream_t)stream_;
    if (hipMemsetAsync(state, 0, sizeof(float) * (size_t)a.stride * batch, st) != hipSuccess) return SWN_E_LAUNCH;
    if (g.kind == SWN_KIND_SOFTMAX) hipLaunchKernelGGL(step_seed_kernel, dim3(batch), dim3(64), 0, st, a);
    const int total = a.n_pro + n_steps;
    const dim3 bgrid8((unsigned)1, (unsigned)((batch + 7) / 8));
    for (int it = 0; it < total; ++it) {
        if (g.kind == SWN_KIND_LAPLACE) hipLaunchKernelGGL(step_in_kernel<SWN_KIND_LAPLACE>, dim3(batch), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(step_in_kernel<SWN_KIND_SOFTMAX>, dim3(batch), dim3(256), 0, st, a);
        for (int l = 0; l < g.L; ++l) {
            if (ni <= 1) launch_layer<1>(a, l, st);
            else if (ni <= 4) launch_layer<4>(a, l, st);
            else if (ni <= 11) launch_layer<11>(a, l, st);
            else launch_layer<14>(a, l, st);
        }
        if (it < a.n_pro) continue;
        hipLaunchKernelGGL(rowvec_kernel, dim3((g.S + 7) / 8, bgrid8.y), dim3(256), 0, st, a, a.y.wsk, g.L * g.Hp, a.y.bsk,
                           g.S, g.L * g.Hp, a.o_hcat, a.o_skip, 1);
        hipLaunchKernelGGL(rowvec_kernel, dim3((g.O1 + 7) / 8, bgrid8.y), dim3(256), 0, st, a, a.y.w1, g.Sp, a.y.b1,
                           g.O1, g.Sp, a.o_skip, a.o_o1, 1);
        hipLaunchKernelGGL(rowvec_kernel, dim3((g.NO + 7) / 8, bgrid8.y), dim3(256), 0, st, a, a.y.w2, g.O1p, a.y.b2,
                           g.NO, g.O1p, a.o_o1, a.o_o2, 0);
        if (g.kind == SWN_KIND_LAPLACE) hipLaunchKernelGGL(step_tail_kernel<SWN_KIND_LAPLACE>, dim3(batch), dim3(64), 0, st, a);
        else hipLaunchKernelGGL(step_tail_kernel<SWN_KIND_SOFTMAX>, dim3(batch), dim3(64), 0, st, a);
    }
    return swn_launch_status("swn_decode(stepped)");
}
