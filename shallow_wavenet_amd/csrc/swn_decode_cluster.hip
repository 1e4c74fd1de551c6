// Cluster decode for LARGE geometries (reference-shipped REF6: H=192/256, K=7, 15-24 MB of weights per generated
// sample) on gfx950: ONE persistent launch, the CUs of one XCD form a cluster that carries an utterance (or up to 8 of
// them in lock step) through every phase of every step, handing the H-vector of each layer to one another through
// memory with flags - SURVEY.md 7.3 option (i).
//
// Why: the stepped decode (swn_decode_stepped.hip) pays a kernel boundary plus two memory round trips per phase,
// 3.3-4.2 us x (L+3) launches per generated step.  Here a phase boundary is a cluster barrier (agent-scope atomic add
// + sc1 poll, ~1-1.5 us), the dil_h weight rows of the NEXT layer stream into registers while the current phase waits,
// and sampling / out_2 / the input layer are evaluated redundantly by every CU, so they need no hand-off at all.
//
//   cluster     = the workgroups (one per CU, forced by the LDS request) that report the same HW_REG_XCC_ID; each takes
//                 a slot from a per-XCD ticket.  Nothing depends on the dispatch order.
//   partition   = slot s owns hidden channels [s*CPC, (s+1)*CPC) of every layer (gate + candidate rows), SPC rows of
//                 the concatenated out_skip 1x1 and OPC rows of out_1 (and of out_2 when it is wide: softmax).
//   hand-off    = producer: sc1 (write-through) stores -> every wave s_waitcnt vmcnt(0) -> workgroup barrier -> one
//                 agent-scope atomic add; consumer: one lane polls with sc1 loads, workgroup barrier, then every load of
//                 handed-off bytes is an sc1 buffer load (MI355X_MICROARCH.md "inter-workgroup visibility", first table row).
//   safety      = every poll loop is bounded; a cluster that cannot assemble (workgroups not co-resident) raises the
//                 abort word, every workgroup leaves at its next barrier and the outputs of the launch are filled with
//                 NaN / -1 so that the failure cannot pass for a result.
// The math, the ring layout and the noise are those of swn_decode.hip / swn_decode_stepped.hip (cswnv_shift1.py:281-430,
// dswnv.py:290-399).
#include <hip/hip_runtime.h>
#include "swn_geom.hpp"
#include "swn_noise.hpp"

namespace {

constexpr int NT = 256;              // 4 waves, one per SIMD: the full 512-entry register file per wave
constexpr int UBM = 8;               // utterances a cluster carries in lock step (per pass)
constexpr int NXCD = 8;
constexpr int CPC_MAX = 8;           // channels per CU (H <= 256 on >= 32 CUs)
constexpr int SEG_MAX = 10;
constexpr unsigned SPIN_LIMIT = 1u << 21;
constexpr unsigned SC1 = 16;         // buffer-instruction cache policy bit: sc1
constexpr unsigned CL_OOB = 0x80000000u;

// control words, each on a 128-byte line of its own (32 uints apart)
enum { CW_TICKET = 0, CW_BAR = 8, CW_ARRIVED = 16, CW_ABORT = 17, CW_COUNT = 18 };
constexpr int CW_STRIDE = 32;

struct ClArgs {
    SwnGeom g;
    SwnLayout y;
    const float* P; const float* cond; SwnNoise nz; const void* forced; const void* seed;
    float* state; void* out; float* heads; unsigned* ctrl;
    int B, Tf, n_steps, n_pro, WN;
    int ring_off[SWN_MAXL], ring_len[SWN_MAXL];      // ring l (l >= 1): input history of layer l, in the state block
    int o_hlast, o_skip, o_o1, o_o2, stride;         // per-utterance float offsets in the state block
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ int pmod(int r, int m) { int t = r % m; return t < 0 ? t + m : t; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cl_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {           // plain (read-only data)
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float4 ld4_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {       // handed-off data
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, SC1));
}
__device__ __forceinline__ float ld1_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, SC1));
}
__device__ __forceinline__ void st1_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, 0, SC1);
}
__device__ __forceinline__ float sum64(float v) {
    v += __shfl_xor(v, 32, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);  v += __shfl_xor(v, 1, 64);
    return v;
}
__device__ __forceinline__ unsigned aload(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void astore(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned aadd(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// bounded wait until *p >= target; false = gave up (abort raised) or somebody else aborted
__device__ __forceinline__ bool wait_ge(const unsigned* p, unsigned target, unsigned* abortw) {
    unsigned spins = 0;
    while (aload(p) < target) {
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if (spins > SPIN_LIMIT) { astore(abortw, 1u); return false; }
        if ((spins & 255u) == 0u && aload(abortw) != 0u) return false;
    }
    return true;
}

// NI = float4 pieces per lane and row (ceil(K*Hp / 256)), RW = rows per wave, KIND
template <int NI, int RW, int KIND>
__global__ __launch_bounds__(NT, 1) void decode_cluster_kernel(const ClArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_dead;
    __shared__ unsigned s_tickets[NXCD];
    const SwnGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = g.H, Hp = g.Hp, K = g.K, L = g.L, H2 = 2 * g.H, seg = g.seg, S = g.S, KH = K * Hp, WN = a.WN;
    const float* __restrict__ P = a.P;
    unsigned* abortw = a.ctrl + CW_ABORT * CW_STRIDE;

    // ---- cluster assembly: which XCD am I on, which slot do I get, how large did every cluster become
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= (NXCD - 1);
    if (tid == 0) {
        s_dead = 0;
        s_tickets[0] = aadd(a.ctrl + (CW_TICKET + xcc) * CW_STRIDE, 1u);      // my slot
        aadd(a.ctrl + CW_ARRIVED * CW_STRIDE, 1u);
        if (!wait_ge(a.ctrl + CW_ARRIVED * CW_STRIDE, gridDim.x, abortw)) s_dead = 1;
    }
    __syncthreads();
    if (s_dead) return;
    const int slot = (int)s_tickets[0];
    __syncthreads();
    if (tid < NXCD) s_tickets[tid] = aload(a.ctrl + (CW_TICKET + tid) * CW_STRIDE);
    __syncthreads();
    int ncl = 0, cidx = 0;
    for (int x = 0; x < NXCD; ++x) { if (s_tickets[x] > 0) { if (x < (int)xcc) ++cidx; ++ncl; } }
    const int NC = (int)s_tickets[xcc];
    unsigned* bar = a.ctrl + (CW_BAR + xcc) * CW_STRIDE;
    unsigned bar_k = 0;                                     // barriers passed by this cluster

    // partition of the rows over the cluster
    const int CPC = (H + NC - 1) / NC;
    if (CPC > CPC_MAX || 2 * CPC > 4 * RW) {                // cluster too small for this instantiation: give up loudly
        if (tid == 0) astore(abortw, 2u);
        return;
    }
    const int ch0 = slot * CPC, nch = ch0 < H ? (H - ch0 < CPC ? H - ch0 : CPC) : 0;
    const int SPC = (S + NC - 1) / NC, sr0 = slot * SPC, nsr = sr0 < S ? (S - sr0 < SPC ? S - sr0 : SPC) : 0;
    const int OPC = (g.O1 + NC - 1) / NC, or0 = slot * OPC, nor = or0 < g.O1 ? (g.O1 - or0 < OPC ? g.O1 - or0 : OPC) : 0;
    const bool wide = g.NO > 64;                             // softmax: out_2 rows are distributed like out_1's
    const int NPC = (g.NO + NC - 1) / NC, nr0 = slot * NPC, nnr = nr0 < g.NO ? (g.NO - nr0 < NPC ? g.NO - nr0 : NPC) : 0;
    // utterances of this cluster: b = cidx, cidx + ncl, ...  ; a pass carries up to UBM of them
    const int n_mine = a.B > cidx ? (a.B - cidx + ncl - 1) / ncl : 0;

    // ---- LDS carve
    const int R0 = a.ring_len[0];
    float* ring0 = lds;                                      // [UBM][R0][Hp]  local history of the input layer
    float* rsum = ring0 + UBM * R0 * Hp;                     // [UBM][SEG_MAX][2*CPC_MAX] gate / candidate row sums
    float* win = rsum + UBM * SEG_MAX * 2 * CPC_MAX;         // [UBM][32] sample windows (float samples | int classes)
    float* o2v = win + UBM * 32;                             // [UBM][round4(NO)]
    float* vec = o2v + UBM * swn_round4(g.NO);               // [UBM][max(Sp, O1p)] staging of a full skip / out_1 vector
    int* iwin = reinterpret_cast<int*>(win);

    const __amdgpu_buffer_rsrc_t rP = cl_rsrc(P), rS = cl_rsrc(a.state);
    auto cluster_barrier = [&]() -> bool {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's sc1 stores have left
        __syncthreads();
        ++bar_k;
        if (tid == 0) {
            aadd(bar, 1u);
            if (!wait_ge(bar, (unsigned)NC * bar_k, abortw)) s_dead = 1;
        }
        __syncthreads();
        return s_dead == 0;
    };

    // weight rows of layer l for this wave: rows rw -> local row index w*RW + rw -> channel ch0 + (idx >> 1), gate | cand
    float4 wcur[RW][NI], wnext[RW][NI];
    auto load_rows = [&](int l, float4 (&dst)[RW][NI]) {
#pragma unroll
        for (int rw = 0; rw < RW; ++rw) {
            const int lr = w * RW + rw, c = lr >> 1;
            const bool live = c < nch;
            const size_t row = (size_t)l * H2 + (lr & 1 ? H : 0) + ch0 + c;
#pragma unroll
            for (int pc = 0; pc < NI; ++pc) {
                const int idx = pc * 256 + lane * 4;
                dst[rw][pc] = ld4(rP, (live && idx < KH) ? (unsigned)((a.y.wd + row * KH + idx) * 4) : CL_OOB);
            }
        }
    };

    for (int pass0 = 0; pass0 < n_mine; pass0 += UBM) {
        const int UB = n_mine - pass0 < UBM ? n_mine - pass0 : UBM;
        auto utt = [&](int u) { return cidx + (pass0 + u) * ncl; };           // global utterance index
        // ---- per-pass initialisation: local rings and sample windows (the global rings were zeroed by the host)
        for (int e = tid; e < UBM * R0 * Hp; e += NT) ring0[e] = 0.f;
        for (int e = tid; e < UBM * 32; e += NT) {
            const int u = e >> 5, k = e & 31;
            if (KIND == SWN_KIND_SOFTMAX) {
                int v = g.Q / 2;
                if (k == WN - 1 && u < UB && a.seed) v = reinterpret_cast<const int*>(a.seed)[utt(u)];
                iwin[e] = v;
            } else {
                float v = 0.f;
                if (u < UB && a.seed && k >= WN - seg && k < WN) v = reinterpret_cast<const float*>(a.seed)[(size_t)utt(u) * seg + (k - (WN - seg))];
                win[e] = v;
            }
        }
        __syncthreads();
        load_rows(0, wcur);

        const int total = a.n_pro + a.n_steps;
        for (int it = 0; it < total; ++it) {
            const bool gen = it >= a.n_pro;
            const int i = it - a.n_pro, np = gen ? seg : 1;
            const int q0 = gen ? g.rf + 1 - seg + i * seg : it;
            const int nl = gen ? L : L - 1;                  // the prologue never needs the last layer's output

            // ---- input layer (every CU, every channel): h0 -> local ring (cswnv_shift1.py:352 / dswnv.py:345)
            for (int e = tid; e < UB * np * H; e += NT) {
                const int o = e % H, j = (e / H) % np, u = e / (H * np);
                const int q = q0 + j;
                float acc = P[a.y.cb + o];
                for (int k = 0; k < K; ++k) {
                    const int rr = q - (K - 1 - k);
                    if (KIND == SWN_KIND_LAPLACE) {
                        const int qe = gen ? g.rf + i * seg : g.rf;
                        int wi = rr - qe + WN - 1; wi = wi < 0 ? 0 : (wi >= WN ? WN - 1 : wi);
                        const float sv = gen ? win[u * 32 + wi] : 0.f;
                        const float t = fmaf(P[a.y.cv + (size_t)k * H + o], sv, P[a.y.cc + (size_t)k * H + o]);
                        acc += (rr >= -(seg - 1)) ? t : 0.f;
                    } else {
                        const int qe = gen ? g.rf + i : g.rf;
                        int wi = rr - qe + WN - 1; wi = wi < 0 ? 0 : (wi >= WN ? WN - 1 : wi);
                        const int idx = gen ? iwin[u * 32 + wi] : g.Q / 2;
                        const float t = P[a.y.ct + ((size_t)k * g.Q + idx) * H + o];
                        acc += (rr >= 0) ? t : 0.f;
                    }
                }
                ring0[(u * R0 + pmod(q, R0)) * Hp + o] = acc / (1.f + fabsf(acc));
            }
            __syncthreads();

            float sacc[UBM][2];                              // this wave's (<= 2) skip rows, lane-partial, per utterance
#pragma unroll
            for (int u = 0; u < UBM; ++u) { sacc[u][0] = 0.f; sacc[u][1] = 0.f; }
            // skip contribution of hidden state `hl` (layer index l = 1..L: columns (l-1)*Hp ..) at the step's last position
            auto skip_add = [&](int l, int src_ring, bool from_hlast) {
                const int q = q0 + np - 1;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int lr = w * 2 + rr;
                    const bool live = lr < nsr;
                    for (int c0 = lane * 4; c0 < Hp; c0 += 256) {
                        const float4 wv = ld4(rP, live ? (unsigned)((a.y.wsk + (size_t)(sr0 + lr) * L * Hp + (size_t)(l - 1) * Hp + c0) * 4) : CL_OOB);
#pragma unroll
                        for (int u = 0; u < UBM; ++u) {
                            if (u < UB) {
                                const size_t base = (size_t)utt(u) * a.stride +
                                    (from_hlast ? (size_t)a.o_hlast : (size_t)a.ring_off[src_ring] + (size_t)pmod(q, a.ring_len[src_ring]) * Hp);
                                const float4 x = ld4_sc1(rS, (unsigned)((base + c0) * 4));
                                sacc[u][rr] = fmaf(wv.x, x.x, sacc[u][rr]); sacc[u][rr] = fmaf(wv.y, x.y, sacc[u][rr]);
                                sacc[u][rr] = fmaf(wv.z, x.z, sacc[u][rr]); sacc[u][rr] = fmaf(wv.w, x.w, sacc[u][rr]);
                            }
                        }
                    }
                }
            };

            // ---- stack
            for (int l = 0; l < nl; ++l) {
                const int nxt = (l + 1 < nl) ? l + 1 : 0;    // the layer whose rows stream in under this phase
                load_rows(nxt, wnext);
                const int dil = g.dil[l], R = a.ring_len[l];
                for (int u = 0; u < UB; ++u) {
                    const size_t ub = (size_t)utt(u) * a.stride;
                    for (int j = 0; j < np; ++j) {
                        const int q = q0 + j;
                        float4 xv[NI];
#pragma unroll
                        for (int pc = 0; pc < NI; ++pc) {
                            const int idx = pc * 256 + lane * 4;
                            const int ic = idx < KH ? idx : 0;
                            const int tap = ic / Hp, ii = ic - tap * Hp;
                            const int slotp = pmod(q - (K - 1 - tap) * dil, R);
                            if (l == 0) {
                                xv[pc] = idx < KH ? *reinterpret_cast<const float4*>(ring0 + (u * R0 + slotp) * Hp + ii)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
                            } else {
                                xv[pc] = ld4_sc1(rS, idx < KH ? (unsigned)((ub + a.ring_off[l] + (size_t)slotp * Hp + ii) * 4) : CL_OOB);
                            }
                        }
#pragma unroll
                        for (int rw = 0; rw < RW; ++rw) {
                            float acc = 0.f;
#pragma unroll
                            for (int pc = 0; pc < NI; ++pc) {
                                acc = fmaf(wcur[rw][pc].x, xv[pc].x, acc); acc = fmaf(wcur[rw][pc].y, xv[pc].y, acc);
                                acc = fmaf(wcur[rw][pc].z, xv[pc].z, acc); acc = fmaf(wcur[rw][pc].w, xv[pc].w, acc);
                            }
                            acc = sum64(acc);
                            if (lane == 0) rsum[(u * SEG_MAX + j) * 2 * CPC_MAX + w * RW + rw] = acc;
                        }
                    }
                }
                if (gen && l >= 1) skip_add(l, l, false);    // h_l (input of this layer) is visible: its out_skip share
                __syncthreads();
                // gate epilogue: one thread per (utterance, position, own channel)
                for (int e = tid; e < UB * np * nch; e += NT) {
                    const int c = e % nch, j = (e / nch) % np, u = e / (nch * np);
                    const int o = ch0 + c, q = q0 + j, b = utt(u);
                    const size_t ub = (size_t)b * a.stride;
                    float gz = P[a.y.bx + (size_t)l * H2 + o], gc = P[a.y.bx + (size_t)l * H2 + H + o];
                    const float* condb = a.cond + (size_t)b * a.Tf * g.N;
                    for (int s = 0; s < seg; ++s) {
                        int tt = q + s - g.rf; tt = tt < 0 ? 0 : tt;
                        int f = tt / g.U; const int jj = tt - f * g.U;
                        f = f < a.Tf ? f : a.Tf - 1;
                        const float wu = P[a.y.wup + jj];
                        const float* cr = condb + (size_t)f * g.N + (size_t)(l * seg + s) * H2;
                        gz = fmaf(wu, cr[o], gz); gc = fmaf(wu, cr[H + o], gc);
                    }
                    const float az = rsum[(u * SEG_MAX + j) * 2 * CPC_MAX + 2 * c] + P[a.y.bd + (size_t)l * H2 + o];
                    const float ac = rsum[(u * SEG_MAX + j) * 2 * CPC_MAX + 2 * c + 1] + P[a.y.bd + (size_t)l * H2 + H + o];
                    const float hp = l == 0 ? ring0[(u * R0 + pmod(q, R0)) * Hp + o]
                                            : ld1_sc1(rS, (unsigned)((ub + a.ring_off[l] + (size_t)pmod(q, R) * Hp + o) * 4));
                    const float z = sigm(gz * az);
                    const float cd = tanhf(gc * ac);
                    const float hn = (1.f - z) * cd + z * hp;
                    if (l + 1 < L) st1_sc1(rS, (unsigned)((ub + a.ring_off[l + 1] + (size_t)pmod(q, a.ring_len[l + 1]) * Hp + o) * 4), hn);
                    else if (j == np - 1) st1_sc1(rS, (unsigned)((ub + a.o_hlast + o) * 4), hn);
                }
                if (!cluster_barrier()) return;
#pragma unroll
                for (int rw = 0; rw < RW; ++rw)
#pragma unroll
                    for (int pc = 0; pc < NI; ++pc) wcur[rw][pc] = wnext[rw][pc];
            }
            if (!gen) continue;

            // ---- head: out_skip (finish) -> relu -> out_1 -> relu -> out_2
            skip_add(L, 0, true);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int lr = w * 2 + rr;
#pragma unroll
                for (int u = 0; u < UBM; ++u) {
                    if (u < UB) {
                        const float v = sum64(sacc[u][rr]);
                        if (lane == 0 && lr < nsr)
                            st1_sc1(rS, (unsigned)(((size_t)utt(u) * a.stride + a.o_skip + sr0 + lr) * 4), fmaxf(v + P[a.y.bsk + sr0 + lr], 0.f));
                    }
                }
            }
            if (!cluster_barrier()) return;
            // rows x vector products of the head: the full input vector is staged in LDS once per utterance
            auto stage_vec = [&](int off, int n4) {          // n4 = padded length in floats (multiple of 4)
                for (int e = tid; e < UB * (n4 >> 2); e += NT) {
                    const int u = e / (n4 >> 2), c4 = e - u * (n4 >> 2);
                    *reinterpret_cast<float4*>(vec + u * 512 + 4 * c4) = ld4_sc1(rS, (unsigned)(((size_t)utt(u) * a.stride + off + 4 * c4) * 4));
                }
                __syncthreads();
            };
            auto rows_times_vec = [&](size_t w_off, int ldw, size_t b_off, int r0, int nrows, int nin, int y_off, bool relu, bool to_lds) {
                // wave w takes local rows w, w+4, ...; lanes split the inputs
                for (int lr = w; lr < nrows; lr += 4) {
                    float acc[UBM];
#pragma unroll
                    for (int u = 0; u < UBM; ++u) acc[u] = 0.f;
                    for (int c0 = lane * 4; c0 < nin; c0 += 256) {
                        const float4 wv = ld4(rP, (unsigned)((w_off + (size_t)(r0 + lr) * ldw + c0) * 4));
#pragma unroll
                        for (int u = 0; u < UBM; ++u) {
                            if (u < UB) {
                                const float4 x = *reinterpret_cast<const float4*>(vec + u * 512 + c0);
                                acc[u] = fmaf(wv.x, x.x, acc[u]); acc[u] = fmaf(wv.y, x.y, acc[u]);
                                acc[u] = fmaf(wv.z, x.z, acc[u]); acc[u] = fmaf(wv.w, x.w, acc[u]);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < UBM; ++u) {
                        if (u < UB) {
                            float v = sum64(acc[u]);
                            if (lane == 0) {
                                v += P[b_off + r0 + lr];
                                v = relu ? fmaxf(v, 0.f) : v;
                                if (to_lds) o2v[u * swn_round4(g.NO) + r0 + lr] = v;
                                else st1_sc1(rS, (unsigned)(((size_t)utt(u) * a.stride + y_off + r0 + lr) * 4), v);
                            }
                        }
                    }
                }
            };
            stage_vec(a.o_skip, g.Sp);
            rows_times_vec(a.y.w1, g.Sp, a.y.b1, or0, nor, g.Sp, a.o_o1, true, false);
            if (!cluster_barrier()) return;
            stage_vec(a.o_o1, g.O1p);
            if (wide) {
                rows_times_vec(a.y.w2, g.O1p, a.y.b2, nr0, nnr, g.O1p, a.o_o2, false, false);
                if (!cluster_barrier()) return;
                for (int e = tid; e < UB * g.NO; e += NT) {
                    const int u = e / g.NO, r = e - u * g.NO;
                    o2v[u * swn_round4(g.NO) + r] = ld1_sc1(rS, (unsigned)(((size_t)utt(u) * a.stride + a.o_o2 + r) * 4));
                }
            } else {
                rows_times_vec(a.y.w2, g.O1p, a.y.b2, 0, g.NO, g.O1p, 0, false, true);      // every CU: all NO rows
            }
            __syncthreads();
            if (a.heads && slot == 0)
                for (int e = tid; e < UB * g.NO; e += NT) {
                    const int u = e / g.NO, r = e - u * g.NO;
                    a.heads[((size_t)utt(u) * a.n_steps + i) * g.NO + r] = o2v[u * swn_round4(g.NO) + r];
                }
            // ---- sampling, evaluated identically by every CU of the cluster (wave u handles utterances u, u+4)
            for (int u = w; u < UB; u += 4) {
                const int b = utt(u);
                const float* o2 = o2v + u * swn_round4(g.NO);
                if (KIND == SWN_KIND_LAPLACE) {
                    if (lane == 0) {
#pragma clang fp contract(off)
                        // Laplace head, cswnv_shift1.py:368-391
                        float* wn = win + u * 32;
                        const float* forced = reinterpret_cast<const float*>(a.forced);
                        float* outp = reinterpret_cast<float*>(a.out) + (size_t)b * a.n_steps * seg + (size_t)i * seg;
                        float lp[16], fed[16];
                        const int lpc = g.lpc;
                        for (int k = 0; k < lpc; ++k) lp[k] = wn[WN - lpc + k];
                        for (int j = 0; j < seg; ++j) {
                            const float mu = o2[j], yv = o2[seg + j];
                            const float bsc = expf(fminf(yv, 0.f) - log1pf(expf(-fabsf(yv))));
                            float lpv = 0.f;
                            for (int k = 0; k < lpc; ++k) lpv += o2[2 * seg + lpc - 1 - k] * lp[k];
                            const float e = swn_noise_laplace(a.nz, b, i, j, a.n_steps, seg);
                            const float sg = (e > 0.f) ? 1.f : ((e < 0.f) ? -1.f : 0.f);
                            const float t = (bsc * sg) * log1pf(-2.f * fabsf(e));
                            float sv = (lpc > 0) ? (lpv + mu) - t : mu - t;
                            sv = fminf(fmaxf(sv, -1.f), 1.f);
                            if (slot == 0) outp[j] = sv;
                            const float fd = forced ? forced[(size_t)b * a.n_steps * seg + (size_t)i * seg + j] : sv;
                            fed[j] = fd;
                            for (int k = 0; k + 1 < lpc; ++k) lp[k] = lp[k + 1];
                            if (lpc > 0) lp[lpc - 1] = fd;
                        }
                        for (int k = 0; k + seg < WN; ++k) wn[k] = wn[k + seg];
                        for (int j = 0; j < seg; ++j) wn[WN - seg + j] = fed[j];
                    }
                } else {
                    // softmax head, dswnv.py:361-369
                    const int Q = g.Q;
                    float m = -INFINITY;
                    for (int e = lane; e < Q; e += 64) m = fmaxf(m, o2[e]);
                    for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
                    float sum = 0.f;
                    for (int e = lane; e < Q; e += 64) sum += expf(o2[e] - m);
                    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d, 64);
                    float sum2 = 0.f;
                    for (int e = lane; e < Q; e += 64) sum2 += expf(o2[e] - m) / sum;
                    for (int d = 32; d >= 1; d >>= 1) sum2 += __shfl_xor(sum2, d, 64);
                    float best = -1.f; int bi = 0x7fffffff;
                    for (int e = lane; e < Q; e += 64) {
                        const float r = ((expf(o2[e] - m) / sum) / sum2) / swn_noise_exp1(a.nz, b, i, e, a.n_steps, Q);
                        if (r > best) { best = r; bi = e; }
                    }
                    for (int d = 32; d >= 1; d >>= 1) {
                        const float ob = __shfl_xor(best, d, 64);
                        const int oi = __shfl_xor(bi, d, 64);
                        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
                    }
                    if (lane == 0) {
                        const int* forced = reinterpret_cast<const int*>(a.forced);
                        if (slot == 0) reinterpret_cast<int*>(a.out)[(size_t)b * a.n_steps + i] = bi;
                        const int fd = forced ? forced[(size_t)b * a.n_steps + i] : bi;
                        int* wn = iwin + u * 32;
                        for (int k = 0; k + 1 < WN; ++k) wn[k] = wn[k + 1];
                        wn[WN - 1] = fd;
                    }
                }
            }
            __syncthreads();
        }
        if (!cluster_barrier()) return;                      // the next pass re-uses nothing of this one, but keep the clusters aligned
    }
}

// fills the outputs with NaN / -1 when the launch aborted (runs after the decode kernel on the same stream)
__global__ void cluster_verdict_kernel(const unsigned* ctrl, void* out, size_t n, int soft) {
    if (ctrl[CW_ABORT * CW_STRIDE] == 0u) return;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        if (soft) reinterpret_cast<int*>(out)[e] = -1;
        else reinterpret_cast<float*>(out)[e] = __builtin_nanf("");
    }
}

int plan(ClArgs& a) {
    const SwnGeom& g = a.g;
    int o = 0;
    for (int l = 0; l < g.L; ++l) { a.ring_off[l] = o; a.ring_len[l] = g.pad[l] + g.seg; o += a.ring_len[l] * g.Hp; }
    a.WN = (g.K - 1 > g.lpc ? g.K - 1 : g.lpc) + g.seg;
    a.o_hlast = o; o += g.Hp;
    a.o_skip = o; o += g.Sp;
    a.o_o1 = o; o += g.O1p;
    a.o_o2 = o; o += swn_round4(g.NO);
    a.stride = (o + 63) & ~63;
    return a.stride;
}

size_t lds_floats(const ClArgs& a) {
    const SwnGeom& g = a.g;
    return (size_t)UBM * a.ring_len[0] * g.Hp + (size_t)UBM * SEG_MAX * 2 * CPC_MAX + UBM * 32 + (size_t)UBM * swn_round4(g.NO) +
           (size_t)UBM * 512;
}

template <int NI, int RW>
int launch_cluster(const ClArgs& a, size_t lds, int grid, hipStream_t st) {
    if (a.g.kind == SWN_KIND_LAPLACE) {
        auto k = decode_cluster_kernel<NI, RW, SWN_KIND_LAPLACE>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SWN_E_LAUNCH;
        hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, a);
    } else {
        auto k = decode_cluster_kernel<NI, RW, SWN_KIND_SOFTMAX>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SWN_E_LAUNCH;
        hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, a);
    }
    return swn_launch_status("swn_decode(cluster)");
}

}  // namespace

extern "C" size_t swn_decode_cluster_state_floats(const swn_net_desc* d, int batch) {
    ClArgs a;
    if (swn_make_geom(d, &a.g) < 0 || batch < 1) return 0;
    return (size_t)plan(a) * batch + (size_t)CW_COUNT * CW_STRIDE + 64;
}

extern "C" int swn_decode_cluster(const swn_net_desc* d, const float* packed, const float* cond, int batch, int n_frames,
                                  int n_steps, const SwnNoise* nz, const void* forced, const void* seed, float* state,
                                  void* out, float* heads, void* stream_) {
    ClArgs a;
    int rc = swn_make_geom(d, &a.g);
    if (rc < 0) return rc;
    const SwnGeom& g = a.g;
    plan(a);
    const int ni = (g.K * g.Hp + 255) / 256;
    if (ni > 8 || g.seg > SEG_MAX || g.lpc > 16 || a.WN > 32 || g.audio_in || g.Sp > 512 || g.O1p > 512 || g.H > 256 ||
        (g.Hp & 3) || (g.Sp & 3))
        return SWN_E_UNSUPPORTED;
    if ((size_t)a.stride * batch * sizeof(float) >= (1ull << 31)) return SWN_E_UNSUPPORTED;     // 32-bit buffer offsets
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return SWN_E_NODEVICE;
    const int grid = prop.multiProcessorCount;                // one workgroup per CU: every CU joins its XCD's cluster
    if (grid < 8 * 8) return SWN_E_UNSUPPORTED;
    const size_t lds = lds_floats(a) * sizeof(float);
    // the LDS request also keeps a second workgroup off the CU (co-residency of the whole grid is what the barriers need)
    const size_t lds_req = lds > 84 * 1024 ? lds : 84 * 1024;
    if (lds_req > 160 * 1024) return SWN_E_UNSUPPORTED;
    swn_make_layout(&a.g, &a.y);
    a.P = packed; a.cond = cond; a.nz = *nz; a.forced = forced; a.seed = seed; a.state = state; a.out = out; a.heads = heads;
    a.B = batch; a.Tf = n_frames; a.n_steps = n_steps; a.n_pro = g.rf - g.seg + 1;
    const size_t st_floats = (size_t)a.stride * batch;
    a.ctrl = reinterpret_cast<unsigned*>(state + ((st_floats + 63) & ~(size_t)63));
    hipStream_t st = (hipStream_t)stream_;
    if (hipMemsetAsync(state, 0, sizeof(float) * (((st_floats + 63) & ~(size_t)63) + (size_t)CW_COUNT * CW_STRIDE), st) != hipSuccess)
        return SWN_E_LAUNCH;
    // rows per wave: 2 * ceil(H / (CUs per XCD)) rows over 4 waves
    const int cpc = (g.H + grid / 8 - 1) / (grid / 8);
    const int rw = (2 * cpc + 3) / 4;
    if (rw > 4) return SWN_E_UNSUPPORTED;
#define SWN_CL(NI_)                                                           \
    do {                                                                      \
        if (rw <= 1) rc = launch_cluster<NI_, 1>(a, lds_req, grid, st);       \
        else if (rw == 2) rc = launch_cluster<NI_, 2>(a, lds_req, grid, st);  \
        else if (rw == 3) rc = launch_cluster<NI_, 3>(a, lds_req, grid, st);  \
        else rc = launch_cluster<NI_, 4>(a, lds_req, grid, st);               \
    } while (0)
    if (ni <= 1) SWN_CL(1);
    else if (ni <= 6) SWN_CL(6);
    else SWN_CL(8);
#undef SWN_CL
    if (rc != SWN_OK) return rc;
    const size_t n_out = (size_t)batch * n_steps * (g.kind == SWN_KIND_SOFTMAX ? 1 : g.seg);
    hipLaunchKernelGGL(cluster_verdict_kernel, dim3(64), dim3(256), 0, st, a.ctrl, out, n_out, g.kind == SWN_KIND_SOFTMAX ? 1 : 0);
    return swn_launch_status("swn_decode(cluster verdict)");
}
