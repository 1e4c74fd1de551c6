// Cluster decode for LARGE geometries (reference-shipped REF6: H=192/256, K=7, 15-24 MB of weights per generated
// sample) on gfx950: ONE persistent launch, the CUs of one XCD form a cluster that carries an utterance (or up to 8 of
// them in lock step) through every phase of every step, handing the H-vector of each layer to one another through
// memory - SURVEY.md 7.3 option (i).
//
// Why: the stepped decode (swn_decode_stepped.hip) pays a kernel boundary plus two memory round trips per phase,
// 3.3-4.2 us x (L+3) launches per generated step.  Here
//   * a hand-off is ONE round trip: every handed-off value travels as an 8-byte granule {value, tag} written with a
//     write-through (sc1) store, and the consumer's sc1 load of the very data it needs is the poll - no barrier, no
//     flag, no second load (MI355X_MICROARCH.md "handoff-1to1": 0.8-1.0 us against 1.7-2.5x that for flag + payload);
//   * the dil_h rows a CU owns stay in its registers for the whole launch where they fit (REF6 Laplace: 378 of the 512
//     registers of a one-wave-per-SIMD lane), otherwise they stream two layers ahead;
//   * everything that does not depend on the hand-off (the K-1 old taps, conditioning, biases) is fetched before the
//     poll; sampling / out_2 / the input layer are evaluated redundantly by every CU and need no hand-off at all.
//
//   cluster     = the workgroups (one per CU, forced by the LDS request) that report the same HW_REG_XCC_ID; each takes
//                 a slot from a per-XCD ticket.  Nothing depends on the dispatch order.
//   partition   = slot s owns hidden channels [s*CPC, (s+1)*CPC) of every layer (gate + candidate rows), SPC rows of
//                 the concatenated out_skip 1x1 and OPC rows of out_1 (and of out_2 when it is wide: softmax).
//   tags        = iteration number + 1 (never 0: the state block starts zeroed).  A mailbox slot is written once per
//                 iteration; its next writer has, through the dependency chain of the phases, consumed something
//                 every reader of the old value produced after reading it, so no value is overwritten unread.  The
//                 float rings (old taps) take the same stores one instruction earlier: stores of a wave complete in
//                 order, so a reader that saw a later granule of a producer sees its ring stores too; rings carry seg
//                 spare slots because clusters may be one iteration apart.
//   safety      = every poll loop is bounded; a cluster that cannot assemble (workgroups not co-resident) raises the
//                 abort word, every workgroup leaves at its next poll and the outputs of the launch are filled with
//                 NaN / -1 so that the failure cannot pass for a result.
// The math, the ring layout and the noise are those of swn_decode.hip / swn_decode_stepped.hip (cswnv_shift1.py:281-430,
// dswnv.py:290-399).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "swn_geom.hpp"
#include "swn_noise.hpp"

namespace {

constexpr int NT = 256;              // 4 waves, one per SIMD: the full 512-entry register file per wave
constexpr int UBM = 8;               // utterances a cluster carries in lock step (per pass)
constexpr int NXCD = 8;
constexpr int CPC_MAX = 8;           // channels per CU (H <= 256 on >= 32 CUs)
constexpr int SEG_MAX = 10;
constexpr int LL = 6;                // stack depth this kernel is unrolled for
#ifndef SWN_XS_LAP
#define SWN_XS_LAP 0                 // XCDs per cluster (log2) at the run.sh Laplace geometry: 0, 1, 2 measured the same
                                     // step rate (35-38 us); one XCD per cluster keeps eight utterances in flight
#endif
constexpr unsigned SPIN_LIMIT = 1u << 22;
constexpr unsigned SC1 = 16;         // buffer-instruction cache policy bit: sc1
constexpr unsigned CL_OOB = 0x80000000u;

// control words, each on a 128-byte line of its own (32 uints apart)
enum { CW_TICKET = 0, CW_ARRIVED = 8, CW_ABORT = 9, CW_COUNT = 10 };
constexpr int CW_STRIDE = 32;

struct ClArgs {
    SwnGeom g;
    SwnLayout y;
    const float* P; const float* cond; SwnNoise nz; const void* forced; const void* seed;
    float* state; void* out; float* heads; unsigned* ctrl;
    int B, Tf, n_steps, n_pro, WN;
    int ring_off[SWN_MAXL], ring_len[SWN_MAXL];      // float ring l (l >= 1): input history of layer l (old taps)
    int mb_off[SWN_MAXL];                            // granule mailbox of layer l's OUTPUT: [seg][Hp] x {value, tag}
    int o_ms, o_m1, o_m2, stride;                    // granule vectors skip[Sp], out_1[O1p], out_2[NO]; per-utterance floats
    int xshift;                                      // a cluster spans 2^xshift XCDs (its rows must fit the CUs' registers)
};

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ int pmod(int r, int m) { int t = r % m; return t < 0 ? t + m : t; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cl_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float ld1(__amdgpu_buffer_rsrc_t r, unsigned off) {             // plain (read-only data)
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ float ld1_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {         // handed-off data
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, SC1));
}
#ifndef SWN_GRANULE_POLICY
#define SWN_GRANULE_POLICY SC1
#endif
__device__ __forceinline__ uint2 ldg_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) {         // one granule {value, tag}
    return __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, SWN_GRANULE_POLICY));
}
__device__ __forceinline__ void st1_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, 0, SC1);
}
__device__ __forceinline__ void stg_sc1(__amdgpu_buffer_rsrc_t r, unsigned off, float v, unsigned tag) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 gr; gr.x = __builtin_bit_cast(unsigned, v); gr.y = tag;
    __builtin_amdgcn_raw_buffer_store_b64(gr, r, off, 0, SC1);
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

// CL = channels per lane (ceil(H / 64)), KT = kernel size, RW = rows per wave, NSET = register sets for the dil_h rows
// (LL: resident; 3: streamed two layers ahead)
template <int CL, int KT, int RW, int NSET, int KIND>
__global__ __launch_bounds__(NT, 1) void decode_cluster_kernel(const ClArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_dead;
    __shared__ unsigned s_tickets[NXCD];
    const SwnGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = g.H, Hp = g.Hp, H2 = 2 * g.H, seg = g.seg, S = g.S, WN = a.WN;
    constexpr int K = KT, L = LL;
    const float* __restrict__ P = a.P;
    unsigned* abortw = a.ctrl + CW_ABORT * CW_STRIDE;

    // ---- cluster assembly: which XCD am I on, which slot do I get, how large did every cluster become
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc = (xcc & (NXCD - 1)) >> a.xshift;                  // cluster index: 2^xshift neighbouring XCDs form one cluster
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
    if (SPC > 8) { if (tid == 0) astore(abortw, 3u); return; }       // two skip rows per wave
    // utterances of this cluster: b = cidx, cidx + ncl, ...  ; a pass carries up to UBM of them
    const int n_mine = a.B > cidx ? (a.B - cidx + ncl - 1) / ncl : 0;

    // ---- LDS carve
    const int R0 = a.ring_len[0];
    float* ring0 = lds;                                      // [UBM][R0][Hp]  local history of the input layer
    float* rsum = ring0 + UBM * R0 * Hp;                     // [UBM][SEG_MAX][2*CPC_MAX] gate / candidate row sums
    float* hpv = rsum + UBM * SEG_MAX * 2 * CPC_MAX;         // [UBM][SEG_MAX][CPC_MAX]   highway inputs of the own channels
    float* win = hpv + UBM * SEG_MAX * CPC_MAX;              // [UBM][32] sample windows (float samples | int classes)
    float* o2v = win + UBM * 32;                             // [UBM][round4(NO)]
    float* vec = o2v + UBM * swn_round4(g.NO);               // [UBM][512] staging of a full skip / out_1 vector
    float* skl = vec + UBM * 512;                            // [UBM][4 waves][2 rows][64 lanes] lane-partial out_skip sums
    int* iwin = reinterpret_cast<int*>(win);

    const __amdgpu_buffer_rsrc_t rP = cl_rsrc(P), rS = cl_rsrc(a.state);
    const int c_lo = lane * CL;                              // this lane's channels c_lo .. c_lo+CL-1 of every H-vector
    unsigned spins_total = 0;
    auto give_up = [&]() __attribute__((always_inline)) { astore(abortw, 1u); s_dead = 1; };

    // CL granules of an H-vector mailbox at float offset `off` (+ this lane's channels): poll until every tag == tag.
    // Returns false when the launch is aborting.
    auto poll_vec = [&](size_t off, unsigned tag, float (&v)[CL]) __attribute__((always_inline)) -> bool {
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int c = 0; c < CL; ++c) {
                const bool live = c_lo + c < Hp;
                const uint2 gr = ldg_sc1(rS, live ? (unsigned)((off + 2 * (size_t)(c_lo + c)) * 4) : CL_OOB);
                v[c] = __builtin_bit_cast(float, gr.x);
                ok = ok && (!live || gr.y == tag);
            }
            if (__all(ok)) return true;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { give_up(); return false; }
            if ((spins & 255u) == 0u && aload(abortw) != 0u) { s_dead = 1; return false; }
        }
    };

    // dil_h rows of this wave: local row lr = w*RW + rw -> channel ch0 + (lr >> 1), gate | candidate
    float wr[NSET][RW][KT][CL];
    auto load_rows = [&](int l, float (&dst)[RW][KT][CL]) __attribute__((always_inline)) {
#pragma unroll
        for (int rw = 0; rw < RW; ++rw) {
            const int lr = w * RW + rw, c = lr >> 1;
            const bool live = c < nch;
            const size_t row = (size_t)l * H2 + (lr & 1 ? H : 0) + ch0 + c;
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int cc = 0; cc < CL; ++cc)
                    dst[rw][k][cc] = ld1(rP, (live && c_lo + cc < Hp) ? (unsigned)((a.y.wd + (row * K + k) * Hp + c_lo + cc) * 4) : CL_OOB);
        }
    };

    unsigned tag_base = 0;
    for (int pass0 = 0; pass0 < n_mine; pass0 += UBM) {
        const int UB = n_mine - pass0 < UBM ? n_mine - pass0 : UBM;
        auto utt = [&](int u) __attribute__((always_inline)) { return cidx + (pass0 + u) * ncl; };           // global utterance index
        // ---- per-pass initialisation: local rings and sample windows (the global state was zeroed by the host)
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
        if (pass0 == 0 || NSET < L) {
#pragma unroll
            for (int l = 0; l < (NSET < L ? 2 : L); ++l) load_rows(l, wr[l % NSET]);      // resident: once; streamed: layers 0, 1
        }

        const int total = a.n_pro + a.n_steps;
        for (int it = 0; it < total; ++it) {
            const bool gen = it >= a.n_pro;
            const int i = it - a.n_pro, np = gen ? seg : 1;
            const int q0 = gen ? g.rf + 1 - seg + i * seg : it;
            const int nl = gen ? L : L - 1;                  // the prologue never needs the last layer's output
            const unsigned tag = tag_base + (unsigned)it + 1u;
            // an opaque zero, re-defined every iteration and folded into every per-lane offset below: without it the
            // compiler hoists the address arithmetic of all six unrolled phases out of this loop (~200 registers on
            // top of the resident rows, i.e. spills); recomputing an offset costs two or three integer instructions
            int zi = 0;
            asm volatile("" : "+s"(zi));
            const int cz = c_lo + zi, ch0z = ch0 + zi, sr0z = sr0 + zi;

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

            // this wave's (<= 2) out_skip rows, lane-partial, per utterance: private LDS words of the lane
            if (gen) for (int u = 0; u < UB; ++u) { skl[((u * 4 + w) * 2 + 0) * 64 + lane] = 0.f; skl[((u * 4 + w) * 2 + 1) * 64 + lane] = 0.f; }
            // out_skip share of hidden state h_l (l = 1..L, column block l-1), held by the lanes as x[CL]; the weights of
            // this wave's two rows are fetched by skip_w ahead of the hand-off
            auto skip_w = [&](int l, float (&wv)[2][CL]) __attribute__((always_inline)) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int lr = w * 2 + rr;
                    const bool live = lr < nsr;
#pragma unroll
                    for (int c = 0; c < CL; ++c)
                        wv[rr][c] = ld1(rP, (live && cz + c < Hp) ? (unsigned)((a.y.wsk + (size_t)(sr0z + lr) * L * Hp + (size_t)(l - 1) * Hp + cz + c) * 4) : CL_OOB);
                }
            };
            auto skip_add = [&](int u, const float (&wv)[2][CL], const float (&x)[CL]) __attribute__((always_inline)) {
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    float acc = skl[((u * 4 + w) * 2 + rr) * 64 + lane];
#pragma unroll
                    for (int c = 0; c < CL; ++c) acc = fmaf(wv[rr][c], x[c], acc);
                    skl[((u * 4 + w) * 2 + rr) * 64 + lane] = acc;
                }
            };

            // ---- stack: one phase per layer, unrolled so that the register sets are named statically
            bool alive = true;
            auto phase = [&](auto lc) __attribute__((always_inline)) {
                constexpr int l = decltype(lc)::value;
                if (!alive || l >= nl) return;
                const int dil = g.dil[l], R = a.ring_len[l];
                float (&wl)[RW][KT][CL] = wr[l % NSET];
                // epilogue operands of the own channels: nothing here depends on the hand-off
                float e_gz = 0.f, e_gc = 0.f, e_bz = 0.f, e_bc = 0.f;
                const bool epi = tid < UB * np * nch;
                int e_c = 0, e_j = 0, e_u = 0;
                if (epi) {
                    const int tz = tid + zi;
                    e_c = tz % nch; e_j = (tz / nch) % np; e_u = tz / (nch * np);
                    const int o = ch0z + e_c, q = q0 + e_j, b = utt(e_u);
                    e_gz = P[a.y.bx + (size_t)l * H2 + o]; e_gc = P[a.y.bx + (size_t)l * H2 + H + o];
                    const float* condb = a.cond + (size_t)b * a.Tf * g.N;
                    for (int s = 0; s < seg; ++s) {
                        int tt = q + s - g.rf; tt = tt < 0 ? 0 : tt;
                        int f = tt / g.U; const int jj = tt - f * g.U;
                        f = f < a.Tf ? f : a.Tf - 1;
                        const float wu = P[a.y.wup + jj];
                        const float* cr = condb + (size_t)f * g.N + (size_t)(l * seg + s) * H2;
                        e_gz = fmaf(wu, cr[o], e_gz); e_gc = fmaf(wu, cr[H + o], e_gc);
                    }
                    e_bz = P[a.y.bd + (size_t)l * H2 + o]; e_bc = P[a.y.bd + (size_t)l * H2 + H + o];
                }
                float wsk_r[2][CL];
                if (gen && l >= 1) skip_w(l, wsk_r);
                for (int u = 0; u < UB && alive; ++u) {
                    const size_t ub = (size_t)utt(u) * a.stride;
                    for (int j = 0; j < np && alive; ++j) {
                        const int q = q0 + j;
                        float x[KT][CL];
                        // the K-1 older taps first (float ring / local ring: all of them are visible - a position of this
                        // very iteration was polled as the newest tap of an earlier j, and its ring store precedes its
                        // granule in the storing thread), then the hand-off: the newest tap
#pragma unroll
                        for (int k = 0; k < KT - (l > 0 ? 1 : 0); ++k) {
                            const int pos = q - (K - 1 - k) * dil;
#pragma unroll
                            for (int c = 0; c < CL; ++c) {
                                if (l == 0) x[k][c] = cz + c < Hp ? ring0[(u * R0 + pmod(pos, R0)) * Hp + cz + c] : 0.f;
                                else x[k][c] = ld1_sc1(rS, cz + c < Hp ? (unsigned)((ub + a.ring_off[l] + (size_t)pmod(pos, R) * Hp + cz + c) * 4) : CL_OOB);
                            }
                        }
                        if (l > 0) {
                            if (!poll_vec(ub + a.mb_off[l - 1] + 2 * (size_t)j * Hp, tag, x[KT - 1])) { alive = false; break; }
                        }
                        // highway inputs of the own channels: the lanes that hold them publish them for the epilogue
                        if (w == 0) {
#pragma unroll
                            for (int c = 0; c < CL; ++c) {
                                const int o = cz + c - ch0z;
                                if (o >= 0 && o < nch) hpv[(u * SEG_MAX + j) * CPC_MAX + o] = x[KT - 1][c];
                            }
                        }
                        if (gen && l >= 1 && j == np - 1) skip_add(u, wsk_r, x[KT - 1]);   // h_l at the step's last position
#pragma unroll
                        for (int rw = 0; rw < RW; ++rw) {
                            float acc = 0.f;
#pragma unroll
                            for (int k = 0; k < KT; ++k)
#pragma unroll
                                for (int c = 0; c < CL; ++c) acc = fmaf(wl[rw][k][c], x[k][c], acc);
                            acc = sum64(acc);
                            if (lane == 0) rsum[(u * SEG_MAX + j) * 2 * CPC_MAX + w * RW + rw] = acc;
                        }
                    }
                }
                // streamed rows: the layer two phases ahead takes the set the previous layer has finished with (set of
                // layer x = x % 3; a prologue iteration runs L-1 layers, a generation step L)
                if (NSET < L) {
                    if (gen) { constexpr int nx = (l + 2) % L; load_rows(nx, wr[nx % NSET]); }
                    else { constexpr int nx = (l + 2) % (L - 1); load_rows(nx, wr[nx % NSET]); }
                }
                __syncthreads();
                if (!alive || s_dead) { alive = false; return; }
                if (epi) {
                    const int o = ch0z + e_c, q = q0 + e_j;
                    const size_t ub = (size_t)utt(e_u) * a.stride;
                    const float az = rsum[(e_u * SEG_MAX + e_j) * 2 * CPC_MAX + 2 * e_c] + e_bz;
                    const float ac = rsum[(e_u * SEG_MAX + e_j) * 2 * CPC_MAX + 2 * e_c + 1] + e_bc;
                    const float hp = hpv[(e_u * SEG_MAX + e_j) * CPC_MAX + e_c];
                    const float z = sigm(e_gz * az);
                    const float cd = tanhf(e_gc * ac);
                    const float hn = (1.f - z) * cd + z * hp;
                    // float ring first (old taps of later iterations), then the granule: stores of a wave complete in order
                    if (l + 1 < L) st1_sc1(rS, (unsigned)((ub + a.ring_off[l + 1] + (size_t)pmod(q, a.ring_len[l + 1]) * Hp + o) * 4), hn);
                    stg_sc1(rS, (unsigned)((ub + a.mb_off[l] + 2 * ((size_t)e_j * Hp + o)) * 4), hn, tag);
                }
                __syncthreads();                             // rsum / hpv are reused by the next phase
            };
            phase(std::integral_constant<int, 0>{}); phase(std::integral_constant<int, 1>{});
            phase(std::integral_constant<int, 2>{}); phase(std::integral_constant<int, 3>{});
            phase(std::integral_constant<int, 4>{}); phase(std::integral_constant<int, 5>{});
            if (!alive || s_dead) return;
            if (!gen) continue;

            // ---- head: out_skip (finish with h_L) -> relu -> out_1 -> relu -> out_2
            {
                float wsk_r[2][CL];
                skip_w(L, wsk_r);
                for (int u = 0; u < UB; ++u) {
                    float x[CL];
                    if (!poll_vec((size_t)utt(u) * a.stride + a.mb_off[L - 1] + 2 * (size_t)(np - 1) * Hp, tag, x)) return;
                    skip_add(u, wsk_r, x);
                }
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int lr = w * 2 + rr;
                for (int u = 0; u < UB; ++u) {
                    const float v = sum64(skl[((u * 4 + w) * 2 + rr) * 64 + lane]);
                    if (lane == 0 && lr < nsr)
                        stg_sc1(rS, (unsigned)(((size_t)utt(u) * a.stride + a.o_ms + 2 * (size_t)(sr0 + lr)) * 4),
                                fmaxf(v + P[a.y.bsk + sr0 + lr], 0.f), tag);
                }
            }
            // full vector of n granules -> LDS staging; thread e polls granule e (n <= 512 -> two per thread at most)
            auto stage_vec = [&](int off, int n) __attribute__((always_inline)) -> bool {
                bool ok_all = true;
                for (int e = tid; e < UB * 512; e += NT) {
                    const int u = e >> 9, c = e & 511;
                    if (c < n) {
                        unsigned spins = 0;
                        for (;;) {
                            const uint2 gr = ldg_sc1(rS, (unsigned)(((size_t)utt(u) * a.stride + off + 2 * (size_t)c) * 4));
                            if (gr.y == tag) { vec[u * 512 + c] = __builtin_bit_cast(float, gr.x); break; }
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > SPIN_LIMIT) { give_up(); ok_all = false; break; }
                            if ((spins & 255u) == 0u && aload(abortw) != 0u) { s_dead = 1; ok_all = false; break; }
                        }
                    } else {
                        vec[u * 512 + c] = 0.f;
                    }
                }
                __syncthreads();
                return ok_all && !s_dead;
            };
            // rows [r0, r0 + nrows) of W (ld ldw, nin inputs) times the staged vectors; result: granule mailbox | LDS
            auto rows_times_vec = [&](size_t w_off, int ldw, size_t b_off, int r0, int nrows, int nin, int y_off, bool relu, bool to_lds) __attribute__((always_inline)) {
                for (int lr = w; lr < nrows; lr += 4) {       // wave w takes local rows w, w+4, ...; lanes split the inputs
                    float acc[UBM];
#pragma unroll
                    for (int u = 0; u < UBM; ++u) acc[u] = 0.f;
                    for (int c0 = lane * 4; c0 < nin; c0 += 256) {
                        const float4 wv = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rP, (unsigned)((w_off + (size_t)(r0 + lr) * ldw + c0) * 4), 0, 0));
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
                                else stg_sc1(rS, (unsigned)(((size_t)utt(u) * a.stride + y_off + 2 * (size_t)(r0 + lr)) * 4), v, tag);
                            }
                        }
                    }
                }
            };
            if (!stage_vec(a.o_ms, S)) return;
            rows_times_vec(a.y.w1, g.Sp, a.y.b1, or0, nor, g.Sp, a.o_m1, true, false);
            __syncthreads();
            if (!stage_vec(a.o_m1, g.O1)) return;
            if (wide) {
                rows_times_vec(a.y.w2, g.O1p, a.y.b2, nr0, nnr, g.O1p, a.o_m2, false, false);
                __syncthreads();
                if (!stage_vec(a.o_m2, g.NO)) return;
                for (int e = tid; e < UB * g.NO; e += NT) {
                    const int u = e / g.NO, r = e - u * g.NO;
                    o2v[u * swn_round4(g.NO) + r] = vec[u * 512 + r];
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
        tag_base += (unsigned)total + 1u;
    }
    (void)spins_total;
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
    // float rings carry seg spare slots: clusters run at most one iteration apart, and the producer of iteration t+1
    // must not reach the oldest tap a consumer of iteration t still reads
    for (int l = 0; l < g.L; ++l) { a.ring_off[l] = o; a.ring_len[l] = g.pad[l] + 2 * g.seg; if (l > 0) o += a.ring_len[l] * g.Hp; }
    a.ring_len[0] = g.pad[0] + g.seg;                        // layer 0's history is local (LDS)
    a.WN = (g.K - 1 > g.lpc ? g.K - 1 : g.lpc) + g.seg;
    for (int l = 0; l < g.L; ++l) { a.mb_off[l] = o; o += 2 * g.seg * g.Hp; }
    a.o_ms = o; o += 2 * g.Sp;
    a.o_m1 = o; o += 2 * g.O1p;
    a.o_m2 = o; o += 2 * swn_round4(g.NO);
    a.stride = (o + 63) & ~63;
    return a.stride;
}

size_t lds_floats(const ClArgs& a) {
    const SwnGeom& g = a.g;
    return (size_t)UBM * a.ring_len[0] * g.Hp + (size_t)UBM * SEG_MAX * 2 * CPC_MAX + (size_t)UBM * SEG_MAX * CPC_MAX + UBM * 32 +
           (size_t)UBM * swn_round4(g.NO) + (size_t)UBM * 512 + (size_t)UBM * 4 * 2 * 64;
}

template <int CL, int KT, int RW, int NSET>
int launch_cluster(const ClArgs& a, size_t lds, int grid, hipStream_t st) {
    if (a.g.kind == SWN_KIND_LAPLACE) {
        auto k = decode_cluster_kernel<CL, KT, RW, NSET, SWN_KIND_LAPLACE>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return SWN_E_LAUNCH;
        hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, a);
    } else {
        auto k = decode_cluster_kernel<CL, KT, RW, NSET, SWN_KIND_SOFTMAX>;
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
    if (g.L != LL || (g.K != 3 && g.K != 7) || g.H > 256 || g.seg > SEG_MAX || g.lpc > 16 || a.WN > 32 || g.audio_in ||
        g.S > 512 || g.O1 > 512 || g.NO > 512)
        return SWN_E_UNSUPPORTED;
    if ((size_t)a.stride * batch * sizeof(float) >= (1ull << 31)) return SWN_E_UNSUPPORTED;     // 32-bit buffer offsets
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return SWN_E_NODEVICE;
    const int grid = prop.multiProcessorCount;                // one workgroup per CU: every CU joins its XCD's cluster
    if (grid < 8 * 8) return SWN_E_UNSUPPORTED;
    const size_t lds = lds_floats(a) * sizeof(float);
    // the LDS request also keeps a second workgroup off the CU (co-residency of the whole grid is what the polls need)
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
    // channels per lane; XCDs per cluster chosen so that the dil_h rows a CU owns (rows/wave x K x CL x L registers per
    // lane) stay resident beside the ~200 registers of hoisted loop invariants: more CUs per cluster = fewer rows per CU.
    // (The kernel aborts loudly if its cluster came out smaller than an evenly filled chip gives.)
    const int cl = (g.Hp + 63) / 64;
    a.xshift = (g.K == 7 && cl == 3) ? SWN_XS_LAP : ((g.K == 7 && cl == 4) ? 2 : 0);
    const int ncu = (grid / 8) << a.xshift;                    // CUs per cluster
    const int cpc = (g.H + ncu - 1) / ncu;
    const int rw = (2 * cpc + 3) / 4;
    if (g.seg * cpc * UBM > NT) return SWN_E_UNSUPPORTED;     // one epilogue thread per (utterance, position, own channel)
    if ((g.S + ncu - 1) / ncu > 8) return SWN_E_UNSUPPORTED;
    rc = SWN_E_UNSUPPORTED;
    if (g.K == 3 && cl == 1 && rw <= 1) rc = launch_cluster<1, 3, 1, LL>(a, lds_req, grid, st);        // tiny fixtures (H <= 64)
    else if (g.K == 7 && cl == 3 && rw <= 1) rc = launch_cluster<3, 7, 1, LL>(a, lds_req, grid, st);   // run.sh Laplace: 128 CUs, 2 channels each
    else if (g.K == 7 && cl == 3 && rw <= 2) rc = launch_cluster<3, 7, 2, LL>(a, lds_req, grid, st);   // 64 CUs, 3 channels each
    else if (g.K == 7 && cl == 3 && rw <= 3) rc = launch_cluster<3, 7, 3, LL>(a, lds_req, grid, st);   // 32 CUs, 6 channels each
    else if (g.K == 7 && cl == 4 && rw <= 1) rc = launch_cluster<4, 7, 1, LL>(a, lds_req, grid, st);   // run.sh softmax: 128 CUs, 2 channels each
    if (rc != SWN_OK) return rc;
    const size_t n_out = (size_t)batch * n_steps * (g.kind == SWN_KIND_SOFTMAX ? 1 : g.seg);
    hipLaunchKernelGGL(cluster_verdict_kernel, dim3(64), dim3(256), 0, st, a.ctrl, out, n_out, g.kind == SWN_KIND_SOFTMAX ? 1 : 0);
    return swn_launch_status("swn_decode(cluster verdict)");
}
