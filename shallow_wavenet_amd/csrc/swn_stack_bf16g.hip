// bf16 MFMA teacher-forced stack for the LARGE geometries (reference-shipped REF6: H = 192 / 256, K = 7): the
// MFMA-bound side of BASELINE cfg4 (AI ~ 630 flop/B at REF6).  The BL6 class (H = 64, K = 2: HBM-bound) has its own
// register-resident kernels in swn_stack_bf16.hip; here every layer is a real GEMM
//     D[2H rows][positions] = Wd[2H][K*H] . Xcol[K*H][positions],  Xcol(tap, i; t) = h[t - (K-1-tap) dil][i]
// and so are skip (K = L*H), out_1 and out_2.  One tiled kernel serves them all:
//   * workgroup tile 128 rows x 128 positions, 4 waves of 64 x 64 (16 accumulators of v_mfma_f32_16x16x32_bf16),
//     k-tiles of 32 staged in LDS (rows padded to 80 B: conflict-free 16-byte fragment reads), the next k-tile's
//     global loads in flight under the MFMAs of the current one;
//   * activations stay bf16 time-major [layer][b][t][H] (a B tile row = 32 consecutive channels of one position
//     = one 64-byte segment); the causal shift of a tap is an index offset, positions before 0 read zeros;
//   * layer epilogue = the gate: the tile's rows are ordered [gate 16 | cand 16 | gate 16 | cand 16] per wave so a
//     lane holds the gate and candidate pre-activations of the same 8 channels; hoisted conditioning, sigmoid /
//     tanh and the highway mix in fp32, bf16 store.  Other epilogues: bias + relu -> bf16 time-major (skip,
//     out_1), bias -> fp32 (B, n_out, Tp) (out_2).
// fp32 accumulation everywhere; parameters are the bf16 roundings of the packed fp32 weights.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include "swn_geom.hpp"

namespace {

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

constexpr int TM = 128, TK = 32;
constexpr int PITCH = 40;        // bf16 elements per LDS row (32 + 8 pad = 80 bytes)

__device__ __forceinline__ unsigned short f2bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
// z = sigmoid(vz), c = tanh(vc) for the gate epilogue: two v_exp_f32 and ONE v_rcp_f32 (the reciprocal of the product of the two
// denominators serves both quotients) instead of the library tanhf and two IEEE divisions - ~20 instructions per channel-position
// where the epilogue of a 128 x 128 tile had ~80; 1-2 ulp of fp32, far inside the bf16 rounding of the hidden state it feeds.
// e_z is capped at 2^100 (sigmoid below 1e-30 reads as 2^-100: nothing a bf16 state can tell from 0) so the product stays finite;
// tanh uses |vc| (its exponential is <= 1) and takes the sign back.
__device__ __forceinline__ void gate_zc(const float vz, const float vc, float& z, float& c) {
    const float ez = fminf(__builtin_amdgcn_exp2f(vz * -1.44269504f), 1.2676506e30f);
    const float ec = __builtin_amdgcn_exp2f(fabsf(vc) * -2.88539008f);
    const float dz = 1.f + ez, dc = 1.f + ec;
    const float r = __builtin_amdgcn_rcpf(dz * dc);
    z = dc * r;
    c = copysignf((1.f - ec) * (dz * r), vc);
}

enum { EPI_GATE = 0, EPI_RELU_BF16 = 1, EPI_F32 = 2, EPI_BF16 = 3 };   // EPI_BF16: bias (optional), no relu -> bf16 time-major

struct GemmArgs {
    const unsigned short* A; int M, Kd;            // bf16 [M][Kd] row-major
    const unsigned short* src; size_t blk_stride; size_t src_bytes;  // B operand (src_bytes < 2 GiB): block blk, channel i, position t:
    int KB, nblk, shift0, shift_step;              //   src[blk*blk_stride + (b*Tp + t - (shift0 - blk*shift_step))*KB + i]
    int Tp, B;
    // epilogues
    const float* P; const float* bias;             // packed fp32 parameters; bias of plain GEMMs
    unsigned short* out_bf; int out_ld;            // RELU_BF16: out_bf[(b*Tp + t)*out_ld + m]
    float* out_f; int NO;                          // F32: out_f[(b*NO + m)*Tp + t]
    // gate
    const unsigned short* hprev; unsigned short* hnext; const float* cond;
    size_t o_bd, o_bx, o_wup, o_wxa;
    int H, l, seg, U, Tf, N, coff;
    const int* aidx; int Q;                        // softmax audio_in: class index of position t, or null
    int ntt, nmt;                                  // time tiles per utterance, row tiles (set by launch_gemm)
    float* a_out;                                  // gate: where to keep the pre-activations (B, 2H, Tp) fp32 in the G4 layout for the backward, or null
    const float* gx; size_t o_bxr; int gx_rows;    // gate, dropout mode: sample-rate in_x products (B, gx_rows = L*2H, Tp) fp32, G4 layout (row l*2H + o)
                                                   // instead of the hoisted cond, + the raw in_x bias (cswnv_shift1.py:194-198,269-278)
};

// ---- gate epilogue, shared by the two layer kernels.  A lane finishes 4 consecutive channels ch .. ch+3 (gate pre-activations az,
// candidate pre-activations ac) of one position t: hoisted conditioning (or the dropout mode's sample-rate in_x rows), sigmoid / tanh,
// the highway mix, one 8-byte bf16 store; the kept pre-activations of the training mode leave as two 16-byte pieces (G4 layout).
struct GateCh { float4 bxz, bxc, bdz, bdc; };       // per-channel constants of a lane's group, loaded once per group
__device__ __forceinline__ GateCh gate_consts(const GemmArgs& a, const int ch) {
    const int H = a.H, H2 = 2 * a.H;
    const size_t ob = (a.gx ? a.o_bxr : a.o_bx) + (size_t)a.l * H2 + ch;
    GateCh k;
    k.bxz = *reinterpret_cast<const float4*>(a.P + ob); k.bxc = *reinterpret_cast<const float4*>(a.P + ob + H);
    k.bdz = *reinterpret_cast<const float4*>(a.P + a.o_bd + (size_t)a.l * H2 + ch);
    k.bdc = *reinterpret_cast<const float4*>(a.P + a.o_bd + (size_t)a.l * H2 + H + ch);
    return k;
}
// conditioning frame and upsampler tap of the positions tt, tt + 16, ...: ONE integer division (~30 instructions), the others by
// stepping (a lane of the 384-row kernel had six of them per tile)
template <int NJ>
__device__ __forceinline__ void gate_frames(const int tt, const int U, int (&fj)[NJ], int (&jj0)[NJ]) {
    int f = tt / U, jj = tt - f * U;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        fj[j] = f; jj0[j] = jj;
        jj += 16;
        while (jj >= U) { jj -= U; ++f; }
    }
}

// One channel group (4 channels from ch) at NJ positions tb, tb + 16, ...: EVERY operand load of the group is issued before the
// first store - written position by position, each store to hnext stood between the next position's loads and their use (the
// compiler cannot know that hnext aliases none of them), i.e. one memory round trip per position and group: 18 in a row for a lane
// of the 384-row kernel, with nothing else on the CU to hide them.  Loads of positions past the end are clamped (never stored).
// (fj, jj0) = conditioning frame and upsampler tap of position t + coff (one division per position, done by the caller).
template <int NJ>
__device__ __forceinline__ void gate_group(const GemmArgs& a, const int b, const int ch, const int tb, const int (&fj)[NJ],
                                           const int (&jj0)[NJ], const f32x4 (&az)[NJ], const f32x4 (&ac)[NJ]) {
    const float* P = a.P;
    const int H = a.H, H2 = 2 * a.H, l = a.l;
    const GateCh k = gate_consts(a, ch);
    const float bz[4] = {k.bdz.x, k.bdz.y, k.bdz.z, k.bdz.w}, bc[4] = {k.bdc.x, k.bdc.y, k.bdc.z, k.bdc.w};
    float gz[NJ][4], gc[NJ][4];
    int tc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int t = tb + 16 * j;
        tc[j] = t < a.Tp ? t : a.Tp - 1;
        gz[j][0] = k.bxz.x; gz[j][1] = k.bxz.y; gz[j][2] = k.bxz.z; gz[j][3] = k.bxz.w;
        gc[j][0] = k.bxc.x; gc[j][1] = k.bxc.y; gc[j][2] = k.bxc.z; gc[j][3] = k.bxc.w;
    }
    if (a.gx) {                                   // G4 layout (swn_geom.hpp): the lane's four channels are one 16-byte piece
        float4 gz4[NJ], gc4[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            gz4[j] = *reinterpret_cast<const float4*>(a.gx + swn_g4(a.gx_rows, a.Tp, b, l * H2 + ch, tc[j]));
            gc4[j] = *reinterpret_cast<const float4*>(a.gx + swn_g4(a.gx_rows, a.Tp, b, l * H2 + H + ch, tc[j]));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            gz[j][0] += gz4[j].x; gz[j][1] += gz4[j].y; gz[j][2] += gz4[j].z; gz[j][3] += gz4[j].w;
            gc[j][0] += gc4[j].x; gc[j][1] += gc4[j].y; gc[j][2] += gc4[j].z; gc[j][3] += gc4[j].w;
        }
    } else {
        int f[NJ], jj[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { f[j] = fj[j]; jj[j] = jj0[j]; }
        for (int s = 0; s < a.seg; ++s) {
            float wu[NJ]; float4 cz[NJ], cc[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int fc = f[j] < a.Tf ? f[j] : a.Tf - 1;
                wu[j] = P[a.o_wup + jj[j]];
                const float* cr = a.cond + ((size_t)b * a.Tf + fc) * a.N + (size_t)(l * a.seg + s) * H2 + ch;
                cz[j] = *reinterpret_cast<const float4*>(cr); cc[j] = *reinterpret_cast<const float4*>(cr + H);
                if (++jj[j] >= a.U) { jj[j] = 0; ++f[j]; }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                gz[j][0] = fmaf(wu[j], cz[j].x, gz[j][0]); gz[j][1] = fmaf(wu[j], cz[j].y, gz[j][1]);
                gz[j][2] = fmaf(wu[j], cz[j].z, gz[j][2]); gz[j][3] = fmaf(wu[j], cz[j].w, gz[j][3]);
                gc[j][0] = fmaf(wu[j], cc[j].x, gc[j][0]); gc[j][1] = fmaf(wu[j], cc[j].y, gc[j][1]);
                gc[j][2] = fmaf(wu[j], cc[j].z, gc[j][2]); gc[j][3] = fmaf(wu[j], cc[j].w, gc[j][3]);
            }
        }
    }
    if (a.aidx) {                                 // one-hot audio input columns of in_x (dswnv.py:255-256)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            int idx = a.aidx[(size_t)b * a.Tp + tc[j]] % a.Q; idx = idx < 0 ? idx + a.Q : idx;
            const float* wa = P + a.o_wxa + ((size_t)l * a.Q + idx) * H2 + ch;
            const float4 az4 = *reinterpret_cast<const float4*>(wa), ac4 = *reinterpret_cast<const float4*>(wa + H);
            gz[j][0] += az4.x; gz[j][1] += az4.y; gz[j][2] += az4.z; gz[j][3] += az4.w;
            gc[j][0] += ac4.x; gc[j][1] += ac4.y; gc[j][2] += ac4.z; gc[j][3] += ac4.w;
        }
    }
    uint2 hp2[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) hp2[j] = *reinterpret_cast<const uint2*>(a.hprev + ((size_t)b * a.Tp + tc[j]) * H + ch);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int t = tb + 16 * j;
        if (t >= a.Tp) continue;
        const float hp[4] = {bf2f((unsigned short)(hp2[j].x & 0xffff)), bf2f((unsigned short)(hp2[j].x >> 16)),
                             bf2f((unsigned short)(hp2[j].y & 0xffff)), bf2f((unsigned short)(hp2[j].y >> 16))};
        if (a.a_out) {     // training: the backward reads these instead of recomputing the dilated conv (G4 layout: 16-byte pieces)
            *reinterpret_cast<f32x4*>(a.a_out + swn_g4(H2, a.Tp, b, ch, t)) = az[j];
            *reinterpret_cast<f32x4*>(a.a_out + swn_g4(H2, a.Tp, b, H + ch, t)) = ac[j];
        }
        unsigned short hv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float z, c;
            gate_zc(gz[j][r] * (az[j][r] + bz[r]), gc[j][r] * (ac[j][r] + bc[r]), z, c);
            hv[r] = f2bf((1.f - z) * c + z * hp[r]);
        }
        uint2 o; o.x = hv[0] | ((unsigned)hv[1] << 16); o.y = hv[2] | ((unsigned)hv[3] << 16);
        *reinterpret_cast<uint2*>(a.hnext + ((size_t)b * a.Tp + t) * H + ch) = o;
    }
}

// Three workgroups per CU (launch bound: 138 registers, accumulators included) with two register stages beat two workgroups with
// three stages (208 registers): 296 -> 274 us per gated layer at REF6 - these loops wait on memory round trips, and a third
// workgroup hides more of them than a third stage.
template <int EPI, int WNT>      // WNT: 16-column accumulator tiles per wave; the workgroup tile is 128 rows x 32*WNT positions
__global__ __launch_bounds__(256, 3) void bf16g_gemm_kernel(const GemmArgs a) {
    constexpr int TNW = 32 * WNT;
    __shared__ __attribute__((aligned(16))) unsigned short As[TM * PITCH];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[TNW * PITCH];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    // XCD-aware order (1-D grid; workgroup id i runs on XCD i % 8): each XCD walks a contiguous range of time tiles, the
    // row tiles of one time tile back to back: they read the same activations, and neighbouring time tiles share their
    // tap halos ((K-1)*dil positions) in that XCD's L2 instead of fetching them into eight L2s.
    const int chunk = (a.ntt * a.B + 7) / 8;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int ttl = idx / a.nmt, by = idx - ttl * a.nmt;
    const int gt = xcd * chunk + ttl;
    if (ttl >= chunk || gt >= a.ntt * a.B) return;
    const int b = gt / a.ntt;
    const int t0 = (gt - b * a.ntt) * TNW, m0 = by * TM;
    // tile row -> matrix row
    auto rowmap = [&](int tr) -> int {
        // bf16 time-major outputs: tile row 16 i + 4 g + r of a wave's 64 holds matrix row 16 g + 4 i + r, so that the 16 values a
        // lane finishes for its position are 16 consecutive outputs (two 16-byte stores; the four lane groups complete a
        // 128-byte row segment - in tile order they were 8-byte stores, 32 contiguous bytes per position)
        if (EPI == EPI_RELU_BF16 || EPI == EPI_BF16) return m0 + (tr & ~63) + 16 * ((tr >> 2) & 3) + 4 * ((tr >> 4) & 3) + (tr & 3);
        if (EPI != EPI_GATE) return m0 + tr;
        const int c0 = by * 64, wmr = tr >> 6, mt = (tr & 63) >> 4, r = tr & 15;
        const int ch = c0 + 32 * wmr + 16 * (mt >> 1) + r;
        return ch < a.H ? (mt & 1) * a.H + ch : -1;
    };
    // staging: thread -> (row = tid/4 (+64), 16-byte piece = tid%4) of a 64-byte k-tile row, for A and for B: four consecutive lanes
    // fetch the 64 contiguous bytes of one row (one request; with a lane pair per row - two pieces 32 bytes apart per instruction -
    // every piece was a request of its own)
    const int sr = tid >> 2, sq = tid & 3;
    int arow[2]; bool aok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) { arow[h] = rowmap(sr + 64 * h); aok[h] = arow[h] >= 0 && arow[h] < a.M; }
    const int tpos = t0 + sr;
    const int ktiles_per_blk = a.KB / TK, nk = a.nblk * ktiles_per_blk;
    constexpr int NST = 2;                      // k-tiles in flight per thread (register stages)
    constexpr int BQ = WNT / 4;                 // 128-row passes over the B tile
    uint4 ra[NST][2], rb[NST][2 * BQ];
    // branch-free AND select-free: operands come through buffer resources, and everything that must read as zero
    // (a k-tile past the end of the padded k loop, a row past M, a position outside the sequence) is an
    // out-of-range offset.  Any branch around the loads - or any use of a loaded value right behind its load -
    // makes the compiler drain the whole prefetch queue (vmcnt(0)) at every k-tile.
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.A), 0, (unsigned)((size_t)a.M * a.Kd * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.src), 0, (unsigned)a.src_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned a_off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) a_off[h] = aok[h] ? (unsigned)(((size_t)arow[h] * a.Kd + 8 * sq) * 2) : OOB;
    auto ld16 = [&](__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); };
    auto fetch = [&](int kt, uint4 (&qa)[2], uint4 (&qb)[2 * BQ]) {
        const bool live = kt < nk;
        const int blk = kt / ktiles_per_blk, kin = (kt - blk * ktiles_per_blk) * TK;
#pragma unroll
        for (int h = 0; h < 2; ++h) qa[h] = ld16(rA, (live && aok[h]) ? a_off[h] + (unsigned)kt * (TK * 2) : OOB);
#pragma unroll
        for (int q = 0; q < 2 * BQ; ++q) {
            const int tp = tpos + 64 * q, ts = tp - (a.shift0 - blk * a.shift_step);
            const bool bok = live && tp < a.Tp && ts >= 0;
            qb[q] = ld16(rB, bok ? (unsigned)(((size_t)blk * a.blk_stride + ((size_t)b * a.Tp + ts) * a.KB + kin + 8 * sq) * 2) : OOB);
        }
    };
    f32x4 acc[4][WNT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NST; ++u) fetch(u, ra[u], rb[u]);
    for (int kt0 = 0; kt0 < nk; kt0 += NST) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int kt = kt0 + u;                          // tiles past nk are all-zero operands: no branch in the loop
            *reinterpret_cast<uint4*>(As + sr * PITCH + 8 * sq) = ra[u][0];
            *reinterpret_cast<uint4*>(As + (sr + 64) * PITCH + 8 * sq) = ra[u][1];
#pragma unroll
            for (int q = 0; q < 2 * BQ; ++q) *reinterpret_cast<uint4*>(Bs + (sr + 64 * q) * PITCH + 8 * sq) = rb[u][q];
            __syncthreads();
            fetch(kt + NST, ra[u], rb[u]);             // refill the stage just consumed: NST k-tiles of loads in flight
            bf16x8 af[4], bfr[WNT];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(As + (64 * wm + 16 * i + (lane & 15)) * PITCH + 8 * (lane >> 4));
#pragma unroll
            for (int j = 0; j < WNT; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8*>(Bs + (16 * WNT * wn + 16 * j + (lane & 15)) * PITCH + 8 * (lane >> 4));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < WNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            __syncthreads();
        }
    }
    // accumulator element (i, j, r): tile row 64 wm + 16 i + 4 (lane/16) + r, tile column 64 wn + 16 j + lane%16
    const int g4 = lane >> 4, n = lane & 15;
    if (EPI == EPI_GATE) {
        // Loop order: channel group outside, positions inside - the 16 per-channel constants of a lane are loaded once per group
        // (inside the position loop every store to hnext stood between them and their reuse), and the conditioning frame of a
        // position comes from one division per position, not one per (position, group, segment tap).
        int fj[WNT], jj0[WNT];
        gate_frames<WNT>(t0 + 16 * WNT * wn + n + a.coff, a.U, fj, jj0);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int ch = by * 64 + 32 * wm + 16 * p + 4 * g4;          // 4 consecutive channels ch..ch+3
            if (ch >= a.H) continue;
#pragma unroll
            for (int j0 = 0; j0 < WNT; j0 += 2)      // two positions per batch: four would spill under the three-workgroup register bound
                gate_group<2>(a, b, ch, t0 + 16 * WNT * wn + 16 * j0 + n, reinterpret_cast<const int (&)[2]>(fj[j0]), reinterpret_cast<const int (&)[2]>(jj0[j0]),
                              reinterpret_cast<const f32x4 (&)[2]>(acc[2 * p][j0]), reinterpret_cast<const f32x4 (&)[2]>(acc[2 * p + 1][j0]));
        }
    } else {
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
            const int t = t0 + 16 * WNT * wn + 16 * j + n;
            if (t >= a.Tp) continue;
            if (EPI == EPI_RELU_BF16 || EPI == EPI_BF16) {
                const int mb = m0 + 64 * wm + 16 * g4;                       // 16 consecutive outputs of this lane (rowmap)
                unsigned short* orow = a.out_bf + ((size_t)b * a.Tp + t) * a.out_ld + mb;
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
                    unsigned wv[4];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int i = 2 * ip + q, m = mb + 4 * i;
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[r] = acc[i][j][r] + ((a.bias && m + r < a.M) ? a.bias[m + r] : 0.f);
                            if (EPI == EPI_RELU_BF16) v[r] = fmaxf(v[r], 0.f);
                        }
                        wv[2 * q] = f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16); wv[2 * q + 1] = f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                    }
                    const int m = mb + 8 * ip;                                // M % 4 == 0
                    if (m + 8 <= a.M && (a.out_ld & 7) == 0) *reinterpret_cast<uint4*>(orow + 8 * ip) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
                    else {
                        if (m < a.M) *reinterpret_cast<uint2*>(orow + 8 * ip) = make_uint2(wv[0], wv[1]);
                        if (m + 4 < a.M) *reinterpret_cast<uint2*>(orow + 8 * ip + 4) = make_uint2(wv[2], wv[3]);
                    }
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + 64 * wm + 16 * i + 4 * g4;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (m + r < a.NO) a.out_f[((size_t)b * a.NO + m + r) * a.Tp + t] = acc[i][j][r] + (a.bias ? a.bias[m + r] : 0.f);
            }
        }
    }
}

// ---- gated layer of the H = 192 geometry (run.sh Laplace net: M = 2H = 384 rows, K = 7 x 192) with LDS-DMA staging ----------
// The 128 x 128 kernel above stages through registers behind two barriers per k-tile and hides its waits behind three workgroups
// per CU; what it cannot hide is that every byte passes a VGPR and an LDS store, and that M = 384 re-reads each activation tile
// three times.  Here ONE 512-thread workgroup per CU owns all 384 rows of a 32 NCW-position tile:
//   * operands go global -> LDS without touching a register (`buffer_load_dwordx4 ... lds`, dma_piece below): a stage is one
//     k-tile of 32 (24 + 2 NCW pieces of 1 KiB = 16 rows of 64 bytes, unpadded); the k-octets of a row are permuted at the source
//     so that the fragment reads are conflict-free (see the lane maps in the kernel);
//   * four stages in a ring, three in flight, ONE barrier per stage: a wave waits for its own pieces of stage kt with a counted
//     vmcnt (the two younger stages stay in flight), the barrier publishes everybody's, and the stage after next goes into the slot
//     the barrier has just freed.  Stages past the end are issued all out-of-range (they land as zeros in a free slot), so the
//     count is the same in every iteration;
//   * wave (wm, wn) owns rows [96 wm, 96 wm + 96) = channels [48 wm, 48 wm + 48) as [gate 16 | cand 16] x 3 and NCW column
//     fragments: 6 + NCW fragment reads feed 6 NCW MFMAs per stage (the 64 x 64 wave tile above: 8 reads for 16);
//   * causal zero padding and the ragged last tile are out-of-range lanes of the DMA (zeros), as everywhere in this file.
// NCW = 6 (192 positions, 144 KB of LDS) or 4 (128 positions, 128 KB): the launcher takes whichever wastes fewer tile rounds.
typedef int g8_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void g8_dma(const g8_v4i rsrc, unsigned voff, unsigned soff, unsigned lds_base) {
    // lane i's 16 bytes land at lds_base + 16 i; the range check is on voff (out of range: zeros); M0 carries the LDS base
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane((int)soff)),
                    "s"(__builtin_amdgcn_readfirstlane((int)lds_base)) : "memory");
}
__device__ __forceinline__ g8_v4i g8_rsrc(const void* p, size_t bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    g8_v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)(a & 0xffffffffull));
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffull));
    r.z = __builtin_amdgcn_readfirstlane((int)(unsigned)bytes);
    r.w = 0x00020000;
    return r;
}
// NW = 8: one 512-thread workgroup per CU owns all 384 rows (four ring stages); NW = 4: a 256-thread workgroup owns HALF the rows (96
// channels, gate and candidate) of the same tile with a three-stage ring of its own, TWO workgroups per CU: same waves per SIMD, same
// wave tile, but the two workgroups are not in step - the gate epilogue of one (a third of a tile's time, pure vector work) runs
// under the MFMAs of the other, and so do the issue slots its DMA pieces hold.
template <int NCW, int NW>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void bf16g_gate8_kernel(const GemmArgs a, const int ntiles) {
    constexpr int BN = 32 * NCW, MW = NW / 2, AB = 6 * MW, BB = BN / 16, MH = 8 / NW, NST = NW == 8 ? 4 : 3;
    constexpr int NBP = (BB + NW - 1) / NW, NP = 3 + NBP;                    // B pieces / all pieces of a stage per wave (at most)
    constexpr unsigned STB = (unsigned)(AB + BB) * 1024u;
    constexpr unsigned OOB = 0x80000000u;
    static_assert(NP <= 6, "one piece per MFMA row");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // ALL LDS of the kernel is this one block
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;
    const int n = lane & 15, g4 = lane >> 4;
    // DMA lane i fetches row i >> 2, k-octet (i & 3) ^ G8_SW(row >> 2) of its piece: four consecutive lanes = the 64 contiguous bytes
    // of one row (one request; row i & 15 per lane made 64 requests of 16 bytes per instruction and ran at half the speed), and
    // the octet permutation makes the fragment read - lane (n, g4) at 64 n + 16 (g4 ^ G8_SW(n >> 2)) - conflict-free in each of
    // the four 16-lane groups a ds_read_b128 is served in (MI355X_MICROARCH.md, LDS table)
    const int dr = lane >> 2, dk = ((lane & 3) ^ ((0x1320 >> (4 * (dr >> 2))) & 3)) * 8;       // G8_SW = {0, 2, 3, 1}
    const unsigned rdl = (unsigned)(64 * n + 16 * (g4 ^ ((0x1320 >> (4 * (n >> 2))) & 3)));
    // XCD-aware order (workgroup id i runs on XCD i % 8): each XCD walks a contiguous range of time tiles, the row halves of a tile
    // back to back (they stage the same activations)
    const int nwork = ntiles * MH, chunk = (nwork + 7) / 8;
    const int gw = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= chunk || gw >= nwork) return;
    const int gt = gw / MH, cb = (gw - gt * MH) * 48 * MW;                   // time tile, first channel of this workgroup
    const int b = gt / a.ntt, t0 = (gt - b * a.ntt) * BN;
    const int H = a.H;
    const g8_v4i rA = g8_rsrc(a.A, (size_t)a.M * a.Kd * 2), rB = g8_rsrc(a.src, a.src_bytes);
    // A pieces of this wave: blocks w, w + NW, w + 2 NW (block q = tile rows 16 q ..: wave row q / 6, fragment q % 6)
    unsigned avo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int q = w + NW * j, qm = q / 6, qi = q - 6 * qm;
        const int row = (qi & 1) * H + cb + 48 * qm + 16 * (qi >> 1) + dr;
        avo[j] = (unsigned)(((size_t)row * a.Kd + dk) * 2);
    }
    // B pieces: blocks w + NW j below BB
    const int nbp = (BB - w + NW - 1) / NW;                                   // of this wave (wave-uniform)
    int btl[NBP]; int bvo[NBP];
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
        btl[j] = t0 + 16 * (w + NW * j) + dr;
        bvo[j] = (int)((((size_t)b * a.Tp + btl[j]) * H + dk) * 2);
    }
    const int kpb = a.KB / 32, nk = a.nblk * kpb;
    auto slot = [&](const int kt) -> unsigned { return (unsigned)(NST == 4 ? (kt & 3) : kt % 3) * STB; };
    // one piece of stage kt (pc = 0..2: A, 3..: B); the pieces of a stage are issued BETWEEN the MFMA rows of the stage in progress:
    // a piece holds its wave's issue for 60-180 cycles (MI355X_MICROARCH.md, cycle constants), which the other wave of the SIMD
    // fills with its MFMAs only if the pieces are not all issued at once behind the barrier, where both waves stand.
    // Stages past the end are issued all out-of-range (zeros into a free slot): the vmcnt arithmetic never changes.
    int iblk = 0, ikin = 0;                      // block (tap) and channel offset of the stage being issued
    auto piece = [&](const int kt, const int pc) __attribute__((always_inline)) {
        const unsigned base = slot(kt);
        const bool live = kt < nk;
        if (pc < 3) { g8_dma(rA, live ? avo[pc] : OOB, live ? (unsigned)kt * 64u : 0u, base + (unsigned)(w + NW * pc) * 1024u); return; }
        const int jb = pc - 3;
        const int shift = a.shift0 - iblk * a.shift_step;
        const int sk = (int)(((long)iblk * (long)a.blk_stride + (long)ikin - (long)shift * H) * 2);
        if (jb < nbp) g8_dma(rB, (live && btl[jb] >= shift) ? (unsigned)(bvo[jb] + sk) : OOB, 0u, base + (unsigned)(AB + w + NW * jb) * 1024u);
        if (jb == NBP - 1) { ikin += 32; if (ikin >= a.KB) { ikin = 0; ++iblk; } }
    };
    f32x4 acc[6][NCW];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < NCW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NST - 1; ++kt)
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) piece(kt, pc);
    const unsigned rd_a = (unsigned)(6 * wm) * 1024u + rdl, rd_b = (unsigned)(AB + NCW * wn) * 1024u + rdl;
    const int inflight = (3 + nbp) * (NST - 2);  // my pieces of the younger stages that may stay in flight
    for (int kt = 0; kt < nk; ++kt) {
        // my pieces of stage kt have landed (the younger stages stay in flight) ...
        if (inflight == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (inflight == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (inflight == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (inflight == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // ... and my fragment reads of stage kt - 1 are back
        __builtin_amdgcn_s_barrier();                            // everybody's: stage kt readable, the slot of stage kt - 1 free
        asm volatile("" ::: "memory");
        const unsigned char* sb = lds + slot(kt);
        bf16x8 af[6], bfr[NCW];
#pragma unroll
        for (int j = 0; j < NCW; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + rd_b + j * 1024);
#pragma unroll
        for (int i = 0; i < 6; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + rd_a + i * 1024);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = 0; j < NCW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            if (i < NP) {                                        // stage kt + NST - 1 goes into the slot the barrier has just freed
                __builtin_amdgcn_sched_barrier(0);
                piece(kt + NST - 1, i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the all-zero stages past the end may still be landing
#ifdef SWN_G8_NOEPI          // diagnostic build (tools/build_variant.sh, tools/ab_variants.sh): the main loop alone - one store keeps the accumulators live
    {
        f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < NCW; ++j) sum += acc[i][j];
        const int t = t0 + 16 * NCW * wn + n;
        if (t < a.Tp) *reinterpret_cast<uint2*>(a.hnext + ((size_t)b * a.Tp + t) * H + cb + 48 * wm + 4 * g4) = make_uint2(__float_as_uint(sum[0] + sum[1]), __float_as_uint(sum[2] + sum[3]));
        return;
    }
#endif
    // ---- gate epilogue: accumulator (i, j, r) = tile row 96 wm + 16 i + 4 g4 + r (i even: gate, odd: candidate of channels
    // cb + 48 wm + 16 (i >> 1) + 4 g4 + r), position t0 + 16 NCW wn + 16 j + n
    int fj[NCW], jj0[NCW];
    gate_frames<NCW>(t0 + 16 * NCW * wn + n + a.coff, a.U, fj, jj0);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int ch = cb + 48 * wm + 16 * p + 4 * g4;
        constexpr int JB = 2;                        // positions per batch of loads (three spill beside the 144 accumulator registers)
#pragma unroll
        for (int j0 = 0; j0 < NCW; j0 += JB)
            gate_group<JB>(a, b, ch, t0 + 16 * NCW * wn + 16 * j0 + n, reinterpret_cast<const int (&)[JB]>(fj[j0]), reinterpret_cast<const int (&)[JB]>(jj0[j0]),
                           reinterpret_cast<const f32x4 (&)[JB]>(acc[2 * p][j0]), reinterpret_cast<const f32x4 (&)[JB]>(acc[2 * p + 1][j0]));
    }
}

// fp32 [rows][ld] (first `cols` of each row) -> bf16 [rows][cols]
__global__ void bf16g_convert_kernel(const float* __restrict__ src, int ld, int rows, int cols, unsigned short* __restrict__ dst) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)rows * cols) return;
    const int r = (int)(e / cols), c = (int)(e - (size_t)r * cols);
    dst[e] = f2bf(src[(size_t)r * ld + c]);
}

// input layer: h0[b][t][o] = softsign(cb + sum_k (cv[k][o] * audio[ai - (K-1-k)] + cc[k][o]))   (Laplace; cswnv_shift1.py:203-206)
// any kernel size (the form used for K > 8)
__global__ __launch_bounds__(256) void bf16g_input_anyk_kernel(const float* __restrict__ P, size_t o_cb, size_t o_cv, size_t o_cc,
                                                          const float* __restrict__ audio, unsigned short* __restrict__ h0,
                                                          int H, int K, int seg, int Tp) {
    const int o = threadIdx.x, b = blockIdx.y;
    const float* au = audio + (size_t)b * (Tp + seg - 1);
    for (int i = 0; i < 16; ++i) {
        const int t = blockIdx.x * 16 + i;
        if (t >= Tp || o >= H) break;
        const int ai = t + seg - 1;
        float acc = P[o_cb + o];
        for (int k = 0; k < K; ++k) {
            const int r = ai - (K - 1 - k);
            if (r >= 0) acc += fmaf(P[o_cv + (size_t)k * H + o], au[r], P[o_cc + (size_t)k * H + o]);
        }
        h0[((size_t)b * Tp + t) * H + o] = f2bf(acc / (1.f + fabsf(acc)));
    }
}

constexpr int GIN_POS = 64;          // positions per workgroup of the input kernel (K <= 8)
__global__ __launch_bounds__(256) void bf16g_input_kernel(const float* __restrict__ P, size_t o_cb, size_t o_cv, size_t o_cc,
                                                          const float* __restrict__ audio, unsigned short* __restrict__ h0,
                                                          int H, int K, int seg, int Tp) {
    // thread = channel; its K fused taps (value, constant) stay in registers for the GIN_POS positions of the workgroup,
    // the waveform window is read once into LDS (it was 2K + 1 loads per output before: 86 us at 8 x 16 500)
    __shared__ float win[GIN_POS + 16];
    const int o = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * GIN_POS;
    const float* au = audio + (size_t)b * (Tp + seg - 1);
    const int a0 = t0 + seg - 1 - (K - 1);                       // waveform index of win[0]
    for (int e = threadIdx.x; e < GIN_POS + K - 1; e += 256) {
        const int r = a0 + e;
        win[e] = (r >= 0 && r < Tp + seg - 1) ? au[r] : 0.f;
    }
    float cvr[8], ccr[8];
    const bool live = o < H;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        cvr[k] = (live && k < K) ? P[o_cv + (size_t)k * H + o] : 0.f;
        ccr[k] = (live && k < K) ? P[o_cc + (size_t)k * H + o] : 0.f;
    }
    const float cb = live ? P[o_cb + o] : 0.f;
    __syncthreads();
    if (!live) return;
    for (int i = 0; i < GIN_POS; ++i) {
        const int t = t0 + i;
        if (t >= Tp) break;
        float acc = cb;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // tap k reads waveform index ai - (K-1-k); taps before the start of the sequence are skipped (the reference's
            // zero padding sits behind wav_conv, so neither the value nor the constant term exists there)
            if (k < K && a0 + i + k >= 0) acc += fmaf(cvr[k], win[i + k], ccr[k]);
        }
        h0[((size_t)b * Tp + t) * H + o] = f2bf(acc * __builtin_amdgcn_rcpf(1.f + fabsf(acc)));
    }
}

// softmax input layer: h0[b][t][o] = softsign(cb + sum_k ct[k][idx[t-(K-1-k)]][o]) over the taps inside the sequence (dswnv.py:257-260)
__global__ __launch_bounds__(256) void bf16g_input_softmax_kernel(const float* __restrict__ P, size_t o_cb, size_t o_ct,
                                                                  const int* __restrict__ audio, unsigned short* __restrict__ h0,
                                                                  int H, int K, int Q, int Tp) {
    const int o = threadIdx.x, b = blockIdx.y;
    const int* au = audio + (size_t)b * Tp;
    for (int i = 0; i < 16; ++i) {
        const int t = blockIdx.x * 16 + i;
        if (t >= Tp || o >= H) break;
        float acc = P[o_cb + o];
        for (int k = 0; k < K; ++k) {
            const int r = t - (K - 1 - k);
            if (r >= 0) { int idx = au[r] % Q; idx = idx < 0 ? idx + Q : idx; acc += P[o_ct + ((size_t)k * Q + idx) * H + o]; }
        }
        h0[((size_t)b * Tp + t) * H + o] = f2bf(acc / (1.f + fabsf(acc)));
    }
}

// dropout mode: a dropped layer's output as the NEXT layer reads it (conv operand and highway term): dst[b][t][c] =
// src[b][t][c] * mask[b][c][t] (bf16 time-major rows x the reference's (B, H, Tp) fp32 mask; 0 or 1/(1-p), a power of two: exact).
// 64 positions x 64 channels per workgroup, the mask tile transposed through LDS.
__global__ __launch_bounds__(256) void bf16g_mask_kernel(const unsigned short* __restrict__ src, const float* __restrict__ mask,
                                                         unsigned short* __restrict__ dst, int H, int Tp) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = (tid >> 6) + 4 * i, t = tid & 63;
        tile[c][t] = (c0 + c < H && t0 + t < Tp) ? mask[((size_t)b * H + c0 + c) * Tp + t0 + t] : 0.f;
    }
    __syncthreads();
    const int t = tid >> 2, cq = (tid & 3) * 16;
    if (t0 + t >= Tp) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c = c0 + cq + 8 * h;
        if (c >= H) break;                                           // H % 8 == 0
        const size_t o = ((size_t)b * Tp + t0 + t) * H + c;
        const uint4 v = *reinterpret_cast<const uint4*>(src + o);
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
        unsigned r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = bf2f((unsigned short)(u[j] & 0xffff)) * tile[cq + 8 * h + 2 * j][t];
            const float hi = bf2f((unsigned short)(u[j] >> 16)) * tile[cq + 8 * h + 2 * j + 1][t];
            r[j] = f2bf(lo) | ((unsigned)f2bf(hi) << 16);
        }
        *reinterpret_cast<uint4*>(dst + o) = make_uint4(r[0], r[1], r[2], r[3]);
    }
}

struct GOff { size_t wd, wsk, w1, w2, total; };
GOff g_offsets(const SwnGeom& g) {
    GOff o;
    o.wd = 0;
    o.wsk = o.wd + (size_t)g.L * 2 * g.H * g.K * g.H;
    o.w1 = o.wsk + (size_t)g.S * g.L * g.H;
    o.w2 = o.w1 + (size_t)g.O1 * g.S;
    o.total = o.w2 + (size_t)g.NO * g.O1;
    return o;
}

template <int EPI>
void launch_gemm(GemmArgs a, int ntt, int nmt, int batch, hipStream_t st) {
    constexpr int WNT = 4;
    a.ntt = ntt; a.nmt = nmt;
    const int chunk = (ntt * batch + 7) / 8;
    hipLaunchKernelGGL((bf16g_gemm_kernel<EPI, WNT>), dim3((unsigned)(8 * chunk * nmt)), dim3(256), 0, st, a);
}

// the LDS-DMA gated layer (M = 384): tile width by fewer wasted rounds over the CUs; false = not launched (attribute refused)
template <int NCW, int NW>
bool launch_gate8_as(GemmArgs a, int batch, hipStream_t st) {
    constexpr int BN = 32 * NCW, NST = NW == 8 ? 4 : 3;
    constexpr size_t ldsb = (size_t)NST * (3 * NW + BN / 16) * 1024;
    static bool attr_ok = false;
    if (!attr_ok) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(bf16g_gate8_kernel<NCW, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) {
            (void)hipGetLastError(); return false;
        }
        attr_ok = true;
    }
    a.ntt = (a.Tp + BN - 1) / BN; a.nmt = 1;
    const int ntiles = a.ntt * batch, chunk = (ntiles * (8 / NW) + 7) / 8;
    hipLaunchKernelGGL((bf16g_gate8_kernel<NCW, NW>), dim3((unsigned)(8 * chunk)), dim3(64 * NW), ldsb, st, a, ntiles);
    return true;
}
bool launch_gate8(GemmArgs a, int batch, hipStream_t st) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    if (a.cond && (size_t)batch * a.Tf * a.N * 4 >= (1ull << 31)) return false;
    auto rounds_cost = [&](int bn) { const long tiles = (long)batch * ((a.Tp + bn - 1) / bn); return ((tiles + ncu - 1) / ncu) * bn; };
    // two half-row workgroups per CU (NW = 4); the 512-thread form (NW = 8) measured 3 % slower (REF6 forward 1.244 against 1.208 ms)
    return rounds_cost(192) <= rounds_cost(128) ? launch_gate8_as<6, 4>(a, batch, st) : launch_gate8_as<4, 4>(a, batch, st);
}

}  // namespace

// Plain GEMM over bf16 time-major rows on the tiled kernel of this file (the dropout-mode in_x products and their data
// gradient, csrc/swn_stack_bf16.hip / swn_bwd_bl6.hip): for every position p = (b, t < Tp)
//     out[p][m] = bias[m] + sum_{blk < nblk} sum_{i < KB} A[m][blk*KB + i] * src[blk*blk_stride + p*KB + i]
// A: bf16 [M][nblk*KB] row-major (KB % 32 == 0; M % 4 == 0 for the bf16 output); bias may be null.  out_bf != null: bf16 rows of out_ld elements;
// else out_f: fp32 channel-major (B, NO, Tp), rows m < NO.
int swn_bf16g_plain(const unsigned short* A, int M, const unsigned short* src, size_t blk_stride, size_t src_bytes, int KB, int nblk,
                    int Tp, int B, const float* bias, unsigned short* out_bf, int out_ld, float* out_f, int NO, hipStream_t st) {
    if (KB % 32 != 0 || (out_bf && M % 4 != 0) || src_bytes >= (1ull << 31) || (size_t)M * nblk * KB * 2 >= (1ull << 31)) return SWN_E_UNSUPPORTED;
    GemmArgs a = {};
    a.A = A; a.M = M; a.Kd = nblk * KB; a.src = src; a.blk_stride = blk_stride; a.src_bytes = src_bytes; a.KB = KB; a.nblk = nblk;
    a.shift0 = 0; a.shift_step = 0; a.Tp = Tp; a.B = B; a.bias = bias; a.out_bf = out_bf; a.out_ld = out_ld; a.out_f = out_f; a.NO = NO;
    const int tx = (Tp + 127) / 128, nmt = (M + TM - 1) / TM;
    if (out_bf) launch_gemm<EPI_BF16>(a, tx, nmt, B, st);
    else launch_gemm<EPI_F32>(a, tx, nmt, B, st);
    return SWN_OK;
}

// geometry class of this file: Laplace or softmax, H a multiple of 64 (row tiles of 64 channels, k-tiles of 32), S and O1
// multiples of 32, at most 256 input-layer channels per block
int swn_bf16g_geom(const swn_net_desc* d, SwnGeom* g) {
    int rc = swn_make_geom(d, g);
    if (rc < 0) return rc;
    if (g->H % 64 != 0 || g->H > 256 || g->S % 32 != 0 || g->O1 % 32 != 0) return SWN_E_UNSUPPORTED;
    return SWN_OK;
}

size_t swn_bf16g_weight_bytes(const SwnGeom& g) { return g_offsets(g).total * sizeof(unsigned short); }

int swn_bf16g_pack(const SwnGeom& g, const float* packed, void* wbf_, hipStream_t st) {
    SwnLayout y; swn_make_layout(&g, &y);
    const GOff o = g_offsets(g);
    unsigned short* wbf = reinterpret_cast<unsigned short*>(wbf_);
    auto conv = [&](size_t src_off, int ld, int rows, int cols, size_t dst_off) {
        const size_t n = (size_t)rows * cols;
        hipLaunchKernelGGL(bf16g_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, packed + src_off, ld, rows, cols, wbf + dst_off);
    };
    conv(y.wd, g.K * g.Hp, g.L * 2 * g.H, g.K * g.H, o.wd);      // Hp == H in this class
    conv(y.wsk, g.L * g.Hp, g.S, g.L * g.H, o.wsk);
    conv(y.w1, g.Sp, g.O1, g.S, o.w1);
    conv(y.w2, g.O1p, g.NO, g.O1, o.w2);
    return swn_launch_status("swn_pack_bf16");
}

// ---- bf16 time-major [.][t][C] -> fp32 channel-major [.][C][Tp] (what the fp32 backward reads), 64 x 64 tiles through
// LDS: 16-byte loads along c, 256-byte rows stored along t.  z = matrix index: src z*src_stride, dst (z % nb)*dst_bs + (z / nb)*dst_ls
__global__ __launch_bounds__(256) void bf16g_expand_kernel(const unsigned short* __restrict__ src, size_t src_stride, int C, int Tp,
                                                           float* __restrict__ dst, int nb, size_t dst_bs, size_t dst_ls) {
    // 128 positions x 64 channels per workgroup: a destination row gets 512 contiguous bytes per workgroup (64-position tiles
    // wrote 256-byte pieces 66 KB apart: 3.8 TB/s for the hidden states of cfg4 at REF6)
    __shared__ float tile[64][129];
    const int z = blockIdx.z, t0 = blockIdx.x * 128, c0 = blockIdx.y * 64, tid = threadIdx.x;
    const unsigned short* S = src + (size_t)z * src_stride;
    float* D = dst + (size_t)(z % nb) * dst_bs + (size_t)(z / nb) * dst_ls;
    uint4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, t = e >> 3, cq = (e & 7) * 8;
        v[i] = make_uint4(0u, 0u, 0u, 0u);
        if (t0 + t < Tp && c0 + cq < C) v[i] = *reinterpret_cast<const uint4*>(S + (size_t)(t0 + t) * C + c0 + cq);     // C % 8 == 0
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, t = e >> 3, cq = (e & 7) * 8;
        const unsigned u[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tile[cq + 2 * j][t] = __uint_as_float(u[j] << 16);
            tile[cq + 2 * j + 1][t] = __uint_as_float(u[j] & 0xffff0000u);
        }
    }
    __syncthreads();
    const int lane = tid & 63;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = (tid >> 6) + 4 * i;
        if (c0 + c >= C) continue;
        float* row = D + (size_t)(c0 + c) * Tp + t0;
        if (t0 + lane < Tp) row[lane] = tile[c][lane];
        if (t0 + 64 + lane < Tp) row[64 + lane] = tile[c][64 + lane];
    }
}

// what swn_bf16g_forward left in `work` -> the fp32 work layout of swn_forward: hs | relu(skip) | relu(out_1)
int swn_bf16g_expand(const SwnGeom& g, const void* work, int batch, long Tp, float* fwd_work, bool hs_only, hipStream_t st) {
    auto r64 = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const unsigned short* hs = reinterpret_cast<const unsigned short*>(work);
    const size_t lstride = (size_t)batch * Tp * g.H;
    const unsigned short* skipb = hs + (size_t)(g.L + 1) * lstride;
    const unsigned short* o1b = skipb + (size_t)batch * Tp * g.S;
    float* s1 = fwd_work + r64((size_t)batch * (g.L + 1) * g.H * Tp);
    float* r1 = s1 + r64((size_t)batch * g.S * Tp);
    const unsigned tx = (unsigned)((Tp + 127) / 128);
    (void)hipGetLastError();
    // hidden states: matrix z = l * B + b  ->  dst [b][l][H][Tp]
    hipLaunchKernelGGL(bf16g_expand_kernel, dim3(tx, (g.H + 63) / 64, (g.L + 1) * batch), dim3(256), 0, st,
                       hs, (size_t)Tp * g.H, g.H, (int)Tp, fwd_work, batch, (size_t)(g.L + 1) * g.H * Tp, (size_t)g.H * Tp);
    if (hs_only) return swn_launch_status("swn_bf16_work_to_f32");
    hipLaunchKernelGGL(bf16g_expand_kernel, dim3(tx, (g.S + 63) / 64, batch), dim3(256), 0, st,
                       skipb, (size_t)Tp * g.S, g.S, (int)Tp, s1, batch, (size_t)g.S * Tp, (size_t)0);
    hipLaunchKernelGGL(bf16g_expand_kernel, dim3(tx, (g.O1 + 63) / 64, batch), dim3(256), 0, st,
                       o1b, (size_t)Tp * g.O1, g.O1, (int)Tp, r1, batch, (size_t)g.O1 * Tp, (size_t)0);
    return swn_launch_status("swn_bf16_work_to_f32");
}

size_t swn_bf16g_work_bytes(const SwnGeom& g, int batch, long Tp) {
    return ((size_t)(g.L + 1) * g.H + g.S + g.O1) * batch * Tp * sizeof(unsigned short);
}

size_t swn_bf16g_keep_floats(const SwnGeom& g, int batch, long Tp) {
    return (size_t)g.L * (((size_t)batch * 2 * g.H * Tp + 63) & ~(size_t)63);
}

// gx != null: the dropout mode (cswnv_shift1.py:194-198,211-217,269-278) - sample-rate in_x products instead of the hoisted
// cond, and drop_h[l] (device pointers, null = no mask) = the (B, H, Tp) mask on layer l's OUTPUT as the next layer reads it
// (the skip GEMM reads the undropped states); hm16 = room for one masked level, B * Tp * H bf16
int swn_bf16g_forward(const SwnGeom& g, const float* packed, const void* wbf_, const float* cond, const void* audio,
                      int batch, int n_frames, void* work, float* out, hipStream_t st, float* a_keep,
                      const float* gx, const float* const* drop_h, unsigned short* hm16) {
    SwnLayout y; swn_make_layout(&g, &y);
    const GOff o = g_offsets(g);
    const long Tp = (long)n_frames * g.U - 2 * g.seg + 1;
    const unsigned short* wbf = reinterpret_cast<const unsigned short*>(wbf_);
    unsigned short* hs = reinterpret_cast<unsigned short*>(work);
    const size_t lstride = (size_t)batch * Tp * g.H;
    unsigned short* skipb = hs + (size_t)(g.L + 1) * lstride;
    unsigned short* o1b = skipb + (size_t)batch * Tp * g.S;
    // 32-bit buffer offsets: every GEMM operand must stay below 2 GiB
    if ((size_t)g.L * lstride * 2 >= (1ull << 31) || (size_t)batch * Tp * (g.S > g.O1 ? g.S : g.O1) * 2 >= (1ull << 31)) return SWN_E_UNSUPPORTED;
    (void)hipGetLastError();
    if (g.kind == SWN_KIND_LAPLACE)
        if (g.K <= 8) hipLaunchKernelGGL(bf16g_input_kernel, dim3((unsigned)((Tp + GIN_POS - 1) / GIN_POS), batch), dim3(256), 0, st, packed, y.cb, y.cv, y.cc,
                           reinterpret_cast<const float*>(audio), hs, g.H, g.K, g.seg, (int)Tp);
        else hipLaunchKernelGGL(bf16g_input_anyk_kernel, dim3((unsigned)((Tp + 15) / 16), batch), dim3(256), 0, st, packed, y.cb, y.cv, y.cc,
                           reinterpret_cast<const float*>(audio), hs, g.H, g.K, g.seg, (int)Tp);
    else
        hipLaunchKernelGGL(bf16g_input_softmax_kernel, dim3((unsigned)((Tp + 15) / 16), batch), dim3(256), 0, st, packed, y.cb, y.ct,
                           reinterpret_cast<const int*>(audio), hs, g.H, g.K, g.Q, (int)Tp);
    GemmArgs a = {};
    a.aidx = (g.kind == SWN_KIND_SOFTMAX && g.audio_in) ? reinterpret_cast<const int*>(audio) : nullptr; a.Q = g.Q; a.o_wxa = y.wxa;
    a.Tp = (int)Tp; a.B = batch; a.P = packed; a.cond = cond; a.H = g.H; a.seg = g.seg; a.U = g.U; a.Tf = n_frames; a.N = g.N;
    a.coff = g.seg; a.o_bd = y.bd; a.o_bx = y.bx; a.o_wup = y.wup;
    a.gx = gx; a.o_bxr = y.bxr; a.gx_rows = g.L * 2 * g.H;
    constexpr int WNT = 4;                      // 128 x 128 workgroup tile (a 128 x 256 tile with 64 x 128 per wave needs 392 registers: one wave per SIMD, 1.4x slower)
    const unsigned tx = (unsigned)((Tp + 32 * WNT - 1) / (32 * WNT));
    for (int l = 0; l < g.L; ++l) {
        a.A = wbf + o.wd + (size_t)l * 2 * g.H * g.K * g.H; a.M = 2 * g.H; a.Kd = g.K * g.H;
        a.src = hs + (size_t)l * lstride; a.blk_stride = 0; a.KB = g.H; a.nblk = g.K; a.src_bytes = lstride * 2;
        a.shift0 = (g.K - 1) * g.dil[l]; a.shift_step = g.dil[l];
        a.hprev = hs + (size_t)l * lstride; a.hnext = hs + (size_t)(l + 1) * lstride; a.l = l;
        if (gx && drop_h && l > 0 && drop_h[l - 1]) {           // layer l-1's output was dropped: this layer reads the masked copy
            hipLaunchKernelGGL(bf16g_mask_kernel, dim3((unsigned)((Tp + 63) / 64), (g.H + 63) / 64, batch), dim3(256), 0, st,
                               hs + (size_t)l * lstride, drop_h[l - 1], hm16, g.H, (int)Tp);
            a.src = hm16; a.hprev = hm16;
        }
        a.a_out = a_keep ? a_keep + (size_t)l * (((size_t)batch * 2 * g.H * Tp + 63) & ~(size_t)63) : nullptr;
        if (g.H == 192 && a.KB == 192 && launch_gate8(a, batch, st)) continue;      // LDS-DMA layer kernel of the run.sh Laplace geometry
        launch_gemm<EPI_GATE>(a, (int)tx, g.H / 64, batch, st);
    }
    a.a_out = nullptr;
    // skip = relu(Wsk . [h_1 .. h_L] + b)
    a.A = wbf + o.wsk; a.M = g.S; a.Kd = g.L * g.H; a.src = hs + lstride; a.blk_stride = lstride; a.KB = g.H; a.nblk = g.L;
    a.src_bytes = (size_t)g.L * lstride * 2;
    a.shift0 = 0; a.shift_step = 0; a.bias = packed + y.bsk; a.out_bf = skipb; a.out_ld = g.S;
    launch_gemm<EPI_RELU_BF16>(a, (int)tx, (g.S + TM - 1) / TM, batch, st);
    a.A = wbf + o.w1; a.M = g.O1; a.Kd = g.S; a.src = skipb; a.blk_stride = 0; a.KB = g.S; a.nblk = 1; a.src_bytes = (size_t)batch * Tp * g.S * 2;
    a.bias = packed + y.b1; a.out_bf = o1b; a.out_ld = g.O1;
    launch_gemm<EPI_RELU_BF16>(a, (int)tx, (g.O1 + TM - 1) / TM, batch, st);
    a.A = wbf + o.w2; a.M = g.NO; a.Kd = g.O1; a.src = o1b; a.KB = g.O1; a.nblk = 1; a.src_bytes = (size_t)batch * Tp * g.O1 * 2;
    a.bias = packed + y.b2; a.out_f = out; a.NO = g.NO;
    launch_gemm<EPI_F32>(a, (int)tx, (g.NO + TM - 1) / TM, batch, st);
    return swn_launch_status("swn_forward_bf16");
}
