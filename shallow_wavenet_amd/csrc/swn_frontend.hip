// Frame-rate front end on gfx950:
//   scale_in (1x1)  ->  two-sided dilated k=3 conv stack  ->  hoisted in_x GEMM
// Replaces  x = conv_aux(scale_in(aux))  and the per-sample-position  in_x[l](x)  1x1 convs of
// cswnv_shift1.py:193,276,297 / dswnv.py:252,285,302.  The rank-1 ConvTranspose2d upsampler
// (cswnv_shift1.py:37-65) is never materialised: consumers rebuild
//   in_x(x)[o,t] = bx[l][o] + sum_s w_up[(t+s)%U] * cond[b][(t+s)/U][l][s][o].
// All fp32 (decode parity needs fp32).  Frame-rate work is <1% of a decode.
#include <hip/hip_runtime.h>
#include "swn_geom.hpp"
#include "swn_mma.hpp"
#include <type_traits>

namespace {

// out[b][co][f] = bias[co] + sum_ci sum_k W[co][ci][k] * in[b][ci][f + (k-(KS-1)/2)*dil]
// block: 64 frames x 4 channel groups; thread computes CPT output channels of one frame.
template <int CPT>
__global__ __launch_bounds__(256) void conv1d_same_kernel(
    const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int cin, int cout, int ks, int dil, int n_frames) {
    const int f = blockIdx.x * 64 + (threadIdx.x & 63);
    const int cg = threadIdx.x >> 6;                       // wave index -> channel group
    const int co0 = (blockIdx.y * 4 + cg) * CPT;
    const int b = blockIdx.z;
    if (co0 >= cout) return;
    float acc[CPT];
#pragma unroll
    for (int r = 0; r < CPT; ++r) acc[r] = (co0 + r < cout) ? bias[co0 + r] : 0.f;
    const int half = (ks - 1) / 2;
    // inputs through a buffer resource (out-of-range offset = zero padding), eight input channels x KS taps in flight;
    // the multiply-adds keep the (ci ascending, k ascending) order of the plain loop
    const __amdgpu_buffer_rsrc_t rI = rsrc_of(in + (size_t)b * cin * n_frames);
    constexpr int CB = 8;
    auto run = [&](auto ks_tag) {
        constexpr int KS = decltype(ks_tag)::value;
        unsigned toff[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int ff = f + (k - half) * dil;
            toff[k] = (f < n_frames && ff >= 0 && ff < n_frames) ? (unsigned)(ff * 4) : SWN_OOB;
        }
        for (int ci0 = 0; ci0 < cin; ci0 += CB) {
            float x[CB][KS];
#pragma unroll
            for (int u = 0; u < CB; ++u)
#pragma unroll
                for (int k = 0; k < KS; ++k)
                    x[u][k] = bld1(rI, ci0 + u < cin ? toff[k] + (unsigned)((ci0 + u) * n_frames * 4) : SWN_OOB);
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int ci = ci0 + u < cin ? ci0 + u : cin - 1;          // past the end: x is zero, any valid weight will do
#pragma unroll
                for (int k = 0; k < KS; ++k)
#pragma unroll
                    for (int r = 0; r < CPT; ++r) {
                        const int co = co0 + r;                            // wave-uniform -> scalar weight load
                        const float wv = (co < cout) ? w[((size_t)co * cin + ci) * KS + k] : 0.f;
                        acc[r] = fmaf(wv, x[u][k], acc[r]);
                    }
            }
        }
    };
    if (ks == 1) run(std::integral_constant<int, 1>{});
    else if (ks == 3) run(std::integral_constant<int, 3>{});
    else {
        const float* inb = in + (size_t)b * cin * n_frames;
        for (int ci = 0; ci < cin; ++ci)
            for (int k = 0; k < ks; ++k) {
                const int ff = f + (k - half) * dil;
                const float x = (f < n_frames && ff >= 0 && ff < n_frames) ? inb[(size_t)ci * n_frames + ff] : 0.f;
#pragma unroll
                for (int r = 0; r < CPT; ++r) {
                    const int co = co0 + r;
                    const float wv = (co < cout) ? w[((size_t)co * cin + ci) * ks + k] : 0.f;
                    acc[r] = fmaf(wv, x, acc[r]);
                }
            }
    }
    if (f < n_frames) {
#pragma unroll
        for (int r = 0; r < CPT; ++r)
            if (co0 + r < cout) out[((size_t)b * cout + co0 + r) * n_frames + f] = acc[r];
    }
}

// The same convolution for ks == 3 with the workgroup's operands staged in LDS: the 64 frames (+ halo) of ALL input
// channels and the 16 output channels' weight rows are fetched once, coalesced, and the multiply-adds then run without a
// memory wait in the loop.  (The kernel above pays one L2 round trip per eight input channels for x and streams ~1 MB of
// weights through the scalar cache, which thrashes: 120 us for 1 200 frames at cin = 162; staging x alone changed
// nothing.)  Weights lie as wl[wave][ci*3 + k][4 outputs]: one broadcast 16-byte read serves the wave's four outputs.
// Same (ci ascending, k ascending) chain per output, so the results are bit-identical to conv1d_same_kernel.
__global__ __launch_bounds__(256) void conv1d_same_lds_kernel(
    const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int cin, int cout, int dil, int n_frames) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int PW = 64 + 2 * dil, J = cin * 3;
    float* wl = lds;                                       // [4][J][4]
    float* xs = lds + 16 * J;                              // [cin][PW]
    const int lane = threadIdx.x & 63;
    const int cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int f0 = blockIdx.x * 64, b = blockIdx.z;
    {
        // eight rows (sixteen loads) in flight per wave before anything is written: a load-store-load chain per element
        // would pay one L2 round trip per row
        const __amdgpu_buffer_rsrc_t rI = rsrc_of(in + (size_t)b * cin * n_frames);
        const int fa = f0 - dil + lane, fb = fa + 64;
        const unsigned oa = (fa >= 0 && fa < n_frames) ? (unsigned)(fa * 4) : SWN_OOB;
        const unsigned ob = (lane + 64 < PW && fb >= 0 && fb < n_frames) ? (unsigned)(fb * 4) : SWN_OOB;
        for (int c0 = cg; c0 < cin; c0 += 32) {
            float va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ci = c0 + 4 * u;
                const unsigned ro = ci < cin ? (unsigned)(ci * n_frames * 4) : SWN_OOB;
                va[u] = bld1(rI, ((oa | ro) & SWN_OOB) ? SWN_OOB : oa + ro);
                vb[u] = bld1(rI, ((ob | ro) & SWN_OOB) ? SWN_OOB : ob + ro);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ci = c0 + 4 * u;
                if (ci < cin) { xs[ci * PW + lane] = va[u]; if (lane + 64 < PW) xs[ci * PW + lane + 64] = vb[u]; }
            }
        }
        const __amdgpu_buffer_rsrc_t rW = rsrc_of(w);
        for (int q = 0; q < 4; ++q) {                      // this wave stages output `cg` of every wave's group of four
            const int co = blockIdx.y * 16 + 4 * q + cg;
            const unsigned ro = co < cout ? (unsigned)((size_t)co * J * 4) : SWN_OOB;
            for (int j0 = 0; j0 < J; j0 += 512) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = j0 + lane + 64 * u; v[u] = bld1(rW, (j < J && ro != SWN_OOB) ? ro + (unsigned)(j * 4) : SWN_OOB); }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = j0 + lane + 64 * u; if (j < J) wl[(q * J + j) * 4 + cg] = v[u]; }
            }
        }
    }
    __syncthreads();
    const int co0 = blockIdx.y * 16 + 4 * cg;
    if (co0 >= cout) return;
    float acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = (co0 + r < cout) ? bias[co0 + r] : 0.f;
    const float* xp = xs + lane;
    const float4* wq = reinterpret_cast<const float4*>(wl) + (size_t)cg * J;
#pragma unroll 4
    for (int ci = 0; ci < cin; ++ci) {
        float x[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = xp[ci * PW + k * dil];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float4 wv4 = wq[ci * 3 + k];
            const float wv[4] = {wv4.x, wv4.y, wv4.z, wv4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = fmaf(wv[r], x[k], acc[r]);
        }
    }
    const int f = f0 + lane;
    if (f < n_frames) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (co0 + r < cout) out[((size_t)b * cout + co0 + r) * n_frames + f] = acc[r];
    }
}

// cond[b][f][n] = sum_c Wx[n][c] * C[b][c][f]     (M = frames of one utterance, N, Kd = A0)
// 64x64 tile, BK = 16, 256 threads x (4x4) outputs, fp32 FMA chains in ascending c.
__global__ __launch_bounds__(256) void cond_gemm_kernel(
    const float* __restrict__ C, const float* __restrict__ Wx, float* __restrict__ cond,
    int n_frames, int N, int A0, int A0p) {
    __shared__ float As[16][64 + 4];
    __shared__ float Bs[16][64 + 4];
    const int b = blockIdx.z;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int tid = threadIdx.x;
    const int tm = (tid & 15) * 4, tn = (tid >> 4) * 4;
    const float* Cb = C + (size_t)b * A0 * n_frames;
    float acc[4][4] = {};
    // the next k-tile's operands are fetched into registers before the current tile's multiply-adds (the loop used to pay
    // one L2 round trip per 16 k); same ascending-k chains
    float ra[4]; float4 rb;
    const int nn = tid >> 2, kq = (tid & 3) * 4;
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, kk = e >> 6, mm = e & 63;
            const int k = k0 + kk, m = m0 + mm;
            ra[i] = (k < A0 && m < n_frames) ? Cb[(size_t)k * n_frames + m] : 0.f;
        }
        const int n = n0 + nn, k = k0 + kq;
        rb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N && k < A0p) rb = *reinterpret_cast<const float4*>(Wx + (size_t)n * A0p + k);
    };
    fetch(0);
    for (int k0 = 0; k0 < A0; k0 += 16) {
        // A tile: 16 k x 64 m, m contiguous in memory ; B tile: 64 n x 16 k, k contiguous (rows padded to A0p, zero filled)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int e = tid + 256 * i; As[e >> 6][e & 63] = ra[i]; }
        Bs[kq + 0][nn] = rb.x; Bs[kq + 1][nn] = rb.y; Bs[kq + 2][nn] = rb.z; Bs[kq + 3][nn] = rb.w;
        __syncthreads();
        if (k0 + 16 < A0) fetch(k0 + 16);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float4 a = *reinterpret_cast<const float4*>(&As[kk][tm]);
            const float4 bb = *reinterpret_cast<const float4*>(&Bs[kk][tn]);
            const float av[4] = {a.x, a.y, a.z, a.w};
            const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + tm + i;
        if (m >= n_frames) continue;
        float* row = cond + ((size_t)b * n_frames + m) * N;
        if (n0 + tn + 3 < N && (N & 3) == 0) {
            *reinterpret_cast<float4*>(row + n0 + tn) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n0 + tn + j < N) row[n0 + tn + j] = acc[i][j];
        }
    }
}

}  // namespace

extern "C" size_t swn_frontend_work_floats(const swn_net_desc* d, int batch, int n_frames) {
    SwnGeom g; if (swn_make_geom(d, &g) < 0 || batch < 1 || n_frames < 1) return 0;
    size_t tot = (size_t)g.n_aux;
    for (int i = 0; i < g.auxl; ++i) tot += g.aux_cout[i];
    return tot * (size_t)batch * n_frames;
}

extern "C" size_t swn_cond_floats(const swn_net_desc* d, int batch, int n_frames) {
    SwnGeom g; if (swn_make_geom(d, &g) < 0 || batch < 1 || n_frames < 1) return 0;
    return (size_t)batch * n_frames * g.N;
}

extern "C" int swn_frontend(const swn_net_desc* d, const float* packed, const float* aux,
                            int batch, int n_frames, float* work, float* cond, void* stream_) {
    SwnGeom g; int rc = swn_make_geom(d, &g);
    if (rc < 0) return rc;
    if (!packed || !aux || !work || batch < 1 || n_frames < 1 || batch > 65535) return SWN_E_BADARG;
    SwnLayout y; swn_make_layout(&g, &y);
    hipStream_t st = (hipStream_t)stream_;
    (void)hipGetLastError();   // drop stale errors of earlier runtime calls; only our launches are reported
    const size_t bt = (size_t)batch * n_frames;
    // scale_in
    float* cur = work;
    {
        dim3 grid((n_frames + 63) / 64, (g.n_aux + 15) / 16, batch);
        hipLaunchKernelGGL(conv1d_same_kernel<4>, grid, dim3(256), 0, st, aux, packed + y.scale_w,
                           packed + y.scale_b, cur, g.n_aux, g.n_aux, 1, 1, n_frames);
    }
    const float* src = cur;
    cur += bt * g.n_aux;
    for (int i = 0; i < g.auxl; ++i) {
        dim3 grid((n_frames + 63) / 64, (g.aux_cout[i] + 15) / 16, batch);
        const size_t lds = ((size_t)g.aux_cin[i] * (64 + 2 * g.aux_dil[i]) + 16 * (size_t)g.aux_cin[i] * 3) * sizeof(float);
        bool staged = g.auxk == 3 && lds <= 150 * 1024;
        if (staged && lds > 48 * 1024) {
            static size_t granted = 0;                     // the attribute is per function: raise it when a larger geometry comes
            if (lds > granted) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv1d_same_lds_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) granted = lds;
                else staged = false;
            }
        }
        if (staged)
            hipLaunchKernelGGL(conv1d_same_lds_kernel, grid, dim3(256), lds, st, src, packed + y.aux_w[i],
                               packed + y.aux_b[i], cur, g.aux_cin[i], g.aux_cout[i], g.aux_dil[i], n_frames);
        else
            hipLaunchKernelGGL(conv1d_same_kernel<4>, grid, dim3(256), 0, st, src, packed + y.aux_w[i],
                               packed + y.aux_b[i], cur, g.aux_cin[i], g.aux_cout[i], g.auxk, g.aux_dil[i], n_frames);
        src = cur;
        cur += bt * g.aux_cout[i];
    }
    if (cond) {                            // null: the caller only wants the conv_aux activations (dropout mode)
        dim3 grid((n_frames + 63) / 64, (g.N + 63) / 64, batch);
        hipLaunchKernelGGL(cond_gemm_kernel, grid, dim3(256), 0, st, src, packed + y.wx, cond,
                           n_frames, g.N, g.A0, g.A0p);
    }
    return swn_launch_status("swn_frontend");
}
