// Device-side parameter packing: the live nn.Parameter tensors (already in HBM) -> the flat kernel layout of
// swn_geom.hpp, without a host round trip.  Same formulas, same summation order and the same double accumulators as
// swn_pack_params (swn_pack.cpp), so the two buffers are bit-identical (tests/test_gpu_device_pack.py).
//
// Why it exists: a training step ends with optimizer.step(), which changes every parameter
// (train_cswnv_laplace-stftcmplx_shift1.py:872-874); the host packer costs one device->host copy per tensor plus an
// upload of the whole buffer (30-50 ms per step, ten times the BL6 forward+backward).  Here it is one launch.
#include <hip/hip_runtime.h>
#include "swn_geom.hpp"

#define SWN_PACK_MAXT 128        // 2 + 2*4 + 2 + 2 + 2 + 2 + 6*16 + 4 = 118 state_dict tensors at most

struct PackArgs {
    SwnGeom g;
    SwnLayout y;
    const float* t[SWN_PACK_MAXT];
    int i_scale, i_aux, i_up, i_c2d, i_wav, i_causal, i_inx, i_dil, i_skip, i_out1, i_out2;
};

enum { SEC_COPY = 0, SEC_CAUSAL, SEC_INX, SEC_INX_BIAS, SEC_DIL, SEC_SKIP, SEC_OUT, SEC_COUNT };

__device__ static inline void copy_n(float* dst, const float* src, size_t n, unsigned tid, unsigned nth) {
    for (unsigned i = tid; i < n; i += nth) dst[i] = src[i];
}

// effective in_x weight W_eff[o][c*seg+s] of layer l (the (seg,1) Conv2d folded in, cswnv_shift1.py:196-198)
__device__ static inline float inx_eff(const PackArgs& a, int l, int o, int c, int s) {
    const SwnGeom& g = a.g;
    const float* w = a.t[a.i_inx + 2 * l];
    if (!g.conv2d) {
        const int A = g.A0 * g.seg + (g.audio_in ? g.Q : 0);
        return w[(size_t)o * A + c * g.seg + s];
    }
    const float* c2w = a.t[a.i_c2d];
    double e = 0.0;
    for (int p = 0; p < g.A0; ++p)
        e += (double)w[(size_t)o * g.A0 + p] * c2w[((size_t)p * g.A0 + c) * g.seg + s];
    return (float)e;
}

__global__ void __launch_bounds__(256) pack_params_kernel(PackArgs a, float* __restrict__ out) {
    const SwnGeom& g = a.g;
    const SwnLayout& y = a.y;
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;      // 32-bit element indices: every section is far below 2^31
    // (64-bit divisions per element made these two kernels 160-250 us at the run.sh geometry)
    const unsigned nth = gridDim.x * blockDim.x;
    const int H = g.H, S = g.S, K = g.K, L = g.L, seg = g.seg, Q = g.Q, H2 = 2 * g.H;
    switch (blockIdx.y) {
    case SEC_COPY: {
        copy_n(out + y.scale_w, a.t[a.i_scale], (size_t)g.n_aux * g.n_aux, tid, nth);
        copy_n(out + y.scale_b, a.t[a.i_scale + 1], g.n_aux, tid, nth);
        for (int i = 0; i < g.auxl; ++i) {
            copy_n(out + y.aux_w[i], a.t[a.i_aux + 2 * i], (size_t)g.aux_cout[i] * g.aux_cin[i] * g.auxk, tid, nth);
            copy_n(out + y.aux_b[i], a.t[a.i_aux + 2 * i + 1], g.aux_cout[i], tid, nth);
        }
        copy_n(out + y.wup, a.t[a.i_up], g.U, tid, nth);
        copy_n(out + y.bup, a.t[a.i_up + 1], 1, tid, nth);
        copy_n(out + y.cb, a.t[a.i_causal + 1], H, tid, nth);
        for (int l = 0; l < L; ++l) copy_n(out + y.bd + (size_t)l * H2, a.t[a.i_dil + 2 * l + 1], H2, tid, nth);
        copy_n(out + y.b1, a.t[a.i_out1 + 1], g.O1, tid, nth);
        copy_n(out + y.b2, a.t[a.i_out2 + 1], g.NO, tid, nth);
        break;
    }
    case SEC_CAUSAL: {
        const float* cw = a.t[a.i_causal];
        const float* wav_w = g.wav ? a.t[a.i_wav] : nullptr;
        const float* wav_b = g.wav ? a.t[a.i_wav + 1] : nullptr;
        const int cin = g.wav ? H : (g.kind == SWN_KIND_LAPLACE ? 1 : Q);
        if (g.kind == SWN_KIND_LAPLACE) {
            for (unsigned e = tid; e < (size_t)K * H; e += nth) {
                const int k = (int)(e / H), o = (int)(e % H);
                if (g.wav) {
                    double sv = 0, sc = 0;
                    for (int i = 0; i < H; ++i) {
                        const double w = cw[((size_t)o * cin + i) * K + k];
                        sv += w * wav_w[i]; sc += w * wav_b[i];
                    }
                    out[y.cv + e] = (float)sv;
                    out[y.cc + e] = (float)sc;
                } else {
                    out[y.cv + e] = cw[(size_t)o * K + k];
                }
            }
        } else {
            for (unsigned e = tid; e < (size_t)K * Q * H; e += nth) {
                const int o = (int)(e % H), q = (int)((e / H) % Q), k = (int)(e / (unsigned)(H * Q));
                float v;
                if (g.wav) {
                    double s = 0;
                    for (int i = 0; i < H; ++i)
                        s += (double)cw[((size_t)o * cin + i) * K + k] * ((double)wav_w[(size_t)i * Q + q] + wav_b[i]);
                    v = (float)s;
                } else {
                    v = cw[((size_t)o * cin + q) * K + k];
                }
                out[y.ct + e] = v;
            }
        }
        break;
    }
    case SEC_INX: {     // stacked rows of the frame-rate GEMM (+ the one-hot columns of audio_in)
        const size_t n = (size_t)L * seg * H2 * g.A0;
        for (unsigned e = tid; e < n; e += nth) {
            const int c = (int)(e % g.A0);
            unsigned r = e / g.A0;
            const int o = (int)(r % H2); r /= H2;
            const int s = (int)(r % seg), l = (int)(r / seg);
            out[y.wx + ((size_t)(l * seg + s) * H2 + o) * g.A0p + c] = inx_eff(a, l, o, c, s);
        }
        if (g.audio_in) {
            const int A = g.A0 * seg + Q;
            for (unsigned e = tid; e < (size_t)L * Q * H2; e += nth) {
                const int o = (int)(e % H2), q = (int)((e / H2) % Q), l = (int)(e / (unsigned)(H2 * Q));
                out[y.wxa + e] = a.t[a.i_inx + 2 * l][(size_t)o * A + g.A0 + q];
            }
        }
        break;
    }
    case SEC_INX_BIAS: {   // bx = b + b_up * sum W_eff (c-major, s-minor, double), bxr = b (+ W_in . b2 with the Conv2d)
        const float bup = a.t[a.i_up + 1][0];
        for (unsigned e = tid; e < (size_t)L * H2; e += nth) {
            const int l = (int)(e / H2), o = (int)(e % H2);
            double bo = a.t[a.i_inx + 2 * l + 1][o], ws = 0.0;
            if (g.conv2d) {
                const float* w = a.t[a.i_inx + 2 * l];
                const float* c2b = a.t[a.i_c2d + 1];
                for (int p = 0; p < g.A0; ++p) bo += (double)w[(size_t)o * g.A0 + p] * c2b[p];
            }
            if (!g.conv2d) {
                // same c-major, s-minor order as the host packer; the loads of 16 terms are issued together (the
                // loop is otherwise one dependent L2 round trip per term: 486 of them per thread)
                const int A = g.A0 * seg + (g.audio_in ? Q : 0), n = g.A0 * seg;
                const float* row = a.t[a.i_inx + 2 * l] + (size_t)o * A;
                int k = 0;
                for (; k + 64 <= n; k += 64) {               // (2 430 terms per row at run.sh's seg = 5: 64 loads per round trip)
                    float v[64];
#pragma unroll
                    for (int u = 0; u < 64; ++u) v[u] = row[k + u];
#pragma unroll
                    for (int u = 0; u < 64; ++u) ws += v[u];
                }
                for (; k + 16 <= n; k += 16) {
                    float v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = row[k + u];
#pragma unroll
                    for (int u = 0; u < 16; ++u) ws += v[u];
                }
                for (; k < n; ++k) ws += row[k];
            } else {
                for (int c = 0; c < g.A0; ++c)
                    for (int s = 0; s < seg; ++s) ws += inx_eff(a, l, o, c, s);
            }
            out[y.bx + e] = (float)(bo + (double)bup * ws);
            out[y.bxr + e] = (float)bo;
        }
        break;
    }
    case SEC_DIL: {     // tap-major rows [l][o][k][i] (+ the register image of the BL6 decode kernel)
        const size_t n = (size_t)L * H2 * K * H;
        for (unsigned e = tid; e < n; e += nth) {
            const int i = (int)(e % H);
            unsigned r = e / H;
            const int k = (int)(r % K); r /= K;
            const int o = (int)(r % H2), l = (int)(r / H2);
            out[y.wd + (((size_t)l * H2 + o) * K + k) * g.Hp + i] = a.t[a.i_dil + 2 * l][((size_t)o * H + i) * K + k];
        }
        if (g.bl6) {
            for (unsigned e = tid; e < (size_t)L * 512 * 32; e += nth) {
                const int me = (int)(e % 16), r = (int)((e / 16) % 2), tt = (int)((e / 32) % 512), l = (int)(e / (32 * 512));
                const int m = me / 4, ee = me % 4, o = tt >> 3, p = tt & 7;
                const int j = 32 * m + 4 * p + ee, k = j / H, i = j % H;
                out[y.wd2 + e] = a.t[a.i_dil + 2 * l][((size_t)(o + r * H) * H + i) * K + k];
            }
        }
        break;
    }
    case SEC_SKIP: {    // skip 1x1s concatenated along the input axis, biases summed over layers in double
        for (unsigned e = tid; e < (size_t)S * L * H; e += nth) {
            const int i = (int)(e % H), l = (int)((e / H) % L), c = (int)(e / (unsigned)(H * L));
            const float v = a.t[a.i_skip + 2 * l][(size_t)c * H + i];
            out[y.wsk + (size_t)c * L * g.Hp + (size_t)l * g.Hp + i] = v;
            if (g.bl6)
                out[y.wsk2 + ((((size_t)l * 4 + i / 16) * S + c) * 4 + (i % 16) / 4) * 4 + (i % 4)] = v;
        }
        for (unsigned c = tid; c < (size_t)S; c += nth) {
            double b = 0.0;
            for (int l = 0; l < L; ++l) b += a.t[a.i_skip + 2 * l + 1][c];
            out[y.bsk + c] = (float)b;
        }
        break;
    }
    case SEC_OUT: {
        const float* w1 = a.t[a.i_out1];
        for (unsigned e = tid; e < (size_t)g.O1 * S; e += nth) {
            const int c = (int)(e % S), o = (int)(e / S);
            out[y.w1 + (size_t)o * g.Sp + c] = w1[e];
            if (g.bl6) out[y.w12 + ((((size_t)(c / 16)) * g.O1 + o) * 4 + (c % 16) / 4) * 4 + (c % 4)] = w1[e];
        }
        const float* w2 = a.t[a.i_out2];
        for (unsigned e = tid; e < (size_t)g.NO * g.O1; e += nth) {
            const int c = (int)(e % g.O1), o = (int)(e / g.O1);
            out[y.w2 + (size_t)o * g.O1p + c] = w2[e];
            if (g.bl6 && g.kind == SWN_KIND_SOFTMAX)
                out[y.w22 + ((((size_t)(c / 16)) * g.NO + o) * 4 + (c % 16) / 4) * 4 + (c % 4)] = w2[e];
        }
        break;
    }
    default: break;
    }
}

extern "C" int swn_pack_params_device(const swn_net_desc* d, const float* const* tensors_dev, int n_tensors,
                                      float* packed_dev, size_t packed_floats, void* stream) {
    PackArgs a;
    int rc = swn_make_geom(d, &a.g);
    if (rc < 0) return rc;
    swn_make_layout(&a.g, &a.y);
    if (!tensors_dev || !packed_dev || n_tensors != swn_tensor_count(&a.g) || n_tensors > SWN_PACK_MAXT ||
        packed_floats < a.y.total)
        return SWN_E_BADARG;
    for (int i = 0; i < n_tensors; ++i) {
        if (!tensors_dev[i]) return SWN_E_BADARG;
        a.t[i] = tensors_dev[i];
    }
    for (int i = n_tensors; i < SWN_PACK_MAXT; ++i) a.t[i] = nullptr;
    int ti = 0;       // state_dict order (config.py param_shapes / swn_pack_params)
    a.i_scale = ti; ti += 2;
    a.i_aux = ti; ti += 2 * a.g.auxl;
    a.i_up = ti; ti += 2;
    a.i_c2d = ti; if (a.g.conv2d) ti += 2;
    a.i_wav = ti; if (a.g.wav) ti += 2;
    a.i_causal = ti; ti += 2;
    a.i_inx = ti; ti += 2 * a.g.L;
    a.i_dil = ti; ti += 2 * a.g.L;
    a.i_skip = ti; ti += 2 * a.g.L;
    a.i_out1 = ti; ti += 2;
    a.i_out2 = ti; ti += 2;
    if (ti != n_tensors) return SWN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    // padding lanes (Hp/Sp/A0p tails, section gaps) must read as zero, like the host packer's memset
    if (hipMemsetAsync(packed_dev, 0, a.y.total * sizeof(float), st) != hipSuccess) return swn_launch_status("pack_params_device memset");
    // enough workgroups for the largest section of the largest nets (run.sh's seg = 5 in_x: 5.6 M elements; with 256 workgroups
    // every thread walked 85 elements through a chain of integer divisions: 161 us)
    dim3 grid(a.y.total > (4u << 20) ? 1024 : 256, SEC_COUNT);
    hipLaunchKernelGGL(pack_params_kernel, grid, dim3(256), 0, st, a, packed_dev);
    return swn_launch_status("pack_params_device");
}
