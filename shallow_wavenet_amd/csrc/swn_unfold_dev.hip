// Device-side unfold of the packed-layout gradients onto the reference's parameter tensors: the chain rule through
// the pack-time folds of swn_pack_params / swn_pack_params_device (bx = b_inx + b_up * sum W ; cv/cc/ct = causal (.)
// wav_conv ; tap-major dil_h ; concatenated out_skip), i.e. what loss.backward() leaves in p.grad for every
// nn.Parameter of CSWNV / DSWNV (train_cswnv_laplace-stftcmplx_shift1.py:868-874).
//
// Why it exists: the same map written with torch ops (nets/_autograd.py unfold_packed_grads) is ~60 tiny launches and
// as many Python dispatches per step; with the fused backward the BL6 training step became bound by the time the HOST
// needs to issue it (2.6 ms of issue against 2.0 ms of kernels).  Here it is one launch that writes straight into the
// gradient tensors.  Reductions accumulate in double (the torch version sums in fp32: results agree to ~1e-7 relative).
// Not covered (SWN_E_UNSUPPORTED, the caller keeps the torch path): aux_conv2d_flag with seg > 1.
#include <hip/hip_runtime.h>
#include "swn_geom.hpp"

#define SWN_PACK_MAXT 128

namespace {

struct UnfoldArgs {
    SwnGeom g;
    SwnLayout y;
    const float* t[SWN_PACK_MAXT];     // parameters (state_dict order)
    float* o[SWN_PACK_MAXT];           // gradient outputs, same order; null = not wanted
    int i_scale, i_aux, i_up, i_wav, i_causal, i_inx, i_dil, i_skip, i_out1, i_out2;
};

enum { U_COPY = 0, U_INX, U_DIL, U_SKIP, U_OUT, U_CAUSAL, U_UPB, U_COUNT };

__device__ inline void copy_n(float* dst, const float* src, size_t n, unsigned tid, unsigned nth) {
    if (!dst) return;
    for (unsigned i = tid; i < n; i += nth) dst[i] = src[i];
}

__global__ void __launch_bounds__(256) unfold_grads_kernel(const UnfoldArgs a, const float* __restrict__ gp) {
    const SwnGeom& g = a.g;
    const SwnLayout& y = a.y;
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;      // 32-bit element indices: every section is far below 2^31
    // (64-bit divisions per element made these two kernels 160-250 us at the run.sh geometry)
    const unsigned nth = gridDim.x * blockDim.x;
    const int H = g.H, S = g.S, K = g.K, L = g.L, seg = g.seg, Q = g.Q, H2 = 2 * g.H;
    switch (blockIdx.y) {
    case U_COPY: {
        copy_n(a.o[a.i_scale], gp + y.scale_w, (size_t)g.n_aux * g.n_aux, tid, nth);
        copy_n(a.o[a.i_scale + 1], gp + y.scale_b, g.n_aux, tid, nth);
        for (int i = 0; i < g.auxl; ++i) {
            copy_n(a.o[a.i_aux + 2 * i], gp + y.aux_w[i], (size_t)g.aux_cout[i] * g.aux_cin[i] * g.auxk, tid, nth);
            copy_n(a.o[a.i_aux + 2 * i + 1], gp + y.aux_b[i], g.aux_cout[i], tid, nth);
        }
        copy_n(a.o[a.i_up], gp + y.wup, g.U, tid, nth);
        copy_n(a.o[a.i_causal + 1], gp + y.cb, H, tid, nth);
        for (int l = 0; l < L; ++l) {
            copy_n(a.o[a.i_dil + 2 * l + 1], gp + y.bd + (size_t)l * H2, H2, tid, nth);
            copy_n(a.o[a.i_skip + 2 * l + 1], gp + y.bsk, S, tid, nth);          // the packed skip bias is the sum over layers
            if (float* d = a.o[a.i_inx + 2 * l + 1])                              // d b_inx: hoisted mode + dropout mode
                for (unsigned i = tid; i < (size_t)H2; i += nth) d[i] = gp[y.bx + (size_t)l * H2 + i] + gp[y.bxr + (size_t)l * H2 + i];
        }
        copy_n(a.o[a.i_out1 + 1], gp + y.b1, g.O1, tid, nth);
        copy_n(a.o[a.i_out2 + 1], gp + y.b2, g.NO, tid, nth);
        break;
    }
    case U_INX: {       // d in_x.W[l][o][c*seg+s] = gwx[l][s][o][c] + gbx[l][o] * b_up  (+ the one-hot columns of audio_in)
        const float bup = a.t[a.i_up + 1][0];
        const int A = g.A0 * seg + (g.audio_in ? Q : 0), n1 = g.A0 * seg;
        for (unsigned e = tid; e < (size_t)L * H2 * A; e += nth) {
            const int j = (int)(e % A);
            unsigned r = e / A;
            const int o = (int)(r % H2), l = (int)(r / H2);
            float* d = a.o[a.i_inx + 2 * l];
            if (!d) continue;
            float v;
            if (j < n1) {
                const int c = j / seg, s = j - c * seg;
                v = gp[y.wx + ((size_t)(l * seg + s) * H2 + o) * g.A0p + c] + gp[y.bx + (size_t)l * H2 + o] * bup;
            } else {
                v = gp[y.wxa + ((size_t)l * Q + (j - n1)) * H2 + o];
            }
            d[(size_t)o * A + j] = v;
        }
        break;
    }
    case U_DIL: {
        for (unsigned e = tid; e < (size_t)L * H2 * H * K; e += nth) {
            const int k = (int)(e % K);
            unsigned r = e / K;
            const int i = (int)(r % H); r /= H;
            const int o = (int)(r % H2), l = (int)(r / H2);
            if (float* d = a.o[a.i_dil + 2 * l]) d[((size_t)o * H + i) * K + k] = gp[y.wd + (((size_t)l * H2 + o) * K + k) * g.Hp + i];
        }
        break;
    }
    case U_SKIP: {
        for (unsigned e = tid; e < (size_t)L * S * H; e += nth) {
            const int i = (int)(e % H), c = (int)((e / H) % S), l = (int)(e / (unsigned)(H * S));
            if (float* d = a.o[a.i_skip + 2 * l]) d[(size_t)c * H + i] = gp[y.wsk + (size_t)c * L * g.Hp + (size_t)l * g.Hp + i];
        }
        break;
    }
    case U_OUT: {
        if (float* d = a.o[a.i_out1])
            for (unsigned e = tid; e < (size_t)g.O1 * S; e += nth) d[e] = gp[y.w1 + (e / S) * g.Sp + (e % S)];
        if (float* d = a.o[a.i_out2])
            for (unsigned e = tid; e < (size_t)g.NO * g.O1; e += nth) d[e] = gp[y.w2 + (e / g.O1) * g.O1p + (e % g.O1)];
        break;
    }
    case U_CAUSAL: {    // causal.conv.weight (H, Cin, K) and the lift wav_conv, through cv/cc (laplace) or the gather table ct
        const float* wc = a.t[a.i_causal];
        float* dwc = a.o[a.i_causal];
        if (g.kind == SWN_KIND_LAPLACE) {
            const float* gcv = gp + y.cv;
            const float* gcc = gp + y.cc;
            if (!g.wav) {
                if (dwc) for (unsigned e = tid; e < (size_t)H * K; e += nth) dwc[e] = gcv[(e % K) * H + e / K];
                break;
            }
            const float* ww = a.t[a.i_wav];
            const float* wb = a.t[a.i_wav + 1];
            if (dwc)
                for (unsigned e = tid; e < (size_t)H * H * K; e += nth) {
                    const int k = (int)(e % K), i = (int)((e / K) % H), o = (int)(e / (unsigned)(K * H));
                    dwc[e] = gcv[k * H + o] * ww[i] + gcc[k * H + o] * wb[i];
                }
            // one wave per lifted channel i, its lanes over the H*K taps (a thread per i walked 1 344 strided loads one after
            // the other at the run.sh geometry: ~200 us, the whole kernel's time)
            for (unsigned i = tid >> 6; i < (unsigned)H; i += nth >> 6) {
                double sw = 0.0, sb = 0.0;
                for (int e = (int)(threadIdx.x & 63); e < H * K; e += 64) {
                    const int o = e / K, k = e - o * K;
                    const double w = wc[((size_t)o * H + i) * K + k];
                    sw += w * gcv[k * H + o]; sb += w * gcc[k * H + o];
                }
#pragma unroll
                for (int sft = 32; sft > 0; sft >>= 1) { sw += __shfl_xor(sw, sft); sb += __shfl_xor(sb, sft); }
                if ((threadIdx.x & 63) == 0) {
                    if (a.o[a.i_wav]) a.o[a.i_wav][i] = (float)sw;
                    if (a.o[a.i_wav + 1]) a.o[a.i_wav + 1][i] = (float)sb;
                }
            }
        } else {
            const float* gct = gp + y.ct;                                 // [K][Q][H]
            if (!g.wav) {
                if (dwc)
                    for (unsigned e = tid; e < (size_t)H * Q * K; e += nth) {
                        const int k = (int)(e % K), q = (int)((e / K) % Q), o = (int)(e / (unsigned)(K * Q));
                        dwc[e] = gct[((size_t)k * Q + q) * H + o];
                    }
                break;
            }
            const float* ww = a.t[a.i_wav];                               // (H, Q, 1)
            const float* wb = a.t[a.i_wav + 1];
            if (dwc)        // d wc[o][i][k] = sum_q gct[k][q][o] * (ww[i][q] + wb[i]);  o fastest across threads: coalesced gct reads
                for (unsigned e = tid; e < (size_t)H * H * K; e += nth) {
                    const int o = (int)(e % H), i = (int)((e / H) % H), k = (int)(e / (unsigned)(H * H));
                    double s = 0.0;
                    const float bi = wb[i];
                    for (int q = 0; q < Q; ++q) s += (double)gct[((size_t)k * Q + q) * H + o] * (double)(ww[(size_t)i * Q + q] + bi);
                    dwc[((size_t)o * H + i) * K + k] = (float)s;
                }
            // d ww[i][q] = sum_{k,o} gct[k][q][o] * wc[o][i][k];   d wb[i] = sum_q of the same terms (zeroed by the host, atomics)
            for (unsigned e = tid; e < (size_t)H * Q; e += nth) {
                const int q = (int)(e % Q), i = (int)(e / Q);
                double s = 0.0;
                for (int k = 0; k < K; ++k)
                    for (int o = 0; o < H; ++o) s += (double)gct[((size_t)k * Q + q) * H + o] * wc[((size_t)o * H + i) * K + k];
                if (a.o[a.i_wav]) a.o[a.i_wav][e] = (float)s;
                if (a.o[a.i_wav + 1]) atomicAdd(a.o[a.i_wav + 1] + i, (float)s);
            }
        }
        break;
    }
    case U_UPB: {       // d b_up = sum_{l,o} gbx[l][o] * sum_j W_inx[l][o][j] + gbup      (zeroed by the host, atomics)
        float* d = a.o[a.i_up + 1];
        if (!d) break;
        __shared__ double red[256];
        const int A = g.A0 * seg + (g.audio_in ? Q : 0), n1 = g.A0 * seg;
        double part = 0.0;
        for (int row = blockIdx.x; row < L * H2; row += gridDim.x) {
            const int l = row / H2, o = row - l * H2;
            const float* wr = a.t[a.i_inx + 2 * l] + (size_t)o * A;
            double s = 0.0;
            for (int j = threadIdx.x; j < n1; j += 256) s += wr[j];
            red[threadIdx.x] = s;
            __syncthreads();
            for (int sft = 128; sft > 0; sft >>= 1) { if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft]; __syncthreads(); }
            if (threadIdx.x == 0) part += red[0] * (double)gp[y.bx + row];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            if (blockIdx.x == 0) part += gp[y.bup];
            atomicAdd(d, (float)part);
        }
        break;
    }
    default: break;
    }
}

}  // namespace

extern "C" int swn_unfold_grads_device(const swn_net_desc* d, const float* gpacked_dev, const float* const* tensors_dev,
                                       float* const* grads_dev, int n_tensors, void* stream) {
    UnfoldArgs a;
    int rc = swn_make_geom(d, &a.g);
    if (rc < 0) return rc;
    if (a.g.conv2d) return SWN_E_UNSUPPORTED;
    swn_make_layout(&a.g, &a.y);
    if (!gpacked_dev || !tensors_dev || !grads_dev || n_tensors != swn_tensor_count(&a.g) || n_tensors > SWN_PACK_MAXT)
        return SWN_E_BADARG;
    for (int i = 0; i < SWN_PACK_MAXT; ++i) { a.t[i] = nullptr; a.o[i] = nullptr; }
    for (int i = 0; i < n_tensors; ++i) {
        if (!tensors_dev[i]) return SWN_E_BADARG;
        a.t[i] = tensors_dev[i]; a.o[i] = grads_dev[i];
    }
    int ti = 0;       // state_dict order (config.py param_shapes / swn_pack_params)
    a.i_scale = ti; ti += 2;
    a.i_aux = ti; ti += 2 * a.g.auxl;
    a.i_up = ti; ti += 2;
    a.i_wav = ti; if (a.g.wav) ti += 2;
    a.i_causal = ti; ti += 2;
    a.i_inx = ti; ti += 2 * a.g.L;
    a.i_dil = ti; ti += 2 * a.g.L;
    a.i_skip = ti; ti += 2 * a.g.L;
    a.i_out1 = ti; ti += 2;
    a.i_out2 = ti; ti += 2;
    if (ti != n_tensors) return SWN_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    (void)hipGetLastError();
    // the two outputs that are accumulated atomically
    if (a.o[a.i_up + 1] && hipMemsetAsync(a.o[a.i_up + 1], 0, sizeof(float), st) != hipSuccess) return SWN_E_LAUNCH;
    if (a.g.kind == SWN_KIND_SOFTMAX && a.g.wav && a.o[a.i_wav + 1] &&
        hipMemsetAsync(a.o[a.i_wav + 1], 0, (size_t)a.g.H * sizeof(float), st) != hipSuccess) return SWN_E_LAUNCH;
    hipLaunchKernelGGL(unfold_grads_kernel, dim3(a.y.total > (4u << 20) ? 768 : 128, U_COUNT), dim3(256), 0, st, a, gpacked_dev);
    return swn_launch_status("swn_unfold_grads_device");
}
