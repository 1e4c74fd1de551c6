// Host-side parameter packing: reference state_dict tensors -> one flat fp32 buffer.
// (replaces the per-process torch.load + load_state_dict of
//  decode_cswnv_laplace-shift1.py:223-224 as the thing that reaches the GPU; the packed
//  buffer is uploaded once and broadcast over RCCL.)
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "swn_geom.hpp"

extern "C" int swn_abi_version(void) { return SWN_ABI_VERSION; }

static thread_local char g_detail[256] = "";
extern "C" void swn_set_error_detail(const char* where, const char* what) {
    snprintf(g_detail, sizeof(g_detail), "%s: %s", where ? where : "?", what ? what : "?");
}
extern "C" const char* swn_last_error_detail(void) { return g_detail; }

extern "C" const char* swn_strerror(int code) {
    switch (code) {
        case SWN_OK: return "ok";
        case SWN_E_BADDESC: return "network descriptor outside supported range";
        case SWN_E_BADARG: return "bad argument (null pointer or size mismatch)";
        case SWN_E_LAUNCH: return "HIP kernel launch failed";
        case SWN_E_UNSUPPORTED: return "configuration not supported by this build";
        case SWN_E_NODEVICE: return "no HIP device";
        default: return "unknown error";
    }
}

extern "C" int swn_receptive_field(const swn_net_desc* d) {
    SwnGeom g; int rc = swn_make_geom(d, &g);
    return rc < 0 ? rc : g.rf;
}

extern "C" int swn_num_tensors(const swn_net_desc* d) {
    SwnGeom g; int rc = swn_make_geom(d, &g);
    return rc < 0 ? rc : swn_tensor_count(&g);
}

extern "C" size_t swn_packed_floats(const swn_net_desc* d) {
    SwnGeom g; if (swn_make_geom(d, &g) < 0) return 0;
    SwnLayout y; swn_make_layout(&g, &y);
    return y.total;
}

// float offsets of the packed sections, fixed order (mirrored by shallow_wavenet_amd/runtime.py LAYOUT_FIELDS)
extern "C" int swn_layout_offsets(const swn_net_desc* d, size_t* out, int n) {
    SwnGeom g; int rc = swn_make_geom(d, &g);
    if (rc < 0) return rc;
    SwnLayout y; swn_make_layout(&g, &y);
    const size_t v[] = {y.scale_w, y.scale_b, y.aux_w[0], y.aux_w[1], y.aux_w[2], y.aux_w[3], y.aux_b[0], y.aux_b[1],
                        y.aux_b[2], y.aux_b[3], y.wx, y.wxa, y.wup, y.bup, y.bx, y.cb, y.cv, y.cc, y.ct, y.wd, y.bd,
                        y.wsk, y.bsk, y.w1, y.b1, y.w2, y.b2, y.total, y.bxr};
    const int cnt = (int)(sizeof(v) / sizeof(v[0]));
    if (!out || n < cnt) return SWN_E_BADARG;
    for (int i = 0; i < cnt; ++i) out[i] = v[i];
    return cnt;
}

extern "C" int swn_pack_params(const swn_net_desc* d, const float* const* t, int n_tensors,
                               float* out, size_t out_floats) {
    SwnGeom g; int rc = swn_make_geom(d, &g);
    if (rc < 0) return rc;
    SwnLayout y; swn_make_layout(&g, &y);
    if (!t || !out || n_tensors != swn_tensor_count(&g) || out_floats < y.total) return SWN_E_BADARG;
    for (int i = 0; i < n_tensors; ++i) if (!t[i]) return SWN_E_BADARG;
    memset(out, 0, y.total * sizeof(float));
    const int H = g.H, S = g.S, K = g.K, L = g.L, seg = g.seg, Q = g.Q, H2 = 2 * g.H;
    int ti = 0;
    // ---- frame-rate section: scale_in, conv_aux (copied as they are)
    memcpy(out + y.scale_w, t[ti++], sizeof(float) * g.n_aux * g.n_aux);
    memcpy(out + y.scale_b, t[ti++], sizeof(float) * g.n_aux);
    for (int i = 0; i < g.auxl; ++i) {
        memcpy(out + y.aux_w[i], t[ti++], sizeof(float) * g.aux_cout[i] * g.aux_cin[i] * g.auxk);
        memcpy(out + y.aux_b[i], t[ti++], sizeof(float) * g.aux_cout[i]);
    }
    // ---- rank-1 upsampler
    const float* wup = t[ti++];
    const float* bup = t[ti++];
    memcpy(out + y.wup, wup, sizeof(float) * g.U);
    out[y.bup] = bup[0];
    // ---- optional (seg,1) Conv2d between the upsampler and in_x (cswnv_shift1.py:196-198): linear, so it is
    //      folded into in_x below:  W_eff[o][c*seg+s] = sum_p in_x.W[o][p] * conv2d.W[p][c][s]
    const float* c2w = nullptr; const float* c2b = nullptr;
    if (g.conv2d) { c2w = t[ti++]; c2b = t[ti++]; }
    // ---- sample lift (optional) and causal input layer
    const float* wav_w = nullptr; const float* wav_b = nullptr;
    if (g.wav) { wav_w = t[ti++]; wav_b = t[ti++]; }
    const float* cw = t[ti++];          // (H, Cin, K)
    const float* cbias = t[ti++];
    memcpy(out + y.cb, cbias, sizeof(float) * H);
    const int cin = g.wav ? H : (g.kind == SWN_KIND_LAPLACE ? 1 : Q);
    if (g.kind == SWN_KIND_LAPLACE) {
        for (int k = 0; k < K; ++k)
            for (int o = 0; o < H; ++o) {
                if (g.wav) {
                    double sv = 0, sc = 0;
                    for (int i = 0; i < H; ++i) {
                        double w = cw[((size_t)o * cin + i) * K + k];
                        sv += w * wav_w[i]; sc += w * wav_b[i];
                    }
                    out[y.cv + (size_t)k * H + o] = (float)sv;
                    out[y.cc + (size_t)k * H + o] = (float)sc;
                } else {
                    out[y.cv + (size_t)k * H + o] = cw[(size_t)o * K + k];
                }
            }
    } else {
        for (int k = 0; k < K; ++k)
            for (int q = 0; q < Q; ++q)
                for (int o = 0; o < H; ++o) {
                    float v;
                    if (g.wav) {
                        double s = 0;
                        for (int i = 0; i < H; ++i)
                            s += (double)cw[((size_t)o * cin + i) * K + k] * ((double)wav_w[(size_t)i * Q + q] + wav_b[i]);
                        v = (float)s;
                    } else {
                        v = cw[((size_t)o * cin + q) * K + k];
                    }
                    out[y.ct + ((size_t)k * Q + q) * H + o] = v;
                }
    }
    // ---- in_x: stacked rows for the frame-rate GEMM, bias folded with the upsampler bias
    const int A = g.A0 * seg + (g.audio_in ? Q : 0);
    std::vector<double> eff(g.conv2d ? (size_t)g.A0 * seg : 0);
    for (int l = 0; l < L; ++l) {
        const float* w = t[ti++];       // (2H, A, 1); with the Conv2d: (2H, A0, 1)
        const float* b = t[ti++];
        for (int o = 0; o < H2; ++o) {
            double ws = 0, bo = b[o];
            if (g.conv2d) {
                std::fill(eff.begin(), eff.end(), 0.0);
                for (int p = 0; p < g.A0; ++p) {
                    const double wp = w[(size_t)o * g.A0 + p];
                    const float* row = c2w + (size_t)p * g.A0 * seg;      // [c][s] of output channel p
                    for (int cs = 0; cs < g.A0 * seg; ++cs) eff[cs] += wp * row[cs];
                    bo += wp * c2b[p];
                }
            }
            for (int c = 0; c < g.A0; ++c)
                for (int s = 0; s < seg; ++s) {
                    const float v = g.conv2d ? (float)eff[(size_t)c * seg + s] : w[(size_t)o * A + c * seg + s];
                    ws += v;
                    out[y.wx + ((size_t)(l * seg + s) * H2 + o) * g.A0p + c] = v;
                }
            out[y.bx + (size_t)l * H2 + o] = (float)(bo + (double)bup[0] * ws);
            out[y.bxr + (size_t)l * H2 + o] = (float)bo;
            if (g.audio_in)
                for (int q = 0; q < Q; ++q)
                    out[y.wxa + ((size_t)l * Q + q) * H2 + o] = w[(size_t)o * A + g.A0 + q];
        }
    }
    // ---- dilated convs: tap-major rows [o][k][i]
    for (int l = 0; l < L; ++l) {
        const float* w = t[ti++];       // (2H, H, K)
        const float* b = t[ti++];
        for (int o = 0; o < H2; ++o)
            for (int i = 0; i < H; ++i)
                for (int k = 0; k < K; ++k)
                    out[y.wd + (((size_t)l * H2 + o) * K + k) * g.Hp + i] = w[((size_t)o * H + i) * K + k];
        memcpy(out + y.bd + (size_t)l * H2, b, sizeof(float) * H2);
        if (g.bl6) {
            // thread t = o*8+p keeps rows (o, o+H); element 4m+e is input j = 32m+4p+e of [tap0|tap1]
            for (int tt = 0; tt < 512; ++tt) {
                const int o = tt >> 3, p = tt & 7;
                for (int r = 0; r < 2; ++r)
                    for (int m = 0; m < 4; ++m)
                        for (int e = 0; e < 4; ++e) {
                            const int j = 32 * m + 4 * p + e, k = j / H, i = j % H;
                            out[y.wd2 + (((size_t)l * 512 + tt) * 2 + r) * 16 + 4 * m + e] =
                                w[((size_t)(o + r * H) * H + i) * K + k];
                        }
            }
        }
    }
    // ---- skip 1x1s concatenated along the input axis, biases summed
    std::vector<double> bs(S, 0.0);
    for (int l = 0; l < L; ++l) {
        const float* w = t[ti++];       // (S, H, 1)
        const float* b = t[ti++];
        for (int c = 0; c < S; ++c) {
            for (int i = 0; i < H; ++i) {
                out[y.wsk + (size_t)c * L * g.Hp + (size_t)l * g.Hp + i] = w[(size_t)c * H + i];
                if (g.bl6)   // [l][mm][row][pp][e], input i = 16mm+4pp+e
                    out[y.wsk2 + ((((size_t)l * 4 + i / 16) * S + c) * 4 + (i % 16) / 4) * 4 + (i % 4)] =
                        w[(size_t)c * H + i];
            }
            bs[c] += b[c];
        }
    }
    for (int c = 0; c < S; ++c) out[y.bsk + c] = (float)bs[c];
    // ---- output 1x1s
    {
        const float* w = t[ti++]; const float* b = t[ti++];      // (O1, S, 1)
        for (int o = 0; o < g.O1; ++o)
            memcpy(out + y.w1 + (size_t)o * g.Sp, w + (size_t)o * S, sizeof(float) * S);
        memcpy(out + y.b1, b, sizeof(float) * g.O1);
        if (g.bl6)
            for (int o = 0; o < g.O1; ++o)
                for (int c = 0; c < S; ++c)
                    out[y.w12 + ((((size_t)(c / 16)) * g.O1 + o) * 4 + (c % 16) / 4) * 4 + (c % 4)] =
                        w[(size_t)o * S + c];
    }
    {
        const float* w = t[ti++]; const float* b = t[ti++];      // (NO, O1, 1)
        for (int o = 0; o < g.NO; ++o)
            memcpy(out + y.w2 + (size_t)o * g.O1p, w + (size_t)o * g.O1, sizeof(float) * g.O1);
        memcpy(out + y.b2, b, sizeof(float) * g.NO);
        if (g.bl6 && g.kind == SWN_KIND_SOFTMAX)
            for (int o = 0; o < g.NO; ++o)
                for (int c = 0; c < g.O1; ++c)
                    out[y.w22 + ((((size_t)(c / 16)) * g.NO + o) * 4 + (c % 16) / 4) * 4 + (c % 4)] =
                        w[(size_t)o * g.O1 + c];
    }
    return ti == n_tensors ? SWN_OK : SWN_E_BADARG;
}
