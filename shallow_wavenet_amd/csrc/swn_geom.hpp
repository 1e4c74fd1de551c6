// Geometry and packed-parameter layout shared by the host packer and the gfx950 kernels.
//
// Everything is derived from the reference constructor arguments
// (cswnv_shift1.py:130-189, dswnv.py:190-248): dilation K^(l mod dd), padding
// K^(d+1)-K^d, rf = sum(padding)+K-1, in_x width A = 9*n_aux*seg, head width 2*seg+lpc | Q.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include "../../include/swn_hip.h"

#ifndef __HIPCC__
#define __host__
#define __device__
#endif

#define SWN_MAXL 16      // stack layers (dilation_depth * dilation_repeat)
#define SWN_MAXAUX 4     // conditioning conv layers

static inline __host__ __device__ int swn_round4(int x) { return (x + 3) & ~3; }

struct SwnGeom {
    int kind, n_aux, H, S, K, L, U, seg, lpc, Q, wav, audio_in;
    int conv2d;    // aux_conv2d_flag with seg > 1: a (seg,1) Conv2d ahead of in_x, folded into in_x at pack time
    int auxk, auxl;
    int A0;        // conditioning channels after conv_aux = n_aux * auxk^auxl
    int O1;        // out_1 width: S (laplace) | Q (softmax)
    int NO;        // out_2 width: 2*seg+lpc | Q
    int rf;
    int Hp;        // H rounded up to 4 (row stride of every H-vector)
    int Sp, O1p, A0p;
    int N;         // conditioning row width per frame: L*seg*2H
    int bl6;       // 1 when the stack is the BASELINE-literal 1x6, H=64, K=2 shape (fast decode kernel)
    int dil[SWN_MAXL];
    int pad[SWN_MAXL];
    int aux_cin[SWN_MAXAUX], aux_cout[SWN_MAXAUX], aux_dil[SWN_MAXAUX], aux_pad[SWN_MAXAUX];
};

// float offsets into the packed parameter buffer
struct SwnLayout {
    // ---- frame-rate section
    size_t scale_w, scale_b;                 // [n_aux][n_aux], [n_aux]
    size_t aux_w[SWN_MAXAUX], aux_b[SWN_MAXAUX];   // [cout][cin][k], [cout]
    size_t wx;                               // [N][A0p]  row n=(l*seg+s)*2H+o : in_x[l].weight[o][c*seg+s]
    size_t wxa;                              // softmax audio_in: [L][Q][2H]
    // ---- sample-rate section
    size_t wup, bup;                         // [U], [1]
    size_t bx;                               // [L][2H] = in_x bias + b_up * sum_c,s W
    size_t bxr;                              // [L][2H] = in_x bias alone (dropout mode: in_x runs at sample rate)
    size_t cb;                               // causal bias [H]
    size_t cv, cc;                           // laplace: fused lift+causal taps [K][H] (value, constant)
    size_t ct;                               // softmax: gather table [K][Q][H]
    size_t wd, bd;                           // [L][2H][K][Hp], [L][2H]
    size_t wsk, bsk;                         // [S][L*Hp], [S] (biases summed over layers)
    size_t w1, b1;                           // [O1][Sp], [O1]
    size_t w2, b2;                           // [NO][O1p], [NO]
    // ---- lane-tiled copies for the BL6-class decode kernel (bl6 only)
    size_t wd2;                              // [L][512 threads][2 rows][16]   register-resident dil_h
    size_t wsk2;                             // [L][4][S][4][4]   out_skip, 4 lanes per row
    size_t w12;                              // [S/16][O1][4][4]  out_1
    size_t w22;                              // [O1/16][NO][4][4] out_2 (softmax only)
    size_t total;
};

static inline __host__ int swn_make_geom(const swn_net_desc* d, SwnGeom* g) {
    if (!d || !g) return SWN_E_BADARG;
    if (d->kind != SWN_KIND_LAPLACE && d->kind != SWN_KIND_SOFTMAX) return SWN_E_BADDESC;
    int L = d->dilation_depth * d->dilation_repeat;
    if (d->n_aux < 1 || d->hid_chn < 4 || d->skip_chn < 1 || d->kernel_size < 2 ||
        d->dilation_depth < 1 || d->dilation_repeat < 1 || L > SWN_MAXL ||
        d->upsampling_factor < 1 || d->aux_kernel_size < 1 || (d->aux_kernel_size & 1) == 0 ||
        d->aux_dilation_size < 1 || d->aux_dilation_size > SWN_MAXAUX)
        return SWN_E_BADDESC;
    g->kind = d->kind; g->n_aux = d->n_aux; g->H = d->hid_chn; g->S = d->skip_chn;
    g->K = d->kernel_size; g->L = L; g->U = d->upsampling_factor;
    g->wav = d->wav_conv_flag ? 1 : 0;
    if (d->kind == SWN_KIND_LAPLACE) {
        if (d->seg < 1 || d->seg > 10 || d->lpc < 0 || d->lpc > 16) return SWN_E_BADDESC;
        g->conv2d = (d->aux_conv2d_flag && d->seg > 1) ? 1 : 0;
        g->seg = d->seg; g->lpc = d->lpc; g->Q = 0; g->audio_in = 0;
        g->O1 = g->S; g->NO = 2 * g->seg + g->lpc;
    } else {
        if (d->n_quantize < 2 || d->n_quantize > 4096) return SWN_E_BADDESC;
        g->seg = 1; g->lpc = 0; g->Q = d->n_quantize; g->audio_in = d->audio_in_flag ? 1 : 0; g->conv2d = 0;
        g->O1 = g->Q; g->NO = g->Q;
    }
    g->auxk = d->aux_kernel_size; g->auxl = d->aux_dilation_size;
    int c = d->n_aux, kp = 1;
    for (int i = 0; i < g->auxl; ++i) {
        g->aux_cin[i] = c; c *= g->auxk; g->aux_cout[i] = c;
        g->aux_dil[i] = kp; g->aux_pad[i] = (kp * g->auxk - kp) / 2; kp *= g->auxk;
    }
    g->A0 = c;
    long rf = g->K - 1;
    for (int l = 0; l < L; ++l) {
        long dl = 1;
        for (int e = 0; e < (l % d->dilation_depth); ++e) dl *= g->K;
        if (dl * g->K > (1 << 24)) return SWN_E_BADDESC;
        g->dil[l] = (int)dl; g->pad[l] = (int)(dl * g->K - dl);
        rf += g->pad[l];
    }
    g->rf = (int)rf;
    g->Hp = swn_round4(g->H); g->Sp = swn_round4(g->S); g->O1p = swn_round4(g->O1);
    g->A0p = swn_round4(g->A0);
    g->N = L * g->seg * 2 * g->H;
    g->bl6 = (g->H == 64 && g->K == 2 && d->dilation_depth == 6 && d->dilation_repeat == 1 &&
              (g->S % 16) == 0 && (g->O1 % 16) == 0) ? 1 : 0;
    return SWN_OK;
}

static inline __host__ size_t swn_al(size_t x) { return (x + 63) & ~(size_t)63; }   // 256-B sections

static inline __host__ void swn_make_layout(const SwnGeom* g, SwnLayout* y) {
    size_t o = 0;
    y->scale_w = o; o = swn_al(o + (size_t)g->n_aux * g->n_aux);
    y->scale_b = o; o = swn_al(o + g->n_aux);
    for (int i = 0; i < SWN_MAXAUX; ++i) { y->aux_w[i] = 0; y->aux_b[i] = 0; }
    for (int i = 0; i < g->auxl; ++i) {
        y->aux_w[i] = o; o = swn_al(o + (size_t)g->aux_cout[i] * g->aux_cin[i] * g->auxk);
        y->aux_b[i] = o; o = swn_al(o + g->aux_cout[i]);
    }
    y->wx = o; o = swn_al(o + (size_t)g->N * g->A0p);
    y->wxa = o; if (g->audio_in) o = swn_al(o + (size_t)g->L * g->Q * 2 * g->H);
    y->wup = o; o = swn_al(o + g->U);
    y->bup = o; o = swn_al(o + 1);
    y->bx = o; o = swn_al(o + (size_t)g->L * 2 * g->H);
    y->bxr = o; o = swn_al(o + (size_t)g->L * 2 * g->H);
    y->cb = o; o = swn_al(o + g->H);
    y->cv = o; y->cc = o; y->ct = o;
    if (g->kind == SWN_KIND_LAPLACE) {
        y->cv = o; o = swn_al(o + (size_t)g->K * g->H);
        y->cc = o; o = swn_al(o + (size_t)g->K * g->H);
    } else {
        y->ct = o; o = swn_al(o + (size_t)g->K * g->Q * g->H);
    }
    y->wd = o; o = swn_al(o + (size_t)g->L * 2 * g->H * g->K * g->Hp);
    y->bd = o; o = swn_al(o + (size_t)g->L * 2 * g->H);
    y->wsk = o; o = swn_al(o + (size_t)g->S * g->L * g->Hp);
    y->bsk = o; o = swn_al(o + g->S);
    y->w1 = o; o = swn_al(o + (size_t)g->O1 * g->Sp);
    y->b1 = o; o = swn_al(o + g->O1);
    y->w2 = o; o = swn_al(o + (size_t)g->NO * g->O1p);
    y->b2 = o; o = swn_al(o + g->NO);
    y->wd2 = y->wsk2 = y->w12 = y->w22 = o;
    if (g->bl6) {
        y->wd2 = o; o = swn_al(o + (size_t)g->L * 512 * 32);
        y->wsk2 = o; o = swn_al(o + (size_t)g->L * g->S * 64);
        y->w12 = o; o = swn_al(o + (size_t)g->O1 * g->S);
        y->w22 = o; if (g->kind == SWN_KIND_SOFTMAX) o = swn_al(o + (size_t)g->NO * g->O1);
    }
    y->total = o;
}

// dropout mode: channel rows per utterance of the masked, upsampled conditioning xm (B, A0x, Tx): A0 rounded up to 32 with
// zero rows, so that the sample-rate in_x GEMM (k = conditioning channel) runs whole 32-wide k-tiles (A0 = 486 took the
// generic 64 x 64 kernel: 0.36 ms per layer at the run.sh geometry against ~0.15 for the tile-uniform form)
static inline __host__ int swn_a0x(const SwnGeom* g) { return (g->A0 + 31) & ~31; }

// dropout mode in the mixed-precision mode: does swn_forward_drop run the gated layers and the two wide head layers on
// bf16 operands (and keep the gate pre-activations for swn_backward_drop)?  The same classes as the bf16 forward without
// dropout (H a multiple of 64): smaller nets keep the exact-fp32 forward, only their contractions of the backward are rounded.
static inline __host__ bool swn_drop_bf16_forward(const SwnGeom* g) { return g->Hp == g->H && (g->H % 64) == 0; }

// ... and, for seg == 1 at such nets, the three sample-rate in_x contractions of the dropout chain (forward product, weight
// gradient, data gradient over every layer's kept d gx) read bf16 COPIES of their operands - the masked conditioning xm, d gx and
// the in_x matrix - instead of rounding fp32 tensors on the way into LDS: the same products (the rounding is the one the fp32
// route applies while staging), half the operand bytes, the kernels of the layer gradients.  Rows of swn_pitch16() elements.
static inline __host__ bool swn_drop_inx16(const SwnGeom* g, long Tp) { return swn_drop_bf16_forward(g) && g->seg == 1 && Tp >= 256; }
static inline __host__ long swn_pitch16(long T) { return (T + 2 + 31) & ~31L; }

// "G4" layout of a (B, R, Tp) fp32 tensor that the MFMA epilogues of the mixed-precision GEMM-stack path write and read (R % 4 == 0):
// blocks of 16 positions x all R rows; inside a block the rows go in groups of four, [row / 4][t % 16][row % 4]; the last block is
// Tp % 16 positions wide (no padding: B * R * Tp floats like the plain layout).  An accumulator lane - four consecutive rows of one
// position - is then ONE 16-byte piece and the 16 lanes of a column group 256 contiguous bytes, where the plain (B, R, Tp) layout
// costs four 4-byte accesses in 64-byte runs (measured on the gated layer of the run.sh geometry: +65 us for storing the
// pre-activations, +100 us for reading the sample-rate in_x rows, of 206).  Tensors in this layout: the kept gate
// pre-activations of csrc/swn_stack_bf16g.hip and the in_x rows gx of the dropout mode when that stack runs the forward.
static inline __host__ __device__ size_t swn_g4(int R, int Tp, int b, int row, int t) {
    const int tb = t >> 4, wid = Tp - 16 * tb < 16 ? Tp - 16 * tb : 16;
    return ((size_t)b * Tp + 16 * (size_t)tb) * R + ((size_t)(row >> 2) * wid + (t & 15)) * 4 + (row & 3);
}
// does the dropout-mode forward of the mixed-precision mode run on the bf16 time-major GEMM stack (csrc/swn_stack.hip)?
bool swn_drop_g16(const swn_net_desc* d, int batch, long Tp);

// BL6 class, mixed-precision mode, dropout as run.sh trains it (dilation_repeat == 1: the only hidden-state mask lands on the
// last layer's output, which feeds nothing, so aux_drop is the one mask that acts - cswnv_shift1.py:194-195,211-217): the
// forward work buffer of the fused path (csrc/swn_stack_bf16.hip: swn_bl6_drop_forward; read back by swn_bl6_bwd_stack).
// Byte offsets, every section 256-byte aligned.
struct SwnBl6DropLayout {
    size_t hs16;       // [L+1][B][Tp][64] bf16 hidden states, time-major
    size_t wbf;        // fragment-ordered bf16 weights of the layer / head kernels (swn_pack_bf16's image)
    size_t wx16;       // [L*128][A0x] bf16: in_x matrices, zero columns A0..A0x
    size_t xm16;       // [B][Tx][A0x] bf16: masked, upsampled conditioning, time-major
    size_t gx16;       // [B][Tp][L*128] bf16: sample-rate in_x products, bias included
    size_t total;
};
bool swn_bl6_drop_supported(const SwnGeom& g, int B, long Tp, int n_frames, const float* const* drop_h);   // csrc/swn_stack_bf16.hip
SwnBl6DropLayout swn_bl6_drop_layout(const SwnGeom& g, int B, long Tp);

// number of state_dict tensors in reference order (shallow_wavenet_amd/config.py param_shapes)
static inline __host__ int swn_tensor_count(const SwnGeom* g) {
    return 2 + 2 * g->auxl + 2 + (g->conv2d ? 2 : 0) + (g->wav ? 2 : 0) + 2 + 6 * g->L + 4;
}

// Arithmetic mode of the training call in progress ON THIS THREAD: every training entry point takes it as an argument
// (SWN_PRECISION_FP32 | SWN_PRECISION_BF16) and holds it in a SwnModeScope for the duration of the call, so the launch
// helpers below it need no extra parameter and nothing outlives the call (the C ABI has no process-wide state).
int  swn_call_mode();                 // csrc/swn_train.hip
void swn_call_mode_set(int mode);
struct SwnModeScope {
    int prev;
    explicit SwnModeScope(int mode) : prev(swn_call_mode()) { swn_call_mode_set(mode); }
    ~SwnModeScope() { swn_call_mode_set(prev); }
    SwnModeScope(const SwnModeScope&) = delete;
    SwnModeScope& operator=(const SwnModeScope&) = delete;
};
static inline bool swn_precision_ok(int mode) { return mode == SWN_PRECISION_FP32 || mode == SWN_PRECISION_BF16; }

// thread-local text of the last HIP failure seen by an entry point (swn_last_error_detail())
extern "C" void swn_set_error_detail(const char* where, const char* what);
#ifdef HIP_INCLUDE_HIP_HIP_RUNTIME_H
static inline int swn_launch_status(const char* where) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return SWN_OK;
    swn_set_error_detail(where, hipGetErrorString(e));
    return SWN_E_LAUNCH;
}
#endif
