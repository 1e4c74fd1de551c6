/*
 * swn_hip.h  --  C ABI of the MI355X (gfx950) shallow-WaveNet hot path.
 *
 * The reference (patrickltobing/shallow-wavenet) has no native code and no FFI: its hot
 * path is the PyTorch module code of src/nets/{cswnv_shift1,dswnv}.py.  This header is the
 * boundary a maintainer would bind from those modules (see INTEGRATION.md for the ctypes
 * stub); every entry point cites the reference lines it replaces.
 *
 * Conventions
 *   - plain C: pointers, sizes, a POD descriptor; no torch / C++ types.
 *   - `*_dev` pointers are device (HBM) addresses, `*_host` pointers host addresses.
 *   - all tensors fp32, contiguous, channels-first (B, C, T) exactly as the reference
 *     holds them; integer data (mu-law indices) are int32 on the device.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream). Launches are
 *     asynchronous; nothing in here synchronises, allocates or frees device memory.
 *   - every function returns 0 on success or a negative SWN_E_* code;
 *     swn_strerror() maps it to text.  No global mutable state (the arithmetic mode of the training
 *     entry points is an argument of each call, SWN_PRECISION_*); nothing reads the environment.
 */
#ifndef SWN_HIP_H
#define SWN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWN_ABI_VERSION 3   /* 3: `precision` argument instead of a process-wide switch; swn_decode_io.rng_utt_ids_dev;
                             *    decode variants 4 / 5 (cohort, cluster) retired */

/* arithmetic of the training contractions, an argument of every entry point it applies to
 *   FP32: fp32 operands on the matrix cores (v_mfma_f32_16x16x4_f32), bit-compatible with an fmaf chain - the parity
 *         mode every gradient fixture is checked in;
 *   BF16: mixed precision - the same fp32 tensors in HBM, operands rounded to bf16 on their way into LDS,
 *         v_mfma_f32_16x16x32_bf16 with fp32 accumulation (what torch.autocast(bfloat16) would do to the reference's
 *         conv forward / backward); gradients agree with FP32 to ~1e-2 relative per tensor.
 * A forward and the backward that reads its work buffer must be given the same value (the BF16 dropout forward keeps
 * gate pre-activations in the work buffer that the FP32 one does not). */
#define SWN_PRECISION_FP32 0
#define SWN_PRECISION_BF16 1

#define SWN_KIND_LAPLACE 0   /* CSWNV, cswnv_shift1.py:130 */
#define SWN_KIND_SOFTMAX 1   /* DSWNV, dswnv.py:190       */

#define SWN_OK            0
#define SWN_E_BADDESC    -1  /* descriptor outside what the kernels support            */
#define SWN_E_BADARG     -2  /* null pointer / size mismatch                           */
#define SWN_E_LAUNCH     -3  /* HIP reported a launch error (hipGetLastError)          */
#define SWN_E_UNSUPPORTED -4 /* valid reference configuration not built yet            */
#define SWN_E_NODEVICE   -5

/* Constructor arguments of CSWNV / DSWNV (cswnv_shift1.py:131-133, dswnv.py:191-193). */
typedef struct swn_net_desc {
    int32_t kind;              /* SWN_KIND_*                                  */
    int32_t n_aux;
    int32_t hid_chn;
    int32_t skip_chn;
    int32_t aux_kernel_size;
    int32_t aux_dilation_size;
    int32_t dilation_depth;
    int32_t dilation_repeat;
    int32_t kernel_size;
    int32_t upsampling_factor;
    int32_t seg;               /* laplace only, 1 for softmax                 */
    int32_t lpc;               /* laplace only                                */
    int32_t n_quantize;        /* softmax only                                */
    int32_t wav_conv_flag;
    int32_t audio_in_flag;     /* softmax only                                */
    int32_t aux_conv2d_flag;   /* laplace only; the (seg,1) Conv2d is folded into in_x at pack time */
} swn_net_desc;

/* ---- introspection --------------------------------------------------------------- */
int         swn_abi_version(void);
const char* swn_strerror(int code);
/* text of the last HIP runtime failure seen by an entry point on this thread ("" if none) */
const char* swn_last_error_detail(void);
/* number of HIP devices visible, or a negative SWN_E_* */
int         swn_device_count(void);
/* receptive field / number of state_dict tensors, as CSWNV.__init__ computes them
 * (cswnv_shift1.py:170-183).  Negative on a bad descriptor. */
int         swn_receptive_field(const swn_net_desc* d);
int         swn_num_tensors(const swn_net_desc* d);

/* ---- parameter packing (host side) ---------------------------------------------------
 * Re-lays the reference state_dict (tensors given in state_dict order, fp32 host pointers,
 * reference shapes: SURVEY.md 8b) into one flat fp32 buffer the kernels stream from:
 * tap-major dilated-conv rows, fused wav_conv+causal taps, the skip 1x1s concatenated, the
 * in_x 1x1s stacked for the frame-rate GEMM, summed biases.  The packed buffer is what is
 * uploaded once and broadcast over RCCL (decode_cswnv_laplace-shift1.py:223-224 loads the
 * checkpoint per process instead). */
size_t swn_packed_floats(const swn_net_desc* d);
/* float offsets of the packed sections in the order scale_w, scale_b, aux_w[0..3], aux_b[0..3], wx, wxa, wup,
 * bup, bx, cb, cv, cc, ct, wd, bd, wsk, bsk, w1, b1, w2, b2, total, bxr (csrc/swn_geom.hpp::SwnLayout); returns
 * the number written (29) or a negative SWN_E_*.  Used to unfold swn_backward's packed gradients. */
int    swn_layout_offsets(const swn_net_desc* d, size_t* out, int n);
int    swn_pack_params(const swn_net_desc* d, const float* const* tensors_host, int n_tensors,
                       float* packed_host, size_t packed_floats);
/* The same re-layout on the device, for parameters that already live in HBM (a training step ends with
 * optimizer.step(), train_cswnv_laplace-stftcmplx_shift1.py:872-874, after which every packed section is stale):
 *   tensors_dev  HOST array of n_tensors DEVICE pointers (state_dict order, fp32, contiguous, reference shapes)
 *   packed_dev   swn_packed_floats() floats, overwritten (padding zeroed); bit-identical to swn_pack_params      */
int    swn_pack_params_device(const swn_net_desc* d, const float* const* tensors_dev, int n_tensors,
                              float* packed_dev, size_t packed_floats, void* stream);
/* The way back for a training step: gradients in the packed layout (swn_backward*) -> the gradient of every parameter
 * tensor in the reference's shapes, i.e. the chain rule through the pack-time folds, in one launch
 * (csrc/swn_unfold_dev.hip; the torch-op version is nets/_autograd.py unfold_packed_grads).
 *   tensors_dev  HOST array of n_tensors DEVICE pointers: the live parameters (state_dict order)
 *   grads_dev    HOST array of n_tensors DEVICE pointers: contiguous fp32 outputs of the same shapes, overwritten;
 *                a NULL entry skips that tensor (requires_grad = False)
 * SWN_E_UNSUPPORTED with aux_conv2d_flag and seg > 1 (callers keep the torch path there). */
int    swn_unfold_grads_device(const swn_net_desc* d, const float* gpacked_dev, const float* const* tensors_dev,
                               float* const* grads_dev, int n_tensors, void* stream);

/* ---- frame-rate front end  (cswnv_shift1.py:193,297 / dswnv.py:252,302) ---------------
 * scale_in -> conv_aux (two-sided dilated k=3 stack) -> hoisted in_x:
 *   cond[b][f][l][s][o] = sum_c in_x[l].weight[o, c*seg+s] * conv_aux(scale_in(aux))[b,c,f]
 * The rank-1 upsampling (ConvTranspose2d (1,U), cswnv_shift1.py:37-65) is folded into the
 * consumers as  in_x(x)[o,t] = bx[l][o] + sum_s w_up[(t+s)%U] * cond[b][(t+s)/U][l][s][o].
 *   aux_dev   (B, n_aux, Tf)             in
 *   work_dev  swn_frontend_work_floats() scratch
 *   cond_dev  (B, Tf, L*seg*2H)          out; NULL = stop after conv_aux (the dropout mode, swn_forward_drop, applies
 *                                        in_x at sample rate and reads only the activations kept in work_dev) */
size_t swn_frontend_work_floats(const swn_net_desc* d, int batch, int n_frames);
size_t swn_cond_floats(const swn_net_desc* d, int batch, int n_frames);
int    swn_frontend(const swn_net_desc* d, const float* packed_dev, const float* aux_dev,
                    int batch, int n_frames, float* work_dev, float* cond_dev, void* stream);

/* ---- autoregressive decode  (CSWNV.batch_fast_generate cswnv_shift1.py:287-430,
 *                              DSWNV.batch_fast_generate dswnv.py:296-399) ---------------
 * One persistent workgroup per utterance runs prologue (rf+1 seed positions) and all
 * n_steps steps; all utterances run n_steps = max(n_samples)/seg steps like the reference.
 *   cond_dev    (B, Tf, L*seg*2H)  from swn_frontend
 *   io          inputs of the sampling loop, see swn_decode_io below
 *   state_dev   swn_decode_state_floats() scratch (history rings; zeroed by the call)
 *   out_dev     laplace: (B, n_steps*seg) fp32 ; softmax: (B, n_steps) int32
 *   heads_dev   optional (B, n_steps, n_out) raw out_2 outputs at each step (may be NULL)
 *   variant     0 = auto, 1 = generic persistent kernel, 2 = register/LDS-resident BL6-class kernels (the wave-specialised
 *                   form for the single-sample Laplace nets, the symmetric form for the others), 6 = the symmetric BL6-class
 *                   kernel whatever the net (A/B and parity runs),
 *               3 = stepped multi-launch decode for large geometries (REF6: what auto picks there)
 *                   (from 24 utterances on in tiles of 8 channel pairs x 8 utterances that fetch a pair's weight rows once per
 *                   tile and stage the utterances' activations in LDS).
 *               Variants 4 and 5 of ABI 2 (cohort / cluster experiments) are retired: SWN_E_BADARG.                       */
typedef struct swn_decode_io {
    /* sampling noise.  noise_dev != NULL: the host-drawn stream (parity mode; the host draws it with the torch CPU
     * generator in the reference's order): laplace (B, n_steps, seg) uniform(-0.4999, 0.5) draws
     * (cswnv_shift1.py:373,380,387), softmax (B, n_steps, Q) Exp(1) draws (the multinomial of dswnv.py:364-365).
     * noise_dev == NULL: the kernels draw the same quantities themselves with a counter-based generator
     * (Philox4x32-10 keyed by rng_seed, counter = (global utterance index of b, step, element): csrc/swn_noise.hpp), like the
     * reference drawing on the model's device; nothing is drawn, stored or uploaded by the host. */
    const float* noise_dev;
    /* optional teacher forcing (may be NULL): laplace (B, n_steps*seg) fp32 samples, softmax (B, n_steps) int32
     * indices fed back instead of the generated ones */
    const void*  forced_dev;
    /* optional seed waveform `audio` of batch_fast_generate (may be NULL = the decode drivers' seed: zeros /
     * mu-law class Q/2, decode_cswnv_laplace-shift1.py:93, decode_dswnv_softmax.py:94-99):
     * laplace (B, seg) fp32 samples, softmax (B) int32 classes (cswnv_shift1.py:300-334, dswnv.py:305-336) */
    const void*  seed_dev;
    /* optional (may be NULL): every noise value used is also written here, layout of noise_dev - lets a test replay
     * a device-drawn run in the CPU oracle ("given the same noise") */
    float*       noise_out_dev;
    uint64_t     rng_seed;      /* used when noise_dev == NULL */
    uint32_t     rng_utt0;      /* global index of utterance 0 (utterance b draws as rng_utt0 + b) ... */
    uint32_t     reserved;      /* 0 */
    /* ... or, when not NULL, (B) uint32 global utterance indices, one per utterance of the batch: a decode driver that
     * sorts utterances by length into batches (decode_cswnv_laplace-shift1.py:77-84) passes each utterance's position in
     * the unsorted list, so that the draws depend neither on batching nor on how the list is sharded over GPUs */
    const uint32_t* rng_utt_ids_dev;
} swn_decode_io;
size_t swn_decode_state_floats(const swn_net_desc* d, int batch);
int    swn_decode(const swn_net_desc* d, const float* packed_dev, const float* cond_dev,
                  int batch, int n_frames, int n_steps, const swn_decode_io* io,
                  float* state_dev, void* out_dev, float* heads_dev,
                  int variant, void* stream);

/* ---- teacher-forced stack  (CSWNV.forward cswnv_shift1.py:191-267,
 *                             DSWNV.forward dswnv.py:250-276) ----------------------------
 *   audio_dev   laplace: (B, 1, T - seg) fp32 samples ; softmax: (B, T - 1) int32 indices
 *               (the one-hot of dswnv.py:68-93 is never materialised)
 *   cond_dev    (B, Tf, L*seg*2H) from swn_frontend, T = Tf * U
 *   work_dev    swn_forward_work_floats() scratch (hidden states, skip accumulator)
 *   out_dev     (B, n_out, Tp) raw out_2 outputs, Tp = T - 2*seg + 1 (softmax: T - 1);
 *               the host splits mu / log b / a (cswnv_shift1.py:228-267)
 *   hs_dev      optional (B, L+1, H, Tp) hidden states h_0..h_L for backward / tests      */
size_t swn_forward_work_floats(const swn_net_desc* d, int batch, int n_frames);
int    swn_forward(const swn_net_desc* d, const float* packed_dev, const float* cond_dev,
                   const void* audio_dev, int batch, int n_frames, float* work_dev,
                   float* out_dev, float* hs_dev, void* stream);

/* ---- bf16 MFMA teacher-forced stack -------------------------------------------------------------
 * Training-speed variant of swn_forward: bf16 weights (device copy made by swn_pack_bf16) and bf16
 * time-major hidden states, fp32 accumulation, fp32 gate math.  Two geometry classes:
 *   BL6 class (H=64, K=2, S=128, Laplace head): register-resident layer kernels (HBM-bound);
 *   H a multiple of 64 up to 256, any K, Laplace or softmax (the reference's run.sh sizes): tiled GEMM stack.
 * Other geometries return SWN_E_UNSUPPORTED (use swn_forward).
 *   wbf16_dev  swn_bf16_weight_bytes() bytes, filled once per parameter set by swn_pack_bf16
 *   audio_dev  as for swn_forward: float waveform (Laplace) or int32 class indices (softmax)
 *   work_dev   swn_forward_bf16_work_bytes() bytes: hidden states [L+1][B][Tp][H] bf16 (+ skip / out_1 activations)
 *   out_dev    (B, n_out, Tp) fp32 raw out_2 outputs, same meaning as swn_forward            */
size_t swn_bf16_weight_bytes(const swn_net_desc* d);
int    swn_pack_bf16(const swn_net_desc* d, const float* packed_dev, void* wbf16_dev, void* stream);
size_t swn_forward_bf16_work_bytes(const swn_net_desc* d, int batch, int n_frames);
int    swn_forward_bf16(const swn_net_desc* d, const float* packed_dev, const void* wbf16_dev,
                        const float* cond_dev, const void* audio_dev, int batch, int n_frames,
                        void* work_dev, float* out_dev, void* stream);

/* mixed-precision training: expand what swn_forward_bf16 kept in work_dev (bf16, time-major) into the fp32 work layout
 * of swn_forward (hidden states | relu(skip) | relu(out_1); swn_forward_work_floats() floats), so that swn_backward can
 * follow a bf16 forward of the same (cond, audio).  GEMM-stack class: all three are expanded from memory.  BL6 class:
 * the head kernel keeps the two activations on chip, so the hidden states are expanded and the two 1x1 products are
 * redone from them in the arithmetic `precision` selects (packed_dev is read only there).  _supported: 1 where swn_forward_bf16 exists, else 0. */
int    swn_bf16_train_forward_supported(const swn_net_desc* d);
int    swn_bf16_work_to_f32(const swn_net_desc* d, const float* packed_dev, const void* work_bf16_dev, int batch,
                            int n_frames, float* fwd_work_dev, int precision, void* stream);

/* ---- Laplace output split  (cswnv_shift1.py:228-267) -----------------------------------
 * raw (B, n_out, Tp) from swn_forward  ->  time-major tensors the reference returns:
 *   mu (B,Tp,seg) ; logb = logsigmoid(.) (B,Tp,seg) ; b = exp(logb) ; a (B,Tp,lpc) (NULL if lpc==0)
 *   b_clip / logb_clip (optional, may be NULL): logb floored at -14.1621 (b >= 7.07e-7), :233-236
 *   below_floor: int32[1] set non-zero when any logb < floor (the reference's torch.min test);
 *   the caller zeroes it before the call.                                                    */
int    swn_laplace_head(const swn_net_desc* d, const float* out_dev, int batch, int tp,
                        float* mu_dev, float* b_dev, float* logb_dev, float* a_dev,
                        float* b_clip_dev, float* logb_clip_dev, int32_t* below_floor_dev,
                        void* stream);

/* ---- backward of the teacher-forced stack (fp32)  (loss.backward() through CSWNV/DSWNV.forward,
 *      train_cswnv_laplace-stftcmplx_shift1.py:724-874) ----------------------------------------------
 *   aux_dev, cond_dev, audio_dev   the forward inputs
 *   fe_work_dev    the work buffer swn_frontend filled (scaled features and conv_aux activations)
 *   fwd_work_dev   the work buffer swn_forward filled (hidden states, relu(skip), relu(out_1))
 *   hs_dev         the hidden states if swn_forward wrote them to a separate hs_dev, else NULL
 *   grad_out_dev   (B, n_out, Tp) gradient of the loss wrt the raw out_2 outputs
 *   work_dev       swn_backward_work_floats() scratch
 *   gpacked_dev    swn_packed_floats() floats, ZEROED AND FILLED by the call: gradients in the packed
 *                  parameter layout (csrc/swn_geom.hpp); the host unfolds them onto the parameters  */
size_t swn_backward_work_floats(const swn_net_desc* d, int batch, int n_frames);
int    swn_backward(const swn_net_desc* d, const float* packed_dev, const float* aux_dev, const float* cond_dev,
                    const float* fe_work_dev, const void* audio_dev, const float* fwd_work_dev,
                    const float* hs_dev, const float* grad_out_dev, int batch, int n_frames,
                    float* work_dev, float* gpacked_dev, int precision, void* stream);
/* The same backward after a bf16 forward of the BL6 class, in the mixed-precision arithmetic (SWN_PRECISION_BF16) only, with
 * everything at sample rate fused (csrc/swn_bwd_bl6.hip): one launch for the head (recompute of relu(skip) / relu(out_1),
 * d out_1, d skip, g out_2), one per gated layer (the layer above's data gradient, the skip path's share, recompute, gate
 * derivative, highway carry, conditioning) plus one that also does the input layer, and one for every other weight
 * gradient - all reading the bf16 time-major hidden states directly.
 *   work_bf16_dev  the work buffer swn_forward_bf16 filled
 *   fwd_work_dev   ignored (may be NULL): the fp32 expansion of swn_bf16_work_to_f32 is not needed on this path
 *   work_dev       swn_backward_bf16_work_floats() floats (0 = geometry / size not covered: seg == 1, 16 <= U <= 112,
 *                  Laplace, BL6 stack, S = 128; the call then returns SWN_E_UNSUPPORTED and the caller uses swn_backward) */
size_t swn_backward_bf16_work_floats(const swn_net_desc* d, int batch, int n_frames);
int    swn_backward_bf16(const swn_net_desc* d, const float* packed_dev, const float* aux_dev, const float* cond_dev,
                         const float* fe_work_dev, const void* audio_dev, const float* fwd_work_dev,
                         const void* work_bf16_dev, const float* grad_out_dev, int batch, int n_frames,
                         float* work_dev, float* gpacked_dev, void* stream);
/* Mixed-precision training at the GEMM-stack geometries (hid_chn % 64 == 0 outside the BL6 class): the bf16 forward that also
 * keeps every layer's gate pre-activations (fp32, swn_forward_bf16_keep_floats() floats; 0 = variant not applicable), and the
 * backward that reads them instead of recomputing each layer's dilated conv (1.7 of 13 ms per step at the run.sh geometry).
 * swn_backward_keep takes swn_backward's arguments with a_keep_dev in place of hs_dev; fwd_work_dev is the fp32 expansion
 * swn_bf16_work_to_f32 made of the same forward's work buffer. */
size_t swn_forward_bf16_keep_floats(const swn_net_desc* d, int batch, int n_frames);
int    swn_forward_bf16_keep(const swn_net_desc* d, const float* packed_dev, const void* wbf16_dev,
                             const float* cond_dev, const void* audio_dev, int batch, int n_frames,
                             void* work_dev, float* out_dev, float* a_keep_dev, void* stream);
int    swn_backward_keep(const swn_net_desc* d, const float* packed_dev, const float* aux_dev, const float* cond_dev,
                         const float* fe_work_dev, const void* audio_dev, const float* fwd_work_dev,
                         const float* a_keep_dev, const float* grad_out_dev, int batch, int n_frames,
                         float* work_dev, float* gpacked_dev, void* stream);
/* ---- training-mode forward / backward WITH DROPOUT  (model.train(), forward(..., do=True) with do_prob > 0:
 *      cswnv_shift1.py:194-195,211-217,269-273 ; dswnv.py:253-254,264-270,278-282) --------------------------
 * The reference draws its Bernoulli masks inside nn.Dropout; here they are explicit inputs, like the decode
 * noise, so that the host can draw them with the torch CPU generator in the reference's order:
 *   drop_x_dev   (B, A0, T - coff) multiplicative mask (0 or 1/(1-p)) on the upsampled conditioning
 *                (A0 = n_aux * aux_kernel^aux_layers, coff = seg | 1 for softmax)
 *   drop_h_host  HOST array of L device pointers: drop_h[l] = (B, H, Tp) mask on the hidden state that layer l
 *                hands to layer l+1 (its skip output is not masked), NULL where the reference does not drop
 * aux_drop acts at sample rate, so the frame-rate hoisting of in_x does not apply: in_x is evaluated as a
 * sample-rate GEMM on the masked conditioning (no cond_dev input; fe_work_dev = swn_frontend's work buffer).
 * fwd_work_dev of swn_backward_drop must be the buffer swn_forward_drop filled, and `precision` must be the same for
 * both calls: with SWN_PRECISION_BF16, for nets with hid_chn % 64 == 0, the forward runs its sample-rate in_x GEMM and, per
 * layer, the dilated conv as a bf16-operand GEMM followed by an element-wise gate kernel (instead of the fused exact-fp32
 * layer kernel) and keeps every layer's gate pre-activations in that buffer; the backward reads them instead of
 * recomputing them. */
/* BL6 class (swn_backward_bf16's geometries) with SWN_PRECISION_BF16 and no mask between layers (drop_h_host[l] == NULL for
 * l < L-1; with dilation_repeat == 1 the reference's only hidden-state mask lands on the last layer's output, which nothing
 * reads - cswnv_shift1.py:211-217): both calls take a fused path instead - masked conditioning and in_x products as bf16
 * time-major rows from one tiled GEMM, the bf16 layer kernels of swn_forward_bf16 reading those rows, and the fused per-layer
 * backward of swn_backward_bf16 handing back their gradients for the two in_x contractions.  swn_drop_fused_path() tells
 * which path a (batch, n_frames, drop_h_host) takes in that mode: 1 fused, 0 the generic chain. */
int    swn_drop_fused_path(const swn_net_desc* d, int batch, int n_frames, const float* const* drop_h_host);
size_t swn_forward_drop_work_floats(const swn_net_desc* d, int batch, int n_frames);
int    swn_forward_drop(const swn_net_desc* d, const float* packed_dev, const float* fe_work_dev, const void* audio_dev,
                        int batch, int n_frames, const float* drop_x_dev, const float* const* drop_h_host,
                        float* work_dev, float* out_dev, float* hs_dev, int precision, void* stream);
size_t swn_backward_drop_work_floats(const swn_net_desc* d, int batch, int n_frames);
int    swn_backward_drop(const swn_net_desc* d, const float* packed_dev, const float* aux_dev, const float* fe_work_dev,
                         const void* audio_dev, const float* fwd_work_dev, const float* hs_dev,
                         const float* drop_x_dev, const float* const* drop_h_host, const float* grad_out_dev,
                         int batch, int n_frames, float* work_dev, float* gpacked_dev, int precision, void* stream);
/* gradient of swn_laplace_head: grads wrt mu / b / logb / a (time-major, any may be NULL) -> grad wrt raw */
int    swn_laplace_head_backward(const swn_net_desc* d, const float* out_dev, int batch, int tp,
                                 const float* gmu_dev, const float* gb_dev, const float* glogb_dev,
                                 const float* ga_dev, const float* gb_clip_dev, const float* glogb_clip_dev,
                                 float* graw_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SWN_HIP_H */
