// Sampling noise of the autoregressive decode: either the host-drawn stream (parity mode: the torch CPU generator in
// the reference's draw order, cswnv_shift1.py:373,380,387 / dswnv.py:364-365) or a counter-based generator evaluated
// inside the kernels, so that nothing has to be drawn, stored (B x n_steps x 256 floats for the softmax model) or
// uploaded by the host.
//
// Generator: Philox4x32-10 (Salmon et al., SC'11), key = the 64-bit seed, counter = (global utterance index, step,
// word group, stream tag); the index is rng_utt0 + b or the caller's per-utterance list (swn_decode_io.rng_utt_ids_dev).  Every draw is a pure function of (seed, utterance, step, element): results do not depend on
// batch composition, on the decode kernel variant, or on how utterances are sharded over GPUs.
//   Laplace: e = -0.4999 + 0.9999 * u24,  u24 = (bits >> 8) * 2^-24 in [0, 1)      (uniform_(-0.4999, 0.5))
//   softmax: q = -log(u),                 u   = ((bits >> 9) + 0.5) * 2^-23 in (0, 1)   (Exp(1) of multinomial's n=1 path)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct SwnNoise {
    const float* ptr;          // host-drawn stream (B, n_steps, width) or nullptr -> generate in the kernel
    float* dump;               // optional: every value used is also written here, same layout (tests replay it in the oracle)
    uint32_t key0, key1;       // 64-bit seed
    uint32_t utt0;             // global index of utterance 0 of this launch (utterance b = utt0 + b) ...
    const uint32_t* ids;       // ... or, when not null, the global index of every utterance of the launch (B values)
};

__device__ __forceinline__ uint32_t swn_utt_id(const SwnNoise& n, uint32_t utt) { return n.ids ? n.ids[utt] : n.utt0 + utt; }

__device__ __forceinline__ uint4 swn_philox4x32_10(uint4 c, uint2 k) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0; k.y += W1;
    }
    return c;
}

__device__ __forceinline__ uint32_t swn_rng_word(const SwnNoise& n, uint32_t utt, uint32_t step, uint32_t elem, uint32_t tag) {
    const uint4 r = swn_philox4x32_10(make_uint4(swn_utt_id(n, utt), step, elem >> 2, tag), make_uint2(n.key0, n.key1));
    const uint32_t s = elem & 3u;
    return s == 0 ? r.x : (s == 1 ? r.y : (s == 2 ? r.z : r.w));
}

// uniform(-0.4999, 0.5) draw j of generation step `step` of utterance b (width = seg draws per step)
__device__ __forceinline__ float swn_noise_laplace(const SwnNoise& n, int b, int step, int j, int n_steps, int seg) {
    const size_t at = ((size_t)b * n_steps + step) * seg + j;
    float e;
    if (n.ptr) {
        e = n.ptr[at];
    } else {
        const float u = (float)(swn_rng_word(n, (uint32_t)b, (uint32_t)step, (uint32_t)j, 0x4C41504Cu) >> 8) * 5.9604644775390625e-8f;
        e = fminf(fmaf(0.9999f, u, -0.4999f), 0.49999997f);
    }
    if (n.dump) n.dump[at] = e;
    return e;
}

// Exp(1) draw of class `cls` at generation step `step` of utterance b
__device__ __forceinline__ float swn_noise_exp1(const SwnNoise& n, int b, int step, int cls, int n_steps, int Q) {
    const size_t at = ((size_t)b * n_steps + step) * Q + cls;
    float q;
    if (n.ptr) {
        q = n.ptr[at];
    } else {
        const float u = ((float)(swn_rng_word(n, (uint32_t)b, (uint32_t)step, (uint32_t)cls, 0x45585031u) >> 9) + 0.5f) * 1.1920928955078125e-7f;
        q = -logf(u);
    }
    if (n.dump) n.dump[at] = q;
    return q;
}

// the four Exp(1) draws of classes 4g .. 4g+3 (one generator call): same values as swn_noise_exp1 element by element.
// Q must be a multiple of 4.
__device__ __forceinline__ float4 swn_noise_exp1x4(const SwnNoise& n, int b, int step, int g, int n_steps, int Q) {
    const size_t at = ((size_t)b * n_steps + step) * Q + 4 * (size_t)g;
    float4 q;
    if (n.ptr) {
        q = *reinterpret_cast<const float4*>(n.ptr + at);
    } else {
        const uint4 r = swn_philox4x32_10(make_uint4(swn_utt_id(n, (uint32_t)b), (uint32_t)step, (uint32_t)g, 0x45585031u),
                                          make_uint2(n.key0, n.key1));
        q.x = -logf(((float)(r.x >> 9) + 0.5f) * 1.1920928955078125e-7f);
        q.y = -logf(((float)(r.y >> 9) + 0.5f) * 1.1920928955078125e-7f);
        q.z = -logf(((float)(r.z >> 9) + 0.5f) * 1.1920928955078125e-7f);
        q.w = -logf(((float)(r.w >> 9) + 0.5f) * 1.1920928955078125e-7f);
    }
    if (n.dump) *reinterpret_cast<float4*>(n.dump + at) = q;
    return q;
}
