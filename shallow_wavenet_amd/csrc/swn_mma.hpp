// fp32 tile product on the matrix cores, shared by the fp32 parity / training kernels
// (tf_layer, gemm_wx, time_gemm, reduce_gemm).
//
// v_mfma_f32_16x16x4_f32 multiplies fp32 operands exactly like an fmaf chain (no reduced-precision inputs), so the
// results stay within the fp32 parity tolerances, while one instruction does 1024 MACs: the 64x64x16 tile step
// that cost 1024 v_fma + 128 LDS reads per thread group now costs 16 MFMAs + 20 LDS reads per wave.
//
// Geometry: a workgroup of 4 waves owns a 64 x 64 output tile; the k-tile lives in LDS as As[16][SWN_MMA_PITCH]
// (k-major: As[k][row]) and Bs[16][SWN_MMA_PITCH] (Bs[k][col]).  Wave w owns columns 16w..16w+15 of all 64 rows as
// four 16x16 accumulators:   acc[mt][i] = C[16*mt + 4*(lane/16) + i][16*w + lane%16]
// Pitch 80 floats: the four k-rows a wave reads at once (k0 + lane/16) fall into four disjoint groups of 16 banks.
#pragma once
#include <hip/hip_runtime.h>

#define SWN_MMA_PITCH 80
typedef float swn_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void swn_mma_64x64x16(const float (*As)[SWN_MMA_PITCH], const float (*Bs)[SWN_MMA_PITCH],
                                                 swn_f32x4 (&acc)[4], const int lane, const int w) {
    const int kq = lane >> 4, rc = lane & 15;
#pragma unroll
    for (int k0 = 0; k0 < 16; k0 += 4) {
        const float b = Bs[k0 + kq][16 * w + rc];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(As[k0 + kq][16 * mt + rc], b, acc[mt], 0, 0, 0);
    }
}
// row / column of accumulator element (mt, i) inside the 64x64 tile
__device__ __forceinline__ int swn_mma_row(int lane, int mt, int i) { return 16 * mt + 4 * (lane >> 4) + i; }
__device__ __forceinline__ int swn_mma_col(int lane, int w) { return 16 * w + (lane & 15); }

// ---- bf16-operand variant (mixed-precision training mode, precision = SWN_PRECISION_BF16): the same 64 x 64 tile and
// accumulator layout, k-tile of 32 with v_mfma_f32_16x16x32_bf16 (16x the rate of the exact fp32 instruction), fp32
// accumulation.  LDS holds the operands row-major in k as packed bf16 pairs: As[row][k/2], Bs[col][k/2], 80-byte rows
// (conflict-free 16-byte fragment reads).
#define SWN_MMB_PITCH 20
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 swn_bf16x8;

__device__ __forceinline__ unsigned swn_pack_bf16(float lo, float hi) {
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
}
__device__ __forceinline__ void swn_mmb_64x64x32(const unsigned (*As)[SWN_MMB_PITCH], const unsigned (*Bs)[SWN_MMB_PITCH],
                                                 swn_f32x4 (&acc)[4], const int lane, const int w) {
    const int kq = lane >> 4, rc = lane & 15;
    const swn_bf16x8 b = *reinterpret_cast<const swn_bf16x8*>(&Bs[16 * w + rc][4 * kq]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const swn_bf16x8 a = *reinterpret_cast<const swn_bf16x8*>(&As[16 * mt + rc][4 * kq]);
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[mt], 0, 0, 0);
    }
}

// Operand loads of the tiled kernels: through buffer resources with 32-bit byte offsets.  Whatever must read as
// zero (k past the end, a row past M, a position outside [0, T)) is the out-of-range offset, so the loads need
// neither branches nor selects - either would make the compiler drain the prefetch queue (s_waitcnt vmcnt(0)) at
// every k-tile, which is what these kernels were bound by.  Every operand (per batch item) must be < 2 GiB.
constexpr unsigned SWN_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float bld1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
typedef float swn_fl4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ swn_fl4 bld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(swn_fl4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

