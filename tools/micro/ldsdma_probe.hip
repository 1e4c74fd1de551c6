// dev probe (not part of the library): semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 as bf16_layer_ring_kernel
// uses it - lane i's 16 bytes at M0 + 16 i, LDS bases above 64 KB, out-of-range offsets, exec-masked lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_piece(const v4i rsrc, unsigned voff, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc), "s"(lds_base) : "memory");
}
__global__ __launch_bounds__(64) void probe(const unsigned* src, unsigned nbytes, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x;
    for (int e = lane; e < 40960; e += 64) reinterpret_cast<unsigned*>(lds)[e] = 0xdeadbeefu;   // 160 KB
    __syncthreads();
    const unsigned long long a = reinterpret_cast<unsigned long long>(src);
    v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)(a & 0xffffffffull));
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffull));
    r.z = __builtin_amdgcn_readfirstlane((int)nbytes);
    r.w = 0x00020000;
    // piece 0: permuted lanes -> base 0 ; piece 1: base 100 KB, lanes >= 48 out of range ; piece 2: base 150 KB, only lanes < 32
    dma_piece(r, (unsigned)((lane * 7) & 63) * 16u, 0u);
    dma_piece(r, lane < 48 ? 1024u + lane * 16u : 0x80000000u, 100u * 1024u);
    if (lane < 32) dma_piece(r, 2048u + lane * 16u, 150u * 1024u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < 3; ++k) {
        const unsigned base = k == 0 ? 0u : (k == 1 ? 100u * 1024u : 150u * 1024u);
        for (int q = 0; q < 4; ++q) out[(k * 64 + lane) * 4 + q] = reinterpret_cast<unsigned*>(lds + base)[lane * 4 + q];
    }
}
int main() {
    const unsigned n = 4096;                                     // dwords
    std::vector<unsigned> h(n);
    for (unsigned i = 0; i < n; ++i) h[i] = i;
    unsigned *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, 3 * 64 * 16);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 160 * 1024, 0, d, n * 4, o);
    std::vector<unsigned> r(3 * 64 * 4);
    hipError_t e = hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
    printf("status %s\n", hipGetErrorString(e));
    int bad0 = 0, bad1 = 0, bad2 = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int q = 0; q < 4; ++q) {
            const unsigned want0 = ((lane * 7) & 63) * 4 + q;
            if (r[(0 * 64 + lane) * 4 + q] != want0) ++bad0;
            const unsigned want1 = lane < 48 ? 256 + lane * 4 + q : 0u;
            if (r[(1 * 64 + lane) * 4 + q] != want1) ++bad1;
            const unsigned want2 = lane < 32 ? 512 + lane * 4 + q : 0xdeadbeefu;
            if (r[(2 * 64 + lane) * 4 + q] != want2) ++bad2;
        }
    printf("piece0 (permuted lanes, base 0): %d bad ; piece1 (base 100 KB, OOB lanes -> 0): %d bad, lane 50 holds %08x ; "
           "piece2 (base 150 KB, exec-masked): %d bad, lane 40 holds %08x\n", bad0, bad1, r[(64 + 50) * 4], bad2, r[(128 + 40) * 4]);
    return 0;
}
