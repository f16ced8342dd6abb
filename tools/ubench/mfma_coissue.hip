// Does the matrix pipe co-issue beside this repo's VALU mix?  ("first a 1-hour ubench".)
// 1024-thread workgroups, one per CU (4 waves per SIMD, as in the column kernels).  Of the four waves of a SIMD, M run a
// chain of independent v_mfma_f32_16x16x4_f32 (8 accumulators, back to back), the others the butterfly's VALU mix
// (mul, fma, add, sub on 8 independent chains).  Every wave runs for the same wall time (s_memrealtime) and counts what
// it issued; the shader clock is read in the kernel.  Printed per M: VALU wave-instructions per SIMD-cycle, MFMAs per
// SIMD-cycle, and the VALU rate relative to M = 0 scaled by the waves left (1.0 = the MFMA waves cost the others nothing).
// build+run on the GPU box:  hipcc -O2 --offload-arch=gfx950 -o /tmp/mfma_coissue tools/ubench/mfma_coissue.hip && /tmp/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void probe(float* out, unsigned long long* rec, int mfma_waves, unsigned long long ticks, int young) {
    const int w = threadIdx.x >> 6;            // wave w sits on SIMD w % 4; waves 0..3 are the first (oldest) of their SIMDs
    // young = 0: the OLDEST mfma_waves waves of every SIMD run the matrix chain (issue arbitration is oldest-first);
    // young = 1: the YOUNGEST do, and the VALU waves raise their priority (s_setprio 3) on top
    const bool mm = young ? (w >> 2) >= 4 - mfma_waves : (w >> 2) < mfma_waves;
    if (young && !mm) __builtin_amdgcn_s_setprio(3);
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = 1.0001f, c = 0.5f;
    asm volatile("" : "+v"(b), "+v"(c));
    f4 c0 = {a0, a1, a2, a3}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long n = 0;
    if (mm) {
        while (__builtin_amdgcn_s_memrealtime() - r0 < ticks) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, c3, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, c4, 0, 0, 0);
                c5 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, c5, 0, 0, 0);
                c6 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, c6, 0, 0, 0);
                c7 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, c7, 0, 0, 0);
            }
            n += 64;
        }
    } else {
        while (__builtin_amdgcn_s_memrealtime() - r0 < ticks) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#define OP(x) asm volatile("v_mul_f32 %0, %0, %1\n\tv_fma_f32 %0, %0, %1, %2\n\tv_add_f32 %0, %0, %2\n\tv_sub_f32 %0, %0, %1" : "+v"(x) : "v"(b), "v"(c));
                OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
            }
            n += 8 * 8 * 4;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c0.x + c1.y + c2.z + c3.w + c4.x + c5.y + c6.z + c7.w;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* r = rec + 4 * (blockIdx.x * 16 + w);
        r[0] = t1 - t0; r[1] = r1 - r0; r[2] = n; r[3] = mm;
    }
}

int main() {
    const int groups = 256;
    float* out; unsigned long long* rec;
    hipMalloc(&out, sizeof(float) * groups * 1024);
    hipMalloc(&rec, sizeof(unsigned long long) * groups * 16 * 4);
    double base = 0;
    for (int young : {0, 1})
    for (int m : {0, 0, 1, 2, 3, 4}) {
        if (young && (m == 0 || m == 4)) continue;
        hipLaunchKernelGGL(probe, dim3(groups), dim3(1024), 0, 0, out, rec, m, 1000000ull, young);   // 10 ms at 100 MHz
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(groups * 16 * 4);
        hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
        double shader = 0, real = 0, nv = 0, nm = 0, cyc_simd = 0;
        for (int g = 0; g < groups * 16; ++g) {
            shader += h[4 * g]; real += h[4 * g + 1];
            (h[4 * g + 3] ? nm : nv) += (double)h[4 * g + 2];
        }
        cyc_simd = shader / (groups * 16) * (groups * 4);      // mean wave duration in shader cycles x number of SIMDs
        const double ghz = shader / real * 0.1;
        const double valu = nv / cyc_simd, mfma = nm / cyc_simd;
        if (m == 0) base = valu;
        printf("%s M = %d MFMA waves of 4 per SIMD: clock %.3f GHz; VALU %.4f wave-instructions per SIMD-cycle (%.2f cycles each)",
               young ? "[MFMA waves youngest, VALU waves at s_setprio 3]" : "[MFMA waves oldest]", m, ghz, valu, valu > 0 ? 1.0 / valu : 0.0);
        if (m > 0) printf("; MFMA %.4f per SIMD-cycle (%.1f cycles each)", mfma, mfma > 0 ? 1.0 / mfma : 0.0);
        if (m > 0 && m < 4) printf("; VALU rate / (M = 0 rate) = %.3f", valu / base);
        printf("\n");
    }
    hipFree(out); hipFree(rec);
    return 0;
}
