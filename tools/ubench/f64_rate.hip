// What does a binary64 vector instruction cost on MI355X?  (Round 5: a timing-only build of the EXACT fused kernel without
// ANY of its LDS plane traffic runs as fast as the real one - the kernel is bound by VALU issue - yet with the datasheet's
// 4 cycles per wave64 v_fma_f64 its pipes would be only ~55 % busy.)
// 1024-thread workgroups, one per CU (4 waves per SIMD, as in the fused kernels); W of the four waves of each SIMD run 8
// independent chains of one instruction kind for a fixed wall time and count what they issued; the shader clock is read in
// the kernel.  Printed: SIMD-cycles per wave-instruction for v_fma_f64 / v_add_f64 / v_mul_f64 / v_fma_f32 / v_cndmask and
// for the butterfly's binary64 mix, at 1, 2 and 4 issuing waves per SIMD.
// build+run on the GPU box:  hipcc -O2 --offload-arch=gfx950 -o /tmp/f64_rate tools/ubench/f64_rate.hip && /tmp/f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ __launch_bounds__(1024) void probe(double* out, unsigned long long* rec, int active, unsigned long long ticks) {
    const int w = threadIdx.x >> 6;            // wave w sits on SIMD w % 4
    const bool on = (w >> 2) < active;
    double a0 = threadIdx.x + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double b = 1.0000001, c = 0.5;
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7, fb = 1.0001f, fc = 0.5f;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(fb), "+v"(fc));
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long n = 0;
    if (on) {
        while (__builtin_amdgcn_s_memrealtime() - r0 < ticks) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (KIND == 0) {
#define OP(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
                    OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
                } else if (KIND == 1) {
#define OP(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(c));
                    OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
                } else if (KIND == 2) {
#define OP(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(b));
                    OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
                } else if (KIND == 3) {
#define OP(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(fb), "v"(fc));
                    OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7)
#undef OP
                } else if (KIND == 4) {
#define OP(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(fb));
                    OP(f0) OP(f1) OP(f2) OP(f3) OP(f4) OP(f5) OP(f6) OP(f7)
#undef OP
                } else {   // the radix-2 butterfly's mix on 4 chains: 2 add, 2 sub, 2 mul, 2 fma
#define OP(x, y) asm volatile("v_add_f64 %0, %0, %1\n\tv_add_f64 %1, %1, -%0\n\tv_mul_f64 %0, %0, %2\n\tv_fma_f64 %1, %1, %2, %0" : "+v"(x), "+v"(y) : "v"(b));
                    OP(a0, a1) OP(a2, a3) OP(a4, a5) OP(a6, a7) OP(a0, a1) OP(a2, a3) OP(a4, a5) OP(a6, a7)
#undef OP
                }
            }
            n += (KIND == 5) ? 8 * 8 * 4 : 64;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* r = rec + 4 * (blockIdx.x * 16 + w);
        r[0] = t1 - t0; r[1] = r1 - r0; r[2] = n; r[3] = on;
    }
}

template <int KIND>
static void run(const char* name, double* out, unsigned long long* rec, int groups) {
    for (int active : {1, 2, 4}) {
        (void)0; hipLaunchKernelGGL(probe<KIND>, dim3(groups), dim3(1024), 0, 0, out, rec, active, 200000ull);   // 2 ms at 100 MHz
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)groups * 16 * 4);
        (void)hipMemcpy(h.data(), rec, h.size() * 8, hipMemcpyDeviceToHost);
        double cyc = 0, ins = 0, ticks = 0;
        for (int g = 0; g < groups; ++g)
            for (int w = 0; w < 16; ++w) {
                const unsigned long long* r = &h[4 * ((size_t)g * 16 + w)];
                if (r[3]) { ins += (double)r[2]; }
                if (w < 4) { cyc += (double)r[0]; ticks += (double)r[1]; }
            }
        // SIMD-cycles available = cycles of one wave per SIMD summed over SIMDs (4 per group)
        printf("%-28s %d issuing wave(s) per SIMD: %.2f SIMD-cycles per wave-instruction (clock %.2f GHz)\n", name, active,
               cyc / ins, cyc / ticks / 10.0);
    }
}

int main() {
    const int groups = 256;
    double* out; unsigned long long* rec;
    (void)hipMalloc(&out, (size_t)groups * 1024 * 8);
    (void)hipMalloc(&rec, (size_t)groups * 16 * 4 * 8);
    run<0>("v_fma_f64", out, rec, groups);
    run<1>("v_add_f64", out, rec, groups);
    run<2>("v_mul_f64", out, rec, groups);
    run<3>("v_fma_f32", out, rec, groups);
    run<4>("v_cndmask_b32", out, rec, groups);
    run<5>("f64 butterfly mix", out, rec, groups);
    return 0;
}
