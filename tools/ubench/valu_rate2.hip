// VALU issue-rate probe, second form (round 2): long kernels (>= 10 ms each, so the clock has settled), the shader
// clock read inside the kernel (s_memtime / s_memrealtime x 100 MHz), 8 independent chains per wave, 1 / 2 / 4 waves
// per SIMD, and instruction mixes closer to the column kernel's (float + integer + compare/select).
// build+run on the GPU box:  hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_rate2 tools/ubench/valu_rate2.hip && /tmp/valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
template <int KIND>
__global__ void probe(float* out, unsigned long long* cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    float b = 1.0001f, c = 0.5f;
    asm volatile("" : "+v"(b), "+v"(c));
    const f2 pb = {b, b}, pc = {c, c};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {
#define OP(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
                REP8(OP)
#undef OP
            } else if (KIND == 1) {
#define OP(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
                REP8(OP)
#undef OP
            } else if (KIND == 2) {
#define OP(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b));
                REP8(OP)
#undef OP
            } else if (KIND == 3) {
#define OP(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));
                REP8(OP)
#undef OP
            } else if (KIND == 4) {
#define OP(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(x) : "v"(b), "v"(c) : "vcc");
                REP8(OP)
#undef OP
            } else if (KIND == 5) {   // the butterfly's mix: mul, fma, add, sub
#define OP(x) asm volatile("v_mul_f32 %0, %0, %1\n\tv_fma_f32 %0, %0, %1, %2\n\tv_add_f32 %0, %0, %2\n\tv_sub_f32 %0, %0, %1" : "+v"(x) : "v"(b), "v"(c));
                REP8(OP)
#undef OP
            } else if (KIND == 6) {
#define OP(x) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
                REP8(OP)
#undef OP
            } else if (KIND == 7) {   // constant operand from an SGPR / inline constant instead of a third VGPR
#define OP(x) asm volatile("v_fma_f32 %0, %0, 0.5, 1.0" : "+v"(x));
                REP8(OP)
#undef OP
            } else if (KIND == 8) {
#define OP(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(pb), "v"(pc));
                OP(p0) OP(p1) OP(p2) OP(p3) OP(p4) OP(p5) OP(p6) OP(p7)
#undef OP
            } else {
#define OP(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(pb));
                OP(p0) OP(p1) OP(p2) OP(p3) OP(p4) OP(p5) OP(p6) OP(p7)
#undef OP
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x +
                                                 p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cyc[2 * w] = t1 - t0;
        cyc[2 * w + 1] = r1 - r0;
    }
}

template <int KIND>
static void run(const char* name, int per_op) {
    const int groups = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * groups * 1024);
    hipMalloc(&cyc, sizeof(unsigned long long) * groups * 16 * 2);
    for (int wps : {1, 2, 4}) {
        const int threads = 64 * 4 * wps;
        const int iters = 120000 / per_op / wps * (wps == 1 ? 1 : 1);
        hipLaunchKernelGGL(probe<KIND>, dim3(groups), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<KIND>, dim3(groups), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const int nw = groups * 4 * wps;
        std::vector<unsigned long long> h(2 * nw);
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * 2 * nw, hipMemcpyDeviceToHost);
        double shader = 0, real = 0, longest = 0;
        for (int w = 0; w < nw; ++w) { shader += h[2 * w]; real += h[2 * w + 1]; if ((double)h[2 * w] > longest) longest = (double)h[2 * w]; }
        const double ghz = shader / real * 0.1;              // s_memrealtime: 100 MHz
        const double insts_per_wave = (double)iters * 64 * per_op;
        // every SIMD holds wps waves for the whole kernel: SIMD cycles = the longest wave's shader cycles
        printf("%-34s %d waves/SIMD: %7.3f ms, clock %.3f GHz, %.3f SIMD-cycles per wave-instruction (slowest wave), %.3f (kernel time x clock)\n",
               name, wps, ms, ghz, longest / (insts_per_wave * wps), ms * 1e-3 * ghz * 1e9 / (insts_per_wave * wps));
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32 (3 VGPR sources)", 1);
    run<1>("v_add_f32", 1);
    run<2>("v_add_u32", 1);
    run<3>("v_and_b32", 1);
    run<4>("v_cmp_lt_f32 + v_cndmask_b32", 2);
    run<5>("mul, fma, add, sub (butterfly mix)", 4);
    run<6>("v_fma_f32 x, x, x, x", 1);
    run<7>("v_fma_f32 x, x, 0.5, 1.0", 1);
    run<8>("v_pk_fma_f32", 1);
    run<9>("v_pk_add_f32", 1);
    return 0;
}
