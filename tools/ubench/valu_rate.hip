// VALU issue-rate probe for gfx950: cycles per wave-instruction per SIMD for v_fma_f32, v_pk_fma_f32,
// v_add_f32 and v_mul_f32 at 1, 2, 4 waves per SIMD (one 64*4*W-thread workgroup per CU).
// build+run on the GPU box:  hipcc -O2 --offload-arch=gfx950 -o /tmp/valu_rate tools/ubench/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void probe(float* out, unsigned long long* cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float b = 1.0001f, c = 0.5f;
    const f2 pb = {b, b}, pc = {c, c};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {
#define OP(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
                OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
            } else if (KIND == 1) {
#define OP(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(pb), "v"(pc));
                OP(p0) OP(p1) OP(p2) OP(p3) OP(p4) OP(p5) OP(p6) OP(p7)
#undef OP
            } else if (KIND == 2) {
#define OP(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));
                OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
            } else if (KIND == 3) {
#define OP(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(b));
                OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#undef OP
            } else {
#define OP(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(pb));
                OP(p0) OP(p1) OP(p2) OP(p3) OP(p4) OP(p5) OP(p6) OP(p7)
#undef OP
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x +
                                                 p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void run(const char* name) {
    const int iters = 2000, groups = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * groups * 1024);
    hipMalloc(&cyc, sizeof(unsigned long long) * groups);
    for (int wps : {1, 2, 4}) {
        const int threads = 64 * 4 * wps;
        hipLaunchKernelGGL(probe<KIND>, dim3(groups), dim3(threads), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<KIND>, dim3(groups), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(groups);
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * groups, hipMemcpyDeviceToHost);
        double mean = 0; for (auto v : h) mean += v; mean /= groups;
        const double insts_per_wave = (double)iters * 64;
        // readcyclecounter ticks at a constant 100 MHz on this part; use the event time and the nominal 2.4 GHz clock too
        printf("%-14s %d waves/SIMD: %.3f ms for %.0f instr/wave -> %.2f SIMD-cycles per wave-instruction at 2.4 GHz (counter ticks %.0f)\n",
               name, wps, ms, insts_per_wave, ms * 1e-3 * 2.4e9 / (insts_per_wave * wps), mean);
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32");
    run<1>("v_pk_fma_f32");
    run<2>("v_add_f32");
    run<3>("v_mul_f32");
    run<4>("v_pk_add_f32");
    return 0;
}
