// launch_sync.hip — what one launch + one wait costs on this stack, three ways (the floor under the live calls' per-call time):
//   (a) empty kernel + hipStreamSynchronize            (b) the same kernel, a 13 us busy one
//   (c) the kernel raises a flag in page-locked host memory behind a system-scope fence and the host spins on it
// build: hipcc -O2 --offload-arch=gfx950 -o /tmp/launch_sync tools/ubench/launch_sync.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <vector>

__global__ void work_kernel(long long ticks, volatile unsigned* flag, unsigned value, unsigned* done, unsigned blocks) {
    const long long t0 = wall_clock64();
    while (ticks > 0 && wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (flag) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence_system();
            if (atomicAdd(done, 1u) + 1u == blocks) {
                *done = 0;
                __threadfence_system();
                *flag = value;
            }
        }
    }
}

static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    unsigned* flag = nullptr;
    hipHostMalloc((void**)&flag, 64, hipHostMallocDefault);
    unsigned* done = nullptr;
    hipMalloc((void**)&done, 4);
    hipMemset(done, 0, 4);
    *flag = 0;
    const int reps = 2000, blocks = 128;
    for (int mode = 0; mode < 4; ++mode) {
        const long long ticks = (mode == 0) ? 0 : 1300;   // 100 MHz: 1300 ticks = 13 us
        const bool spin = mode >= 2;
        std::vector<double> t;
        for (int i = 0; i < reps; ++i) {
            const auto a = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(work_kernel, dim3(blocks), dim3(256), 0, st, ticks, spin ? flag : nullptr, (unsigned)(i + 1), done, (unsigned)blocks);
            if (spin) {
                while (*(volatile unsigned*)flag != (unsigned)(i + 1)) {}
                if (mode == 3) hipStreamSynchronize(st);       // (what a conservative caller adds)
            } else {
                hipStreamSynchronize(st);
            }
            const auto b = std::chrono::steady_clock::now();
            if (i >= reps / 10) t.push_back(std::chrono::duration<double, std::micro>(b - a).count());
        }
        const char* name[] = {"empty kernel + hipStreamSynchronize", "13 us kernel + hipStreamSynchronize", "13 us kernel + host spin on a pinned flag",
                              "13 us kernel + flag spin + hipStreamSynchronize"};
        printf("%-52s median %.1f us\n", name[mode], med(t));
        hipStreamSynchronize(st);
    }
    return 0;
}
