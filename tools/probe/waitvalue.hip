// Probe (development tool): does this box support stream-ordered waits on a host-written flag (hipStreamWaitValue64),
// and what does the round trip cost?  A kernel is queued behind the wait; the host writes the flag; we time flag -> kernel done.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void touch(volatile double *out, const double *late) { if (threadIdx.x == 0) out[0] = late[0] + 1.0; }
int main()
{
    int can = -1;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (!can) return 0;
    hipStream_t s; CK(hipStreamCreate(&s));
    uint64_t *flag; CK(hipHostMalloc((void **)&flag, 64, hipHostMallocDefault));
    double *late; CK(hipHostMalloc((void **)&late, 64, hipHostMallocDefault));
    double *out;  CK(hipHostMalloc((void **)&out, 64, hipHostMallocDefault));
    uint64_t *dflag; CK(hipHostGetDevicePointer((void **)&dflag, flag, 0));
    double *dlate, *dout; CK(hipHostGetDevicePointer((void **)&dlate, late, 0)); CK(hipHostGetDevicePointer((void **)&dout, out, 0));
    double total = 0; int n = 200;
    for (int k = 1; k <= n + 5; ++k) {
        *flag = 0; out[0] = 0; late[0] = k;
        CK(hipStreamWaitValue64(s, dflag, (uint64_t)k, hipStreamWaitValueGte, 0xffffffffffffffffull));
        hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, dout, dlate);
        // the queue is waiting; nothing must have run yet
        for (volatile int spin = 0; spin < 20000; ++spin) {}
        if (out[0] != 0) { printf("kernel ran before the flag!\n"); return 1; }
        auto t0 = std::chrono::steady_clock::now();
        __atomic_store_n(flag, (uint64_t)k, __ATOMIC_RELEASE);
        while (((volatile double *)out)[0] != k + 1.0) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) { printf("timeout\n"); return 1; }
        }
        auto t1 = std::chrono::steady_clock::now();
        if (k > 5) total += std::chrono::duration<double, std::micro>(t1 - t0).count();
        CK(hipStreamSynchronize(s));
    }
    printf("flag written -> kernel result visible on the host: %.2f us (mean of %d)\n", total / n, n);
    // for comparison: a plain launch of the same kernel on an idle stream, launch -> result visible
    total = 0;
    for (int k = 1; k <= n + 5; ++k) {
        out[0] = 0; late[0] = k;
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, s, dout, dlate);
        while (((volatile double *)out)[0] != k + 1.0) {}
        auto t1 = std::chrono::steady_clock::now();
        if (k > 5) total += std::chrono::duration<double, std::micro>(t1 - t0).count();
        CK(hipStreamSynchronize(s));
    }
    printf("plain launch -> kernel result visible on the host: %.2f us (mean of %d)\n", total / n, n);
    return 0;
}
