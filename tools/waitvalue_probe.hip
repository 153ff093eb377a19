// Feasibility probe: can a stream be held back by hipStreamWaitValue64 until a KERNEL on another stream stores to signal memory, and how
// long after the store does the held kernel start?   hipcc --offload-arch=gfx950 -O2 -o waitvalue_probe waitvalue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void setter(unsigned long long *sig, unsigned long long value, long long delay_ticks, long long *t)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(8);
    t[0] = wall_clock64();
    __hip_atomic_store(sig, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // keep running a little so that "kernel end" cannot be what releases the waiter
    while (wall_clock64() - t0 < delay_ticks + 5000) __builtin_amdgcn_s_sleep(8);
    t[2] = wall_clock64();
}
__global__ void held(long long *t) { t[1] = wall_clock64(); }
int main()
{
    unsigned long long *sig = nullptr;
    CK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
    CK(hipMemset(sig, 0, 8));
    long long *t; CK(hipMalloc(&t, 64)); CK(hipMemset(t, 0, 64));
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (int rep = 1; rep <= 4; ++rep) {
        CK(hipStreamWaitValue64(a, sig, (uint64_t)rep, hipStreamWaitValueGte, 0xffffffffffffffffull));
        hipLaunchKernelGGL(held, dim3(1), dim3(64), 0, a, t);
        hipLaunchKernelGGL(setter, dim3(1), dim3(64), 0, b, sig, (unsigned long long)rep, 5000LL /* 50 us */, t);
        CK(hipDeviceSynchronize());
        long long h[3]; CK(hipMemcpy(h, t, 24, hipMemcpyDeviceToHost));
        printf("rep %d: held kernel started %.2f us after the store (setter ran on for %.2f us after it)\n", rep, (h[1] - h[0]) * 0.01, (h[2] - h[0]) * 0.01);
    }
    // the host-side release: hipStreamWriteValue64 on another stream
    CK(hipStreamWaitValue64(a, sig, 100, hipStreamWaitValueGte, 0xffffffffffffffffull));
    hipLaunchKernelGGL(held, dim3(1), dim3(64), 0, a, t);
    CK(hipStreamWriteValue64(b, sig, 100, 0));
    CK(hipDeviceSynchronize());
    printf("write-value release ok\n");
    // device memory that is NOT signal memory
    unsigned long long *plain; CK(hipMalloc(&plain, 8)); CK(hipMemset(plain, 0, 8));
    hipError_t e = hipStreamWaitValue64(a, plain, 1, hipStreamWaitValueGte, 0xffffffffffffffffull);
    printf("wait on plain device memory: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(held, dim3(1), dim3(64), 0, a, t);
        hipLaunchKernelGGL(setter, dim3(1), dim3(64), 0, b, plain, 1ull, 5000LL, t);
        CK(hipDeviceSynchronize());
        long long h[3]; CK(hipMemcpy(h, t, 24, hipMemcpyDeviceToHost));
        printf("plain memory: held kernel started %.2f us after the store\n", (h[1] - h[0]) * 0.01);
    }
    return 0;
}
