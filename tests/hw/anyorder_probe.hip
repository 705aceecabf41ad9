// anyorder_probe.hip -- does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) clear the AQL barrier bit on gfx950, i.e. do
// kernels queued on ONE stream overlap instead of waiting for each other's last workgroup?  (hip_ext.h says the flag is "not
// supported on GFX9xx" for the module-launch entry point; this asks the hardware.)
//
//   A  8 spin kernels (1 workgroup each, ~1 ms) on one stream, ordinary launches      expected ~8 ms
//   B  the same with hipExtAnyOrderLaunch                                              ~1 ms if the flag is honoured
//   C  an any-order kernel queued behind hipStreamWaitEvent(ev) where ev follows a 5 ms spin on another stream: does it
//      wait for the event?  (it reads a flag the long kernel writes last)
//   D  ordinary kernel, then any-order kernels, then hipEventRecord: does the event wait for all of them?
// Build:  hipcc --offload-arch=gfx950 -O2 tests/hw/anyorder_probe.hip -o tests/hw/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_spin(unsigned long long ticks, int* flag_out, const int* flag_in, int* seen) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    if (flag_in && threadIdx.x == 0) *seen = *(volatile const int*)flag_in;   // what was visible when I started
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (flag_out && threadIdx.x == 0) { *(volatile int*)flag_out = 1; __threadfence_system(); }
}

static float run(hipStream_t st, int n, unsigned long long ticks, unsigned flags) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < n; ++i)
        hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, nullptr, nullptr, flags, ticks, (int*)nullptr, (const int*)nullptr, (int*)nullptr);
    CK(hipGetLastError());
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms;
}

int main() {
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const unsigned long long ms1 = 100000;                                    // 1 ms of the 100 MHz counter
    run(s1, 2, ms1, 0);                                                       // warm
    const float A = run(s1, 8, ms1, 0);
    const float B = run(s1, 8, ms1, hipExtAnyOrderLaunch);
    printf("A ordered   8 x 1 ms on one stream: %.2f ms\n", A);
    printf("B any-order 8 x 1 ms on one stream: %.2f ms  -> flag %s\n", B, B < 0.5f * A ? "HONOURED (kernels overlap)" : "IGNORED");
    // C
    int *flag, *seen; CK(hipMalloc(&flag, 4)); CK(hipMalloc(&seen, 4)); CK(hipMemset(flag, 0, 4)); CK(hipMemset(seen, 0xff, 4));
    CK(hipDeviceSynchronize());
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s2, 5 * ms1, flag, (const int*)nullptr, (int*)nullptr);
    CK(hipEventRecord(ev, s2));
    CK(hipStreamWaitEvent(s1, ev, 0));
    hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, ms1 / 10, (int*)nullptr, (const int*)flag, seen);
    CK(hipDeviceSynchronize());
    int h = -2; CK(hipMemcpy(&h, seen, 4, hipMemcpyDeviceToHost));
    printf("C any-order kernel behind hipStreamWaitEvent saw flag = %d  -> %s\n", h, h == 1 ? "waited for the event" : "DID NOT WAIT");
    // D
    CK(hipMemset(flag, 0, 4)); CK(hipDeviceSynchronize());
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, s1));
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, ms1 / 10, (int*)nullptr, (const int*)nullptr, (int*)nullptr);
    for (int i = 0; i < 4; ++i)
        hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, (i == 0 ? 3 : 1) * ms1, i == 0 ? flag : (int*)nullptr, (const int*)nullptr, (int*)nullptr);
    CK(hipEventRecord(b, s1));
    CK(hipEventSynchronize(b));
    CK(hipMemcpy(&h, flag, 4, hipMemcpyDeviceToHost));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    printf("D event after ordered + 4 any-order (3,1,1,1 ms): %.2f ms, long kernel's flag %d -> %s\n", ms, h, h == 1 ? "event waited for all" : "EVENT FIRED EARLY");
    // E: an ordinary kernel behind any-order ones waits for them?
    CK(hipMemset(flag, 0, 4)); CK(hipMemset(seen, 0xff, 4)); CK(hipDeviceSynchronize());
    hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, 2 * ms1, flag, (const int*)nullptr, (int*)nullptr);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, ms1 / 10, (int*)nullptr, (const int*)flag, seen);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&h, seen, 4, hipMemcpyDeviceToHost));
    printf("E ordinary kernel behind an any-order one saw flag = %d -> %s\n", h, h == 1 ? "waited" : "DID NOT WAIT");
    return 0;
}
