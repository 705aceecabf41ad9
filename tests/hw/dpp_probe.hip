// Hardware probe: does DPP wave_shr:1 (full-wave shift by one lane) behave on gfx950 as the
// systolic alignment kernel assumes?  lane k must receive lane k-1's value, lane 0 keeps `old`.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
    int x = 1000 + (int)threadIdx.x;
    int y = __builtin_amdgcn_update_dpp(-7, x, 0x138, 0xf, 0xf, false);   // wave_shr:1
    int z = __builtin_amdgcn_update_dpp(-7, x, 0x111, 0xf, 0xf, false);   // row_shr:1
    out[threadIdx.x] = y;
    out[64 + threadIdx.x] = z;
}
int main() {
    int* d; int h[128];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess) { printf("no device\n"); return 2; }
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
    int bad = 0;
    for (int k = 0; k < 64; ++k) { int want = k == 0 ? -7 : 1000 + k - 1; if (h[k] != want) { ++bad; printf("wave_shr lane %d: got %d want %d\n", k, h[k], want); } }
    int badr = 0;
    for (int k = 0; k < 64; ++k) { int want = (k % 16) == 0 ? -7 : 1000 + k - 1; if (h[64 + k] != want) ++badr; }
    printf("wave_shr:1 %s (%d bad lanes); row_shr:1 %s\n", bad ? "BROKEN" : "OK", bad, badr ? "BROKEN" : "OK");
    return bad ? 1 : 0;
}
