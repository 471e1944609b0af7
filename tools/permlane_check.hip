// Checks the permlane-swap forms of the cross-row lane exchanges against __shfl_xor on the device.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../image-captioning_amd/csrc/decode_shared.h"
void mic_set_error(const char*, ...) {}
__global__ void k(const float* in, float* out) {
    const int l = threadIdx.x;
    const float v = in[l];
    out[l] = lane_xor<16>(v);
    out[64 + l] = __shfl_xor(v, 16, 64);
    out[128 + l] = lane_xor<32>(v);
    out[192 + l] = __shfl_xor(v, 32, 64);
    out[256 + l] = sum_xor16(v);
    out[320 + l] = v + __shfl_xor(v, 16, 64);
    out[384 + l] = sum_xor32(v);
    out[448 + l] = v + __shfl_xor(v, 32, 64);
}
int main() {
    float h[64], o[512], *di, *dout;
    for (int i = 0; i < 64; ++i) h[i] = 1.0f + i * 1.000123f;
    hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 64; ++i)
            if (o[t * 128 + i] != o[t * 128 + 64 + i]) { if (bad < 8) printf("test %d lane %d: %g vs %g\n", t, i, o[t * 128 + i], o[t * 128 + 64 + i]); ++bad; }
    printf("mismatches: %d\n", bad);
    return bad != 0;
}
