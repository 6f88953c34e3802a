// Hardware probe: semantics of v_cvt_pk_u8_f32 (rounding, saturation, NaN) and of packed f32 math.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
__global__ void k(const float* in, unsigned* out, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, 0xAABBCCDDu);
}
int main() {
    float h[] = {0.5f, 1.5f, 2.5f, 0.49999997f, 0.50000006f, -1.0f, -0.4f, 300.0f, 254.5f, 255.5f, 255.49f, NAN, INFINITY, -INFINITY, 127.5f, 128.5f, 1e-40f, 3.5f};
    int n = sizeof h / sizeof h[0];
    float* d; unsigned* o; unsigned r[32];
    hipMalloc(&d, sizeof h); hipMalloc(&o, 32 * 4);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
    hipMemcpy(r, o, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) {
        float c = fminf(fmaxf(h[i], 0.0f), 255.0f);
        unsigned expect = (unsigned)rintf(c);
        printf("%14.8g -> %08x  byte1=%3u  rne_expect=%3u %s\n", h[i], r[i], (r[i] >> 8) & 0xff, expect, ((r[i] >> 8) & 0xff) == expect ? "ok" : "DIFF");
    }
    return 0;
}
