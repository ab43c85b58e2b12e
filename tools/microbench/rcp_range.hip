// rcp_range.hip -- per binary exponent of x: how many floats have  v_rcp_f32 + one FMA correction  !=  1.0f/x  (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void check(unsigned long long *bad) {       // bad[256], indexed by the exponent field
    for (unsigned long long k = (unsigned long long)blockIdx.x * 256u + threadIdx.x; k < (1ull << 32); k += (unsigned long long)gridDim.x * 256u) {
        const float x = __uint_as_float((unsigned)k);
        const float r = __builtin_amdgcn_rcpf(x);
        const float q = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
        const float e = 1.0f / x;
        const bool both_nan = (q != q) && (e != e);
        if (__float_as_uint(q) != __float_as_uint(e) && !both_nan) atomicAdd(&bad[((unsigned)k >> 23) & 255u], 1ull);
    }
}
int main() {
    unsigned long long *d, h[256];
    hipMalloc(&d, sizeof h); hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(check, dim3(8192), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int e = 0; e < 256; e++) if (h[e]) printf("exponent field %3d (|x| ~ 2^%d): %llu mismatches (of 2 x 8388608)\n", e, e - 127, h[e]);
    printf("done\n");
    return 0;
}
