// rcp_variants.hip -- which short v_rcp_f32 + FMA sequences equal the correctly rounded 1.0f/x for EVERY float in
// 2^-100 <= |x| <= 2^100?  (exhaustive: all 2^32 bit patterns).  hipcc --offload-arch=gfx950 -O3 -fhip-fp32-correctly-rounded-divide-sqrt
#include <hip/hip_runtime.h>
#include <cstdio>

template <int V> __device__ float cand(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    if (V == 0) {            // the shipped one: Newton step + two residual corrections (6 fma)
        float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r);
        float q = r; float rem = __builtin_fmaf(-x, q, 1.0f); q = __builtin_fmaf(rem, r, q);
        rem = __builtin_fmaf(-x, q, 1.0f); return __builtin_fmaf(rem, r, q);
    }
    if (V == 1) {            // Newton step + one residual correction (4 fma)
        float e = __builtin_fmaf(-x, r, 1.0f); r = __builtin_fmaf(e, r, r);
        float rem = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(rem, r, r);
    }
    if (V == 2) {            // one residual correction only (2 fma)
        float rem = __builtin_fmaf(-x, r, 1.0f); return __builtin_fmaf(rem, r, r);
    }
    if (V == 3) {            // two residual corrections with the raw estimate as the multiplier (4 fma)
        float q = r; float rem = __builtin_fmaf(-x, q, 1.0f); q = __builtin_fmaf(rem, r, q);
        rem = __builtin_fmaf(-x, q, 1.0f); return __builtin_fmaf(rem, r, q);
    }
    return r;
}

template <int V> __global__ void check(unsigned long long *bad) {
    unsigned long long b = 0;
    for (unsigned long long k = (unsigned long long)blockIdx.x * 256u + threadIdx.x; k < (1ull << 32); k += (unsigned long long)gridDim.x * 256u) {
        const float x = __uint_as_float((unsigned)k);
        const float ax = __builtin_fabsf(x);
        if (!(ax >= 0x1p-100f && ax <= 0x1p100f)) continue;
        if (__float_as_uint(cand<V>(x)) != __float_as_uint(1.0f / x)) b++;
    }
    if (b) atomicAdd(bad, b);
}

template <int V> void run(const char *name) {
    unsigned long long *d, h = 0;
    hipMalloc(&d, 8); hipMemset(d, 0, 8);
    hipLaunchKernelGGL(check<V>, dim3(8192), dim3(256), 0, 0, d);
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost); hipFree(d);
    printf("%-70s mismatches %llu\n", name, h);
}

int main() {
    run<0>("rcp + Newton + 2 corrections (6 fma, shipped)");
    run<1>("rcp + Newton + 1 correction (4 fma)");
    run<2>("rcp + 1 correction (2 fma)");
    run<3>("rcp + 2 corrections, raw estimate as multiplier (4 fma)");
    run<4>("rcp alone");
    return 0;
}
