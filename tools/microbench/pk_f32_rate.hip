// GPU box: issue rate of packed fp32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) against the scalar forms on gfx950.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/pk_f32_rate.hip -o /tmp/pk_rate && /tmp/pk_rate
// Each kernel runs ITER x 16 independent instructions per wave; 8 waves per SIMD so that latency is hidden and the issue port is the limit.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float pk2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;
template <int KIND> __global__ __launch_bounds__(512) void rate(float *out, float seed) {
    float a[16]; pk2 p[16];
    for (int k = 0; k < 16; k++) { a[k] = seed + k + threadIdx.x; p[k] = pk2{ a[k], a[k] + 0.5f }; }
    const float m = 1.0000001f; const pk2 mm = pk2{ m, m };
    for (int i = 0; i < ITER; i++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(mm));
            if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(mm));
            if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(m));
            if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[k]) : "v"(mm));
        }
    }
    float s = 0; for (int k = 0; k < 16; k++) s += a[k] + p[k].x + p[k].y;
    if (s == 12345.678f) out[0] = s;
}
template <int KIND> void run(const char *name, float *d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 4;              // 256 CUs x 4 workgroups of 8 waves = 8 waves per SIMD
    rate<KIND><<<blocks, 512>>>(d, 1.0f);
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) rate<KIND><<<blocks, 512>>>(d, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double wave_instr = (double)blocks * 8 * ITER * 16;
    // per SIMD: 1024 SIMDs; cycles at 2.4 GHz
    printf("%-14s %8.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, ms, ms * 1e-3 * 2.4e9 / (wave_instr / 1024.0));
}
int main() {
    float *d; hipMalloc(&d, 4);
    run<0>("v_mul_f32", d); run<1>("v_pk_mul_f32", d); run<2>("v_add_f32", d); run<3>("v_pk_add_f32", d); run<4>("v_fma_f32", d); run<5>("v_pk_fma_f32", d);
    return 0;
}
