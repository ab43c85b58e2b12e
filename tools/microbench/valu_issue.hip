// valu_issue.hip -- how many wave64 VALU instructions per cycle one gfx950 SIMD sustains, by instruction kind and by
// waves per SIMD.  Standalone: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o valu_issue_noslp valu_issue.hip && ./valu_issue_noslp
// (-fno-slp-vectorize as the library is built: a plain -O3 build packs the independent adds / multiplies / fmas of the first rows into
// v_pk_*_f32 -- 4 cycles for two operations per lane, the same rate for add and mul, twice the rate for fma; profiles/r02_valu_issue.txt
// holds both builds' output)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int kIters = 4096;

// Each kernel: 16 independent chains x 8 ops per loop trip = 128 VALU ops per trip.
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cycles, float seed, int n_iter) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    f2_t p[16];
    for (int i = 0; i < 16; i++) { p[i].x = seed + i + threadIdx.x * 1e-3f; p[i].y = seed * 0.5f + i; }
    const f2_t pc = { 1.0001f, 0.9999f };
    float a[16];
    unsigned u[16];
    for (int i = 0; i < 16; i++) { a[i] = seed + i + threadIdx.x * 1e-3f; u[i] = (unsigned)(seed * 977) + i * 7919u + threadIdx.x; }
    float s = seed * 1.0001f;             // wave-uniform -> SGPR
    unsigned um = 747796405u + 2u * threadIdx.x; unsigned long long m = 0, msk = 0x5555555555555555ull * (unsigned long long)(seed * 3);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n_iter; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (KIND == 0) a[i] = a[i] * 1.0001f;                                   // v_mul_f32
                if (KIND == 1) a[i] = a[i] + 1.0001f;                                   // v_add_f32
                if (KIND == 2) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f);              // v_fma_f32
                if (KIND == 3) a[i] = a[i] * s;                                         // v_mul_f32 with SGPR operand
                if (KIND == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);                      // v_rcp_f32
                if (KIND == 5) u[i] = u[i] * um + 3u;                                   // v_mul_lo_u32 + v_add (2 ops) (um is per-thread)
                if (KIND == 6) u[i] = (u[i] >> 3) ^ (u[i] + um);                        // shift + add + xor (3 ops)
                if (KIND == 7) a[i] = (a[i] < s) ? a[i] * 1.0001f : 1.5f;               // v_cmp + v_mul + v_cndmask (3 ops)
                if (KIND == 8) a[i] = __builtin_fminf(a[i] * 1.0001f, s);               // v_mul + v_min (2 ops)
                if (KIND == 9) u[i] = __umul24(u[i], um) + 3u;                          // v_mad_u32_u24 (um per-thread)
                if (KIND == 10) a[i] = __builtin_sqrtf(a[i] * 1.01f);                   // v_mul + IEEE sqrt expansion
                if (KIND == 11) a[i] = 1.0f / (a[i] * 1.01f);                           // v_mul + IEEE div expansion
                if (KIND == 12) a[i] = (float)u[i] * 0x1p-31f - a[i];                   // v_cvt_f32_u32 + v_mul + v_sub (3 ops) 
                if (KIND == 15) { if (i == 0) a[0] = __builtin_fmaf(a[0], 1.0001f, 0.5f); }                // ONE dependent chain
                if (KIND == 16) { if (i < 4) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f); }                  // four chains
                if (KIND == 17) { if (i < 2) a[i] = __builtin_fmaf(a[i], 1.0001f, 0.5f); }                  // two chains
                if (KIND == 20) p[i] = p[i] * pc;                                       // v_pk_mul_f32
                if (KIND == 21) p[i] = p[i] + pc;                                       // v_pk_add_f32
                if (KIND == 22) p[i] = __builtin_elementwise_fma(p[i], pc, pc);         // v_pk_fma_f32
                if (KIND == 23) { p[i].x = p[i].x * 1.0001f; p[i].y = p[i].y * 0.9999f; }   // two v_mul_f32 (same work as KIND 20)
                if (KIND == 13) { bool c = a[i] < s; m ^= __builtin_amdgcn_ballot_w64(c); }   // v_cmp to SGPR + s_xor
                if (KIND == 14) a[i] = (msk >> ((i + r) & 63) & 1) ? a[i] * 1.0001f : a[i];     // scalar-mask select
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0; unsigned ua = 0;
    for (int i = 0; i < 16; i++) { acc += a[i] + p[i].x + p[i].y; ua ^= u[i]; }
    out[blockIdx.x * 256 + threadIdx.x] = acc + (float)ua + (float)(m & 0xff);
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
int run(const char *name, int ops_per_elem) {
    int dev_cus = 0;
    CHECK(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0));
    float *out; unsigned long long *cyc;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * dev_cus * 8));
    CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * dev_cus * 8));
    printf("%-34s", name);
    for (int blocks_per_cu : { 1, 2, 4, 6, 8 }) {       // 256-thread blocks: waves per SIMD = blocks per CU
        int blocks = dev_cus * blocks_per_cu;
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.25f, 64);           // warm up
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.25f, kIters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks);
        CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
        double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;                         // s_memtime ticks (100 MHz const clock?)
        double wave_instr = (double)kIters * 128 * ops_per_elem;                                  // per wave
        double total = wave_instr * blocks * 4;                                                    // wave-instructions
        double per_simd_per_s = total / (dev_cus * 4.0) / (ms * 1e-3);
        printf("  w/SIMD %d: %6.3f Ginstr/s/SIMD", blocks_per_cu, per_simd_per_s / 1e9);
    }
    printf("\n");
    CHECK(hipFree(out)); CHECK(hipFree(cyc));
    return 0;
}

int main() {
    printf("wave64 VALU instruction rate per SIMD (Ginstr/s); at ~2.4 GHz, 1.2 = one instruction every 2 cycles\n");
    run<0>("v_mul_f32", 1); run<1>("v_add_f32", 1); run<2>("v_fma_f32", 1); run<3>("v_mul_f32 (SGPR operand)", 1);
    run<4>("v_rcp_f32", 1); run<5>("v_mul_lo_u32 + v_add", 2); run<6>("v_lshrrev + v_add + v_xor", 3); run<7>("v_cmp + v_mul + v_cndmask", 3);
    run<8>("v_mul + v_min", 2); run<9>("v_mad_u32_u24", 1); run<10>("v_mul + IEEE sqrtf (instr count?)", 1); run<11>("v_mul + IEEE 1/x", 1);
    run<12>("v_cvt_f32_u32 + v_mul + v_sub", 3); run<13>("v_cmp->SGPR (+s_xor)", 1);
    printf("dependent chains (rate counts only the chain ops: 1/16, 4/16, 2/16 of the slots):\n");
    run<20>("v_pk_mul_f32 (2 mul/lane)", 1); run<21>("v_pk_add_f32", 1); run<22>("v_pk_fma_f32", 1); run<23>("2 x v_mul_f32 (same work)", 2);
    run<15>("1 dependent v_fma chain (x16)", 1); run<17>("2 chains (x8)", 1); run<16>("4 chains (x4)", 1);
    return 0;
}
