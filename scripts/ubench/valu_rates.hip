// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD at 1..4 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N_ITER 2048
#define UNROLL 8
template <int OP> __global__ void k(double *out, double seed, int n) {
    double a[UNROLL];
    unsigned u[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + threadIdx.x * 1e-3 + i; u[i] = threadIdx.x + i; }
    double b = seed * 1.0000001, c = seed * 0.999;
    unsigned long long mask = __builtin_amdgcn_ballot_w64(threadIdx.x & 1);
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 3) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 4) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]));
            if (OP == 6) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]), "s"(mask));
            if (OP == 16) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %3, vcc" : "+v"(u[i]) : "v"(a[i]), "v"(b), "v"(u[(i + 1) % UNROLL]) : "vcc");
            if (OP == 17) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) % UNROLL]));
            if (OP == 18) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]));
            if (OP == 19) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(u[i]));
            if (OP == 7) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[i]));
            if (OP == 8) asm volatile("v_rsq_f64 %0, %0" : "+v"(a[i]));
            if (OP == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]));
            if (OP == 10) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]), "v"(u[(i + 2) % UNROLL]));
            if (OP == 11) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(b) : "vcc");
            if (OP == 12) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 13) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[i]) : "v"(u[i]));
            if (OP == 14) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
            if (OP == 15) asm volatile("v_rndne_f64 %0, %0" : "+v"(a[i]));
            // 32-bit float forms (round 3: the conservative single-precision box filter of the node loop)
            if (OP == 20) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 21) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 22) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 23) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]), "v"(u[(i + 2) % UNROLL]));
            if (OP == 24) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]), "v"(u[(i + 2) % UNROLL]));
            if (OP == 25) asm volatile("v_max_f32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]));
            if (OP == 26) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(u[i]), "v"(u[(i + 1) % UNROLL]) : "vcc");
            if (OP == 27) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(u[i]) : "v"(a[i]));
            if (OP == 28) asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]), "v"(u[(i + 2) % UNROLL]));
            if (OP == 29) asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(u[i]), "v"(u[(i + 1) % UNROLL]) : "vcc");
            if (OP == 30) asm volatile("v_cmp_eq_u16 vcc, 0, %0" : : "v"(u[i]) : "vcc");
            if (OP == 31) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 32) asm volatile("v_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %3, %4, %2, %3" : "+v"(u[i]), "+v"(u[(i + 1) % UNROLL]) : "v"(u[(i + 2) % UNROLL]), "v"(u[(i + 3) % UNROLL]), "v"(u[(i + 4) % UNROLL]));
            if (OP == 33) asm volatile("v_add_f32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) % UNROLL]));
        }
    }
    double s = 0; unsigned t = 0;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { s += a[i]; t += u[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + t;
}
template <int OP> double run(int wavesPerSimd, double *d) {
    const int cus = 256, block = 256 * wavesPerSimd; // 4 SIMDs per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(block), 0, 0, d, 1.5, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(block), 0, 0, d, 1.5, N_ITER);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / ((double) N_ITER * UNROLL * wavesPerSimd); // seconds per wave-instruction per SIMD
}
int main() {
    double *d; hipMalloc(&d, 256 * 1024 * sizeof(double));
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("clock %d kHz\n", clk);
    const char *names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_max_f64", "v_cmp_lt_f64", "v_add_u32", "v_cndmask_b32", "v_rcp_f64", "v_rsq_f64", "v_mul_lo_u32", "v_fma_f32", "v_div_scale_f64", "v_div_fixup_f64", "v_ldexp_f64", "v_cvt_f64_u32", "v_rndne_f64", "cmp_f64+cndmask(vcc)", "v_mov_b32", "v_xor_b32", "v_lshlrev_b32",
                           "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_max3_f32", "v_min3_f32", "v_max_f32", "v_cmp_ge_f32", "v_cvt_f32_f64", "v_alignbit_b32", "v_cmp_gt_i32", "v_cmp_eq_u16", "v_pk_fma_f32(op_sel_hi)", "2xv_fma_f32(pair)", "v_add_f32"};
    for (int w = 1; w <= 4; w *= 2) {
        double r[34] = {run<0>(w, d), run<1>(w, d), run<2>(w, d), run<3>(w, d), run<4>(w, d), run<5>(w, d), run<6>(w, d), run<7>(w, d), run<8>(w, d), run<9>(w, d), run<10>(w, d), run<11>(w, d), run<12>(w, d), run<13>(w, d), run<14>(w, d), run<15>(w, d), run<16>(w, d), run<17>(w, d), run<18>(w, d), run<19>(w, d),
                        run<20>(w, d), run<21>(w, d), run<22>(w, d), run<23>(w, d), run<24>(w, d), run<25>(w, d), run<26>(w, d), run<27>(w, d), run<28>(w, d), run<29>(w, d), run<30>(w, d), run<31>(w, d), run<32>(w, d), run<33>(w, d)};
        printf("waves/SIMD %d:", w);
        for (int i = 0; i < 34; ++i) printf(" %s=%.2f", names[i], r[i] * clk * 1e3);
        printf("  (cycles per wave64 instruction per SIMD at the reported clock; the 2x pair row is per PAIR)\n");
    }
    return 0;
}
