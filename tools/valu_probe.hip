// valu_probe.hip -- issue rate of the VALU instructions the proposal kernel's sweep is made of, per SIMD, at 1..8 waves per
// SIMD (design input: is a wave64 VALU instruction 2 or 4 cycles of a gfx950 SIMD?).
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define REP8(x) x x x x x x x x

typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k_probe(unsigned* out, int iters, unsigned long long* cyc) {
    unsigned a0 = threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    f2 p0 = {(float)a0, 1.f}, p1 = {(float)a1, 2.f}, p2 = {(float)a2, 3.f}, p3 = {(float)a3, 4.f};
    f2 p4 = {(float)a4, 5.f}, p5 = {(float)a5, 6.f}, p6 = {(float)a6, 7.f}, p7 = {(float)a7, 8.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {   // v_min_u32 (plain 32-bit integer VALU), 8 independent chains x 8
            REP8(asm volatile("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n"
                              "v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));)
        } else if (KIND == 1) {   // v_med3_u32 (VOP3, three sources)
            REP8(asm volatile("v_med3_u32 %0, %0, %1, %8\n v_med3_u32 %1, %1, %2, %8\n v_med3_u32 %2, %2, %3, %8\n v_med3_u32 %3, %3, %4, %8\n"
                              "v_med3_u32 %4, %4, %5, %8\n v_med3_u32 %5, %5, %6, %8\n v_med3_u32 %6, %6, %7, %8\n v_med3_u32 %7, %7, %0, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));)
        } else if (KIND == 2) {   // v_fma_f32
            float* q = reinterpret_cast<float*>(&p0);
            (void)q;
            REP8(asm volatile("v_fma_f32 %0, %0, %0, %8\n v_fma_f32 %1, %1, %1, %8\n v_fma_f32 %2, %2, %2, %8\n v_fma_f32 %3, %3, %3, %8\n"
                              "v_fma_f32 %4, %4, %4, %8\n v_fma_f32 %5, %5, %5, %8\n v_fma_f32 %6, %6, %6, %8\n v_fma_f32 %7, %7, %7, %8"
                              : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x), "+v"(p4.x), "+v"(p5.x), "+v"(p6.x), "+v"(p7.x) : "v"(1.0e-9f));)
        } else if (KIND == 3) {   // v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %8\n v_pk_fma_f32 %1, %1, %1, %8\n v_pk_fma_f32 %2, %2, %2, %8\n v_pk_fma_f32 %3, %3, %3, %8\n"
                              "v_pk_fma_f32 %4, %4, %4, %8\n v_pk_fma_f32 %5, %5, %5, %8\n v_pk_fma_f32 %6, %6, %6, %8\n v_pk_fma_f32 %7, %7, %7, %8"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(p0));)
        } else if (KIND == 4) {   // v_alignbit_b32
            REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 20\n v_alignbit_b32 %1, %1, %2, 20\n v_alignbit_b32 %2, %2, %3, 20\n v_alignbit_b32 %3, %3, %4, 20\n"
                              "v_alignbit_b32 %4, %4, %5, 20\n v_alignbit_b32 %5, %5, %6, 20\n v_alignbit_b32 %6, %6, %7, 20\n v_alignbit_b32 %7, %7, %0, 20"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 5) {   // v_pk_add_f32
            REP8(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                              "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(p0));)
        } else if (KIND == 6) {   // v_pk_min_u16 / v_pk_max_u16 (two 16-bit lanes per register)
            REP8(asm volatile("v_pk_min_u16 %0, %0, %1\n v_pk_max_u16 %1, %1, %2\n v_pk_min_u16 %2, %2, %3\n v_pk_max_u16 %3, %3, %4\n"
                              "v_pk_min_u16 %4, %4, %5\n v_pk_max_u16 %5, %5, %6\n v_pk_min_u16 %6, %6, %7\n v_pk_max_u16 %7, %7, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 8) {   // v_med3_f32
            REP8(asm volatile("v_med3_f32 %0, %0, %1, %8\n v_med3_f32 %1, %1, %2, %8\n v_med3_f32 %2, %2, %3, %8\n v_med3_f32 %3, %3, %4, %8\n"
                              "v_med3_f32 %4, %4, %5, %8\n v_med3_f32 %5, %5, %6, %8\n v_med3_f32 %6, %6, %7, %8\n v_med3_f32 %7, %7, %0, %8"
                              : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x), "+v"(p4.x), "+v"(p5.x), "+v"(p6.x), "+v"(p7.x) : "v"(1.0e-9f));)
        } else if (KIND == 9) {   // v_min_f32
            REP8(asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                              "v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8"
                              : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x), "+v"(p4.x), "+v"(p5.x), "+v"(p6.x), "+v"(p7.x) : "v"(1.0e-9f));)
        } else if (KIND == 10) {   // the sorted top-5 insert as it is today: 4 x v_med3_u32 + v_min_u32, twice, on one key set
            REP8(asm volatile("v_med3_u32 %4, %3, %4, %5\n v_med3_u32 %3, %2, %3, %5\n v_med3_u32 %2, %1, %2, %5\n v_med3_u32 %1, %0, %1, %5\n v_min_u32 %0, %0, %5\n"
                              "v_med3_u32 %4, %3, %4, %6\n v_med3_u32 %3, %2, %3, %6\n v_med3_u32 %2, %1, %2, %6\n v_med3_u32 %1, %0, %1, %6\n v_min_u32 %0, %0, %6"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(a5), "v"(a6));)
        } else if (KIND == 11) {   // the same network on float keys
            REP8(asm volatile("v_med3_f32 %4, %3, %4, %5\n v_med3_f32 %3, %2, %3, %5\n v_med3_f32 %2, %1, %2, %5\n v_med3_f32 %1, %0, %1, %5\n v_min_f32 %0, %0, %5\n"
                              "v_med3_f32 %4, %3, %4, %6\n v_med3_f32 %3, %2, %3, %6\n v_med3_f32 %2, %1, %2, %6\n v_med3_f32 %1, %0, %1, %6\n v_min_f32 %0, %0, %6"
                              : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x), "+v"(p4.x) : "v"(p5.x), "v"(p6.x));)
        } else if (KIND == 12) {   // v_sub_f32 + v_mul_f32 + v_add_f32 (plain fp32 arithmetic mix)
            REP8(asm volatile("v_sub_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                              "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                              : "+v"(p0.x), "+v"(p1.x), "+v"(p2.x), "+v"(p3.x), "+v"(p4.x), "+v"(p5.x), "+v"(p6.x), "+v"(p7.x) : "v"(1.0000001f));)
        } else if (KIND == 13) {   // integer housekeeping: v_add_u32, v_and_b32, v_lshlrev_b32, v_cndmask
            REP8(asm volatile("v_add_u32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_lshlrev_b32 %2, 1, %2\n v_cndmask_b32 %3, %3, %4, vcc\n"
                              "v_add_u32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_lshlrev_b32 %6, 1, %6\n v_cndmask_b32 %7, %7, %0, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i) : "vcc");)
        } else if (KIND >= 20 && KIND < 40) {
#define ONE2(op) REP8(asm volatile(op " %0, %0, %8\n " op " %1, %1, %8\n " op " %2, %2, %8\n " op " %3, %3, %8\n " op " %4, %4, %8\n " op " %5, %5, %8\n " op " %6, %6, %8\n " op " %7, %7, %8" \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));)
#define ONE3(op) REP8(asm volatile(op " %0, %0, %1, %8\n " op " %1, %1, %2, %8\n " op " %2, %2, %3, %8\n " op " %3, %3, %4, %8\n " op " %4, %4, %5, %8\n " op " %5, %5, %6, %8\n " op " %6, %6, %7, %8\n " op " %7, %7, %0, %8" \
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));)
            if (KIND == 20) { ONE2("v_add_u32") }
            if (KIND == 21) { ONE2("v_and_b32") }
            if (KIND == 22) { ONE2("v_lshlrev_b32") }
            if (KIND == 23) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                                                 "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");) }
            if (KIND == 24) { ONE3("v_lshl_or_b32") }
            if (KIND == 25) { ONE3("v_bfe_u32") }
            if (KIND == 26) { ONE2("v_max_f32") }
            if (KIND == 27) { ONE3("v_mad_u32_u24") }
            if (KIND == 28) { ONE2("v_xor_b32") }
            if (KIND == 29) { ONE3("v_add3_u32") }
            if (KIND == 30) { ONE2("v_mul_f32") }
            if (KIND == 31) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cmp_lt_u32 vcc, %1, %8\n v_cmp_lt_u32 vcc, %2, %8\n v_cmp_lt_u32 vcc, %3, %8\n"
                                                 "v_cmp_lt_u32 vcc, %4, %8\n v_cmp_lt_u32 vcc, %5, %8\n v_cmp_lt_u32 vcc, %6, %8\n v_cmp_lt_u32 vcc, %7, %8"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i) : "vcc");) }
            if (KIND == 32) { ONE2("v_sub_u32") }
            if (KIND == 33) { ONE2("v_max_u32") }
            if (KIND == 34) { REP8(asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
            if (KIND == 35) { ONE3("v_min3_u32") }
            if (KIND == 36) { ONE3("v_perm_b32") }
        } else if (KIND == 7) {   // v_fma_f64
            double* d = reinterpret_cast<double*>(&p0);
            (void)d;
            REP8(asm volatile("v_fma_f64 %0, %0, %0, %4\n v_fma_f64 %1, %1, %1, %4\n v_fma_f64 %2, %2, %2, %4\n v_fma_f64 %3, %3, %3, %4\n"
                              "v_fma_f64 %0, %0, %0, %4\n v_fma_f64 %1, %1, %1, %4\n v_fma_f64 %2, %2, %2, %4\n v_fma_f64 %3, %3, %3, %4"
                              : "+v"(*reinterpret_cast<double*>(&p0)), "+v"(*reinterpret_cast<double*>(&p1)), "+v"(*reinterpret_cast<double*>(&p2)),
                                "+v"(*reinterpret_cast<double*>(&p3)) : "v"(*reinterpret_cast<double*>(&p4)));)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ __float_as_uint(p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x +
                                                                                                      p0.y + p1.y + p2.y + p3.y + p4.y + p5.y + p6.y + p7.y);
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
int run(const char* name, unsigned* out, unsigned long long* cyc) {
    const int iters = 2000, per_iter = 64;   // (KIND 10 / 11: 80 per iteration, see the note printed)
    for (int wps : {1, 2, 4}) {   // waves per SIMD: a 256-thread block is one wave on each of a CU's four SIMDs
        const int grid = 256 * wps;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_probe<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_probe<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, cyc);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long c = 0;
        CK(hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost));
        const double inst_per_wave = (double)iters * per_iter;
        printf("%-16s %d waves/SIMD: %7.3f ms  in-kernel %.2f cycles per instruction per wave -> %.2f cycles per instruction per SIMD  (clock %.2f GHz)\n",
               name, wps, ms, (double)c / inst_per_wave, (double)c / inst_per_wave / wps, (double)c / (ms * 1e6));
    }
    return 0;
}

int main() {
    unsigned* out = nullptr;
    unsigned long long* cyc = nullptr;
    CK(hipMalloc((void**)&out, sizeof(unsigned) * 256 * 256 * 8));
    CK(hipMalloc((void**)&cyc, sizeof(unsigned long long)));
    run<0>("v_min_u32", out, cyc);
    run<1>("v_med3_u32", out, cyc);
    run<2>("v_fma_f32", out, cyc);
    run<3>("v_pk_fma_f32", out, cyc);
    run<5>("v_pk_add_f32", out, cyc);
    run<4>("v_alignbit_b32", out, cyc);
    run<6>("v_pk_min/max_u16", out, cyc);
    run<7>("v_fma_f64", out, cyc);
    run<8>("v_med3_f32", out, cyc);
    run<9>("v_min_f32", out, cyc);
    run<12>("fp32 sub/mul/add", out, cyc);
    run<13>("int add/and/shl/cnd", out, cyc);
    run<20>("v_add_u32", out, cyc); run<21>("v_and_b32", out, cyc); run<22>("v_lshlrev_b32", out, cyc); run<23>("v_cndmask_b32", out, cyc);
    run<24>("v_lshl_or_b32", out, cyc); run<25>("v_bfe_u32", out, cyc); run<26>("v_max_f32", out, cyc); run<27>("v_mad_u32_u24", out, cyc);
    run<28>("v_xor_b32", out, cyc); run<29>("v_add3_u32", out, cyc); run<30>("v_mul_f32", out, cyc); run<31>("v_cmp_lt_u32", out, cyc);
    run<32>("v_sub_u32", out, cyc); run<33>("v_max_u32", out, cyc); run<34>("v_cvt_f32_u32", out, cyc); run<35>("v_min3_u32", out, cyc); run<36>("v_perm_b32", out, cyc);
    printf("(the next two run 80 instructions per loop iteration: multiply the cycles by 0.8)\n");
    run<10>("top5 insert u32", out, cyc);
    run<11>("top5 insert f32", out, cyc);
    return 0;
}
