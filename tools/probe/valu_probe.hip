// VALU issue-rate probe for gfx950: cycles per wave-instruction per SIMD for the instruction kinds the reprojection
// kernel is made of.  Each block = 256 threads (one wave per SIMD) x WPS blocks per CU resident; every wave runs N
// back-to-back independent instructions of one kind (8 independent register chains).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    float a[8], b[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8], q[8];
    for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x * 0.001f + k; b[k] = 1.0001f + k * 1e-4f; p[k] = f2{a[k], b[k]}; q[k] = f2{b[k], a[k]}; }
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 1) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[k]) : "v"(q[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 2) {
#define X(k) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 3) {
#define X(k) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 4) {
#define X(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 5) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 6) {
#define X(k) asm volatile("v_mov_b32 %0, %1" : "=v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 7) {
#define X(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(q[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 8) {
#define X(k) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(q[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 9) {
#define X(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 10) {
#define X(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[k]), "v"(b[k]) : "vcc");
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 11) {
#define X(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 12) {
#define X(k) asm volatile("v_floor_f32 %0, %0" : "+v"(a[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 13) {
#define X(k) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 15) {
#define X(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 16) {
#define X(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[k]) : "v"(b[k]) : "s10", "s11");
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 17) {
#define X(k) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b[k]) : "vcc");
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 18) {
#define X(k) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 19) {
#define X(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 20) {
#define X(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 21) {
#define X(k) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 22) {
#define X(k) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 23) {
#define X(k) asm volatile("v_fma_f32 %0, |%0|, %1, -%1" : "+v"(a[k]) : "v"(b[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 24) {
#define X(k) asm volatile("v_exp_f32 %0, %0" : "+v"(a[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 25) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b[k]));
            asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        } else if (KIND == 14) {
#define X(k) asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(p[k]) : "v"(q[k]));
            REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
        }
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += a[k] + p[k].x + p[k].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, int wps) {
    float* out;
    const int blocks = 256 * wps;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    const int iters = 60000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<blocks, 256>>>(out, 100);
    hipEventRecord(e0);
    probe<KIND><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: wps waves x iters x 32 instructions
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * iters * 32);
    printf("%-28s waves/SIMD=%d  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, wps, cyc);
    hipFree(out);
}

int main() {
    for (int wps : {4}) {
        run<15>("v_add_f32", wps);
        run<16>("v_cndmask_b32_e64 sgpr", wps);
        run<17>("v_cmp+v_cndmask (2 instr)", wps);
        run<25>("v_cndmask vcc (defined)", wps);
        run<18>("v_min_i32", wps);
        run<19>("v_mad_u32_u24", wps);
        run<20>("v_and_b32", wps);
        run<21>("v_lshl_add_u32", wps);
        run<22>("v_med3_f32", wps);
        run<23>("v_fma_f32 abs/neg mods", wps);
        run<24>("v_exp_f32", wps);
        run<0>("v_fma_f32", wps);
        run<9>("v_mul_f32", wps);
        run<1>("v_pk_fma_f32", wps);
        run<7>("v_pk_mul_f32", wps);
        run<8>("v_pk_add_f32", wps);
        run<2>("v_add_f32 dpp row_shr", wps);
        run<3>("v_add_f32 dpp wave_shr", wps);
        run<4>("v_rcp_f32", wps);
        run<5>("v_cndmask_b32", wps);
        run<6>("v_mov_b32", wps);
        run<14>("v_pk_mov_b32", wps);
        run<10>("v_cmp_lt_f32", wps);
        run<11>("v_max_f32", wps);
        run<12>("v_floor_f32", wps);
        run<13>("v_cvt_i32_f32", wps);
    }
    return 0;
}
