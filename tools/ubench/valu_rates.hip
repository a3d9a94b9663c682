// Issue cost of the vector instructions the phase kernels are made of, on the GPU it runs on (gfx950):
// cycles per wavefront-instruction per SIMD, measured with enough independent chains that latency is hidden.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/ubench/valu_rates tools/ubench/valu_rates.hip
// Every test kernel runs `waves` wavefronts per SIMD (blocks of 256 threads, blocks = 256 CUs x waves), each
// executing ITER x 16 instructions of the kind under test on 16 independent registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

#define BODY16(STMT)                                                                               \
        _Pragma("unroll") for(int k = 0; k < 16; k++) { STMT; }

template <int OP>
__global__ __launch_bounds__(256) void k_rate(float *out, float seed)
{
        float a[16];
        v2f p[16];
        double d[16];
#pragma unroll
        for(int k = 0; k < 16; k++) {
                a[k] = seed + (float)(threadIdx.x + k) * 0.001f;
                p[k] = v2f{a[k], a[k] * 1.5f};
                d[k] = (double)a[k];
        }
        const float b = 0.9999f + seed * 1e-9f, c = seed * 1e-7f;
        const v2f pb = v2f{b, b}, pc = v2f{c, c};
        const double db = (double)b, dc = (double)c;
        for(int i = 0; i < ITER; i++) {
                // inline asm: the compiler would pair plain f32 operations into packed ones (SLP) and pick its own forms
                if(OP == 0) { BODY16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c))) }
                if(OP == 1) { BODY16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc))) }
                if(OP == 2) { BODY16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b))) }
                if(OP == 3) { BODY16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pb))) }
                if(OP == 4) { BODY16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c))) }
                if(OP == 5) { BODY16(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pc))) }
                if(OP == 6) { BODY16(asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]))) }
                if(OP == 7) { BODY16(asm volatile("v_rsq_f32 %0, %0" : "+v"(a[k]))) }
                if(OP == 8) { BODY16(asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[k]))) }
                if(OP == 9) { BODY16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(db), "v"(dc))) }
                if(OP == 10) { BODY16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(db))) }
                if(OP == 11) { BODY16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(dc))) }
                if(OP == 12) { BODY16(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[k]) : "v"(a[k]))) }
                if(OP == 13) { BODY16(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[k]) : "v"(d[k]))) }
                if(OP == 14) { BODY16(asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[k]))) }
                if(OP == 15) { BODY16(asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c))) }
                if(OP == 16) { BODY16(asm volatile("v_mov_b32 %0, %0" : "+v"(a[k]))) }
        }
        float s = 0.f;
#pragma unroll
        for(int k = 0; k < 16; k++) { s += a[k] + p[k].x + p[k].y + (float)d[k]; }
        if(s == 123.456f) { out[threadIdx.x] = s; }
}

template <int OP>
static double run(int waves_per_simd, float *dout)
{
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        const int blocks = 256 * waves_per_simd;       // 4 waves per block = 1 per SIMD per CU
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, dout, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, dout, 1.0f);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        return ms;
}

int main()
{
        float *dout;
        hipMalloc(&dout, 4096);
        int clk_khz = 0;
        hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
        const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f32", "v_pk_mul_f32", "v_add_f32", "v_pk_add_f32", "v_rcp_f32",
                               "v_rsq_f32", "v_sqrt_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_cvt_f64_f32",
                               "v_cvt_f32_f64", "v_mov_dpp wave_shr", "v_max_f32", "v_mov_b32"};
        printf("{\"clock_khz\": %d, \"iter\": %d, \"rates\": {", clk_khz, ITER);
        for(int w : {1, 2, 4, 8}) {
                double ms[17];
                ms[0] = run<0>(w, dout); ms[1] = run<1>(w, dout); ms[2] = run<2>(w, dout); ms[3] = run<3>(w, dout);
                ms[4] = run<4>(w, dout); ms[5] = run<5>(w, dout); ms[6] = run<6>(w, dout); ms[7] = run<7>(w, dout);
                ms[8] = run<8>(w, dout); ms[9] = run<9>(w, dout); ms[10] = run<10>(w, dout); ms[11] = run<11>(w, dout);
                ms[12] = run<12>(w, dout); ms[13] = run<13>(w, dout); ms[14] = run<14>(w, dout); ms[15] = run<15>(w, dout);
                ms[16] = run<16>(w, dout);
                printf("%s\"waves_per_simd_%d\": {", w == 1 ? "" : ", ", w);
                for(int i = 0; i < 17; i++) {
                        // wavefront-instructions per SIMD = w * ITER * 16; cycles at the nominal clock
                        const double cyc = ms[i] * 1e-3 * (double)clk_khz * 1e3 / ((double)w * ITER * 16);
                        printf("%s\"%s\": %.2f", i ? ", " : "", names[i], cyc);
                }
                printf("}");
        }
        printf("}, \"unit\": \"cycles per wavefront-instruction per SIMD at the nominal clock (kernel time x clock / instructions per SIMD)\"}\n");
        return 0;
}
