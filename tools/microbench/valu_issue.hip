// valu_issue.hip -- how many vector instructions per cycle one SIMD of gfx950 retires with 1..4 waves
// resident, for the forms the decoders' inner loops are made of (plain and packed f32 add / mul,
// independent and dependent).  Answers whether a kernel that runs two waves per SIMD at ~45 % issue
// activity each is issue-bound or latency-bound.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 0: 8 independent v_add_f32      1: 8 independent v_pk_add_f32     2: 8 independent v_pk_mul_f32
//      3: one dependent v_add_f32 chain 4: one dependent v_pk_add_f32 chain   5: 4 v_mul + 4 v_add
//      6: 8 independent v_fma_f32      7: 8 independent v_pk_fma_f32
template <int MODE>
__global__ __launch_bounds__(256) void k_issue(float *out, int iters)
{
    float a[8];
    v2f p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 0.001f + i; p[i] = v2f{a[i], a[i] + 0.5f}; }
    const float c = 1.0000001f;
    const v2f pc = v2f{c, c};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
        } else if (MODE == 3) {
#pragma unroll
            for (int r = 0; r < 32; r++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
        } else if (MODE == 4) {
#pragma unroll
            for (int r = 0; r < 32; r++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[0]) : "v"(pc));
        } else if (MODE == 5) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
#pragma unroll
                for (int i = 0; i < 4; i++) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#pragma unroll
                for (int i = 4; i < 8; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            }
        } else if (MODE == 6) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
        } else {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(pc));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) reinterpret_cast<unsigned long long *>(out + (1 << 20))[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *name, float *out, unsigned long long *h_cyc)
{
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps++) {       // waves per SIMD: wps blocks of 4 waves per CU
        const int grid = 256 * wps;
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k_issue<MODE>), dim3(grid), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(h_cyc, out + (1 << 20), grid * 8, hipMemcpyDeviceToHost);
        double cyc = 0;
        for (int i = 0; i < grid; i++) cyc += (double)h_cyc[i];
        cyc /= grid;                           // s_memtime ticks (shader clock) of one wave's loop
        const double inst = 32.0 * iters;      // per wave
        printf("%-28s %d waves/SIMD: %8.1f us  %6.2f cycles/inst/wave  %5.2f inst/cycle/SIMD  (clock %.2f GHz)\n",
               name, wps, ms * 1e3, cyc / inst, inst * wps / cyc, cyc / (ms * 1e6));
    }
}

int main()
{
    float *out;
    hipMalloc(&out, ((1 << 20) + 4096) * 4);
    unsigned long long *h = (unsigned long long *)malloc(1024 * 8);
    run<0>("v_add_f32 x8 independent", out, h);
    run<5>("v_mul/v_add_f32 independent", out, h);
    run<6>("v_fma_f32 x8 independent", out, h);
    run<1>("v_pk_add_f32 x8 independent", out, h);
    run<2>("v_pk_mul_f32 x8 independent", out, h);
    run<7>("v_pk_fma_f32 x8 independent", out, h);
    run<3>("v_add_f32 dependent chain", out, h);
    run<4>("v_pk_add_f32 dependent chain", out, h);
    return 0;
}
