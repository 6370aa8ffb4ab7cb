// stream_rows.hip -- practical HBM ceiling for the decoders' access pattern: every wave owns
// one record at a time and moves it in 256-byte rows (one dword per lane) or 1-KiB rows
// (one dwordx4 per lane).  Prints achieved TB/s for a few read:write mixes.
//   hipcc --offload-arch=gfx950 -O3 -o stream_rows stream_rows.hip && ./stream_rows
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int VEC, int INFLIGHT>
__global__ __launch_bounds__(512) void k_rows(const float *__restrict__ in, float *__restrict__ out,
                                              int rd_words, int wr_words, unsigned long long n)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x % 64;
    const int waves = blockDim.x / 64;
    for (unsigned long long f = (unsigned long long)blockIdx.x * waves + wave; f < n;
         f += (unsigned long long)gridDim.x * waves) {
        const float *src = in + f * rd_words;
        float *dst = out + f * wr_words;
        float acc = 0.0f;
        if (VEC == 1) {
            for (int r = 0; r < rd_words; r += 64 * INFLIGHT) {
                float v[INFLIGHT];
#pragma unroll
                for (int i = 0; i < INFLIGHT; i++) v[i] = r + i * 64 < rd_words ? src[r + i * 64 + lane] : 0.0f;
#pragma unroll
                for (int i = 0; i < INFLIGHT; i++) acc += v[i];
            }
            for (int r = 0; r < wr_words; r += 64) dst[r + lane] = acc + (float)r;
        } else {
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            float4 *d4 = reinterpret_cast<float4 *>(dst);
            for (int r = 0; r < rd_words / 4; r += 64 * INFLIGHT) {
                float4 v[INFLIGHT];
#pragma unroll
                for (int i = 0; i < INFLIGHT; i++) v[i] = r + i * 64 < rd_words / 4 ? s4[r + i * 64 + lane] : make_float4(0, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < INFLIGHT; i++) acc += v[i].x + v[i].w;
            }
            for (int r = 0; r < wr_words / 4; r += 64) d4[r + lane] = make_float4(acc, acc, (float)r, acc);
        }
    }
}

int main()
{
    const unsigned long long n = 65536;
    const int cases[][2] = { {9 * 1024, 16 * 1024}, {16 * 1024, 16 * 1024}, {24 * 1024, 8 * 1024}, {16 * 1024, 0}, {0, 16 * 1024},
                             {1536, 2560} };   // words: PS-like, 1:1, synth-like, read only, write only, LC-like (6 KiB : 10 KiB)
    float *in, *out;
    hipMalloc(&in, n * 24 * 1024 * 4);
    hipMalloc(&out, n * 16 * 1024 * 4);
    hipMemset(in, 0, n * 24 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto &c : cases) {
        for (int variant = 0; variant < 4; variant++) {
            const int waves = (variant & 1) ? 16 : 8;          // waves per workgroup-CU: 8 or 16 (2 blocks of 8)
            const int grid = 256 * (waves / 8);
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (variant < 2) hipLaunchKernelGGL((k_rows<1, 16>), dim3(grid), dim3(512), 0, 0, in, out, c[0], c[1], n);
                else             hipLaunchKernelGGL((k_rows<4, 4>), dim3(grid), dim3(512), 0, 0, in, out, c[0], c[1], n);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double bytes = (double)n * (c[0] + c[1]) * 4;
            printf("rd %5.1f KiB wr %5.1f KiB  %s  %2d waves/CU : %7.1f us  %5.2f TB/s\n", c[0] / 256.0, c[1] / 256.0,
                   variant < 2 ? "dword  " : "dwordx4", waves, ms * 1e3, bytes / ms / 1e9);
        }
    }
    return 0;
}
