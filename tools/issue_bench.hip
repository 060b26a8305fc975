// Micro-benchmark (diagnostics, not part of the product): cycles per instruction seen by ONE
// wavefront alone on a CU for the instruction kinds the urn chain is made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
template <int KIND> __global__ void k(float* out, unsigned long long* cyc, float seed) {
    float a = seed + threadIdx.x, b = seed * 2, c = seed * 3, d = seed * 4, e = 1.0001f, f = 0.5f, g = 0.25f, h = 0.125f;
    unsigned long long t0 = clock64();
    for (int it = 0; it < N; it++) {
        if (KIND == 0) {          // 8 independent fma
            a = fmaf(a, e, f); b = fmaf(b, e, f); c = fmaf(c, e, f); d = fmaf(d, e, f);
            g = fmaf(g, e, f); h = fmaf(h, e, a * 0); e = fmaf(e, 1.0f, 0.0f); f = fmaf(f, 1.0f, 0.0f);
        } else if (KIND == 1) {   // 8 dependent fma
            a = fmaf(a, e, f); a = fmaf(a, e, f); a = fmaf(a, e, f); a = fmaf(a, e, f);
            a = fmaf(a, e, f); a = fmaf(a, e, f); a = fmaf(a, e, f); a = fmaf(a, e, f);
        } else if (KIND == 2) {   // 8 dependent dpp adds (row_shr:1)
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x111, 0xF, 0xF, true));
        } else if (KIND == 3) {   // 8 dependent permlane16_swap + add
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) { uint2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(a), false, false); a = __uint_as_float(r[0]) + __uint_as_float(r[1]); }
        } else if (KIND == 4) {   // 8 dependent cmp+cndmask
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) a = (a >= b) ? a + 1.0f : a - 1.0f;
        } else if (KIND == 5) {   // 8 independent pk_fma (16 floats)
            typedef float float2v __attribute__((ext_vector_type(2)));
            float2v x = {a, b}, y = {c, d}, z = {g, h}, w = {e, f};
            x = x * w + z; y = y * w + z; x = x * w + z; y = y * w + z; x = x * w + z; y = y * w + z; x = x * w + z; y = y * w + z;
            a = x[0]; b = x[1]; c = y[0]; d = y[1];
        } else if (KIND == 6) {   // 8 dependent v_cvt_f32_ubyte + add
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) a += (float)((__float_as_uint(a) >> 8) & 0xFF);
        } else if (KIND == 7) {   // readlane + use (dependent)
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) a += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 15));
        } else if (KIND == 8) {   // ballot + ctz + compare (SALU round trip, dependent)
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) { unsigned long long m = __ballot(a >= b); a += (float)__builtin_ctzll(m | (1ull << 63)); }
        }
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x + KIND * 64] = a + b + c + d + e + f + g + h;
    if (threadIdx.x == 0) cyc[KIND] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 64 * 16 * 4); hipMalloc(&cyc, 16 * 8);
    hipMemset(cyc, 0, 16 * 8);
#define RUN(K) hipLaunchKernelGGL(k<K>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f); hipLaunchKernelGGL(k<K>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"8 independent v_fma_f32", "8 dependent v_fma_f32", "8 dependent v_add_f32_dpp row_shr:1", "8 dependent permlane16_swap+add",
                           "8 dependent cmp+cndmask(+add)", "8 independent-ish v_pk_fma_f32", "8 dependent cvt_ubyte+add", "8 dependent readlane+add", "8 dependent ballot+ctz+cvt+add"};
    for (int i = 0; i < 9; i++) printf("%-40s %8.1f cycles / iteration  (%.1f per group of the 8)\n", names[i], (double)h[i] / N, (double)h[i] / N / 8);
    return 0;
}
