// Micro-benchmark (diagnostics): cost of one cross-wave exchange through LDS with 4 waves
// (one per SIMD): every wave writes one float per lane, barrier, every wave reads all four.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
typedef float f4v __attribute__((ext_vector_type(4)));
template <int KIND> __global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float seed) {
    __shared__ __attribute__((aligned(16))) float x[64 * 4];
    __shared__ __attribute__((aligned(16))) float y[64 * 8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float a = seed + lane + w;
    __syncthreads();
    unsigned long long t0 = clock64();
    for (int it = 0; it < N; it++) {
        if (KIND == 0) {          // one exchange
            x[lane * 4 + w] = a;
            __syncthreads();
            const f4v v = *(const f4v*)(x + lane * 4);
            a = (v.x + v.y) + (v.z + v.w) * 0.25f;
        } else if (KIND == 1) {   // two exchanges (as one pass of the split chain)
            x[lane * 4 + w] = a;
            __syncthreads();
            const f4v v = *(const f4v*)(x + lane * 4);
            a = (v.x + v.y) + (v.z + v.w) * 0.25f;
            y[(lane * 4 + w) * 2] = a; y[(lane * 4 + w) * 2 + 1] = a * 0.5f;
            __syncthreads();
            const f4v p = *(const f4v*)(y + lane * 8), q = *(const f4v*)(y + lane * 8 + 4);
            a = fminf(fminf(p.y, p.w), fminf(q.y, q.w)) + (p.x + p.z + q.x + q.z) * 0.125f;
        } else {                  // barrier only
            __syncthreads();
            a += 1.0f;
        }
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x + KIND * 256] = a;
    if (threadIdx.x == 0) cyc[KIND] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 4 * 4); hipMalloc(&cyc, 16 * 8);
    hipMemset(cyc, 0, 16 * 8);
#define RUN(K) hipLaunchKernelGGL(k<K>, dim3(1), dim3(256), 0, 0, out, cyc, 1.0f); hipLaunchKernelGGL(k<K>, dim3(1), dim3(256), 0, 0, out, cyc, 1.0f);
    RUN(0) RUN(1) RUN(2)
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"write b32 + barrier + read b128 + 3 adds", "two exchanges", "barrier + 1 add"};
    for (int i = 0; i < 3; i++) printf("%-48s %8.1f cycles / iteration\n", names[i], (double)h[i] / N);
    return 0;
}
