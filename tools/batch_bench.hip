// Diagnostics only: cycles of one urn-chain batch (NPL = 1 layout) with pieces switched off.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
template <int CTRL, int RM, int BM, bool BC> __device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, RM, BM, BC));
}
__device__ __forceinline__ float row_scan16(float v) {
    v += dpp<0x111, 0xF, 0xF, true>(v); v += dpp<0x112, 0xF, 0xF, true>(v);
    v += dpp<0x114, 0xF, 0xF, true>(v); v += dpp<0x118, 0xF, 0xF, true>(v); return v;
}
__device__ __forceinline__ float rows_sum4(float x) {
    uint2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    const float y = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    uint2v r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
    return __uint_as_float(r2[0]) + __uint_as_float(r2[1]);
}
// MASK bits: 32 blocks of 4 batches with ballot + commit, 64 other waves parked at a barrier; 1 exchange, 2 margin, 4 history, 8 use med3 flags (else saturate), 16 LDS operand loads
template <int MASK> __global__ void k(float* out, unsigned long long* cyc, const float* uin, int slot) {
    __shared__ float rows[4096];
    __shared__ float ubuf[1024];
    const int lane = threadIdx.x & 63, row = lane >> 4, col = lane & 15;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) rows[i] = 0.5f + 0.001f * (i % 97);
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) ubuf[i] = uin[i];
    __syncthreads();
    float a0f = 10.0f + col, kf = 0.0f, slack = 1e30f;
    unsigned pk = 0;
    const float mg_add = (float)row + 1e-37f;
    float Ln = rows[col], ufn = ubuf[row];
    int qrn = row, tn = row;
    unsigned cnt[7] = {0, 0, 0, 0, 0, 0, 0};
    float kf0 = 0;
    unsigned long long t0 = clock64();
    if (threadIdx.x < 64)
#pragma unroll 1
    for (int it = 0; it < N; it++) {
        if ((MASK & 32) && (it & 3) == 0) { kf0 = kf; pk = 0; slack = 1e30f; }
        float L = Ln, uf = ufn;
        if (MASK & 16) { qrn += 4; qrn = qrn >= 200 ? qrn - 200 : qrn; tn += 4; Ln = rows[qrn * 16 + col]; ufn = ubuf[tn & 1023]; }
        const float tot = (a0f + kf) * L;
        const float incl = row_scan16(tot);
        const float base = dpp<0x111, 0xF, 0xF, true>(incl);
        const float T = dpp<0x15F, 0xF, 0xF, true>(incl);
        const float tgt = uf * T;
        const float mg = fmaf(1e-5f, T, mg_add);
        const float d0 = base - tgt, d1 = incl - tgt;
        float f0, f1;
        if (MASK & 8) { f0 = __builtin_amdgcn_fmed3f(d0 * 1e30f, 0.0f, 1.0f); f1 = __builtin_amdgcn_fmed3f(d1 * 1e30f, 0.0f, 1.0f); }
        else { f0 = __saturatef(d0 * 1e30f); f1 = __saturatef(d1 * 1e30f); }
        const float sel = f1 - f0;
        if (MASK & 1) kf += rows_sum4(sel); else kf += sel;
        if (MASK & 4) pk += ((unsigned)sel) << (slot & 31);
        if (MASK & 2) slack = fminf(slack, fminf(fabsf(d0), fabsf(d1)) - mg);
        if ((MASK & 32) && (it & 3) == 3) {
            if (__ballot(!(slack >= 0.0f)) == 0ull) {
#pragma unroll
                for (int b = 0; b < 7; b++) cnt[b] += (pk >> (4 * b)) & 15u;
            } else { kf = kf0 + 1.0f; }
        }
    }
    if (MASK & 64) __syncthreads();
    unsigned long long t1 = clock64();
    out[lane & 63] = kf + slack + pk + cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4] + cnt[5] + cnt[6];
    if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
    float *out, *uin; unsigned long long* cyc;
    hipMalloc(&out, 256); hipMalloc(&cyc, 8); hipMalloc(&uin, 4096);
    float hu[1024]; for (int i = 0; i < 1024; i++) hu[i] = (float)((i * 7919) % 1000) / 1000.0f;
    hipMemcpy(uin, hu, 4096, hipMemcpyHostToDevice);
#define RUN(M, NAME) { hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, uin, 4); hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, uin, 4); hipDeviceSynchronize(); \
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-60s %7.1f cycles / batch\n", NAME, (double)h / N); }
    RUN(0, "core only (scan, T, flags by saturate, k += sel)")
    RUN(8, "core, flags by med3")
    RUN(8 | 1, "+ cross-row exchange (permlane16/32 swap)")
    RUN(8 | 1 | 2, "+ margin / slack")
    RUN(8 | 1 | 2 | 4, "+ history")
    RUN(8 | 1 | 2 | 4 | 16, "+ LDS operand loads one batch ahead")
    RUN(1 | 2 | 4 | 16, "same with saturate flags")
    RUN(8 | 1 | 2 | 4 | 16 | 32, "+ blocks of 4 batches: ballot + commit")
#undef RUN
#define RUN(M, NAME) { hipLaunchKernelGGL(k<M>, dim3(1), dim3(256), 0, 0, out, cyc, uin, 4); hipLaunchKernelGGL(k<M>, dim3(1), dim3(256), 0, 0, out, cyc, uin, 4); hipDeviceSynchronize(); \
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-60s %7.1f cycles / batch\n", NAME, (double)h / N); }
    RUN(8 | 1 | 2 | 4 | 16 | 32 | 64, "+ 3 more waves parked at __syncthreads")
    return 0;
}
