// Micro-benchmark (diagnostics, not part of the product): cycles per instruction seen by ONE
// wavefront alone on a CU for the instruction kinds of the wide urn chain (urn_chain_w).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
template <int KIND> __global__ void k(float* out, unsigned long long* cyc, float seed, int zero) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int lane = threadIdx.x;
    for (int i = lane; i < 8192; i += 64) lds[i] = 0.001f * i;
    __syncthreads();
    float a = seed + lane, b = seed * 2, c = seed * 3, d = seed * 4, e = 1.0001f, f = 0.5f, g = 0.25f, h = 0.125f;
    unsigned w0 = lane, w1 = lane * 3, w2 = 5, w3 = 7, w4 = 9, w5 = 11, w6 = 13, w7 = 15;
    f2v p0 = {a, b}, p1 = {c, d}, p2 = {e, f}, p3 = {g, h};
    int acc = 0;
    unsigned long long t0 = clock64();
    for (int it = 0; it < N; it++) {
        if (KIND == 0) {          // 8 independent v_alignbit
            w0 = __builtin_amdgcn_alignbit(w0, __float_as_uint(a), 31); w1 = __builtin_amdgcn_alignbit(w1, __float_as_uint(b), 31);
            w2 = __builtin_amdgcn_alignbit(w2, __float_as_uint(c), 31); w3 = __builtin_amdgcn_alignbit(w3, __float_as_uint(d), 31);
            w4 = __builtin_amdgcn_alignbit(w4, __float_as_uint(e), 31); w5 = __builtin_amdgcn_alignbit(w5, __float_as_uint(f), 31);
            w6 = __builtin_amdgcn_alignbit(w6, __float_as_uint(g), 31); w7 = __builtin_amdgcn_alignbit(w7, __float_as_uint(h), 31);
        } else if (KIND == 1) {   // 8 dependent v_alignbit
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) w0 = __builtin_amdgcn_alignbit(w0, __float_as_uint(a), 31);
        } else if (KIND == 2) {   // 8 independent v_min3 |x| |y|
            asm volatile("v_min3_f32 %0, %0, |%8|, |%9|\n v_min3_f32 %1, %1, |%8|, |%9|\n v_min3_f32 %2, %2, |%8|, |%9|\n v_min3_f32 %3, %3, |%8|, |%9|\n"
                         "v_min3_f32 %4, %4, |%8|, |%9|\n v_min3_f32 %5, %5, |%8|, |%9|\n v_min3_f32 %6, %6, |%8|, |%9|\n v_min3_f32 %7, %7, |%8|, |%9|\n"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(seed), "v"(seed));
        } else if (KIND == 3) {   // 8 dependent v_min3
            asm volatile("v_min3_f32 %0, %0, |%1|, |%2|\n v_min3_f32 %0, %0, |%1|, |%2|\n v_min3_f32 %0, %0, |%1|, |%2|\n v_min3_f32 %0, %0, |%1|, |%2|\n"
                         "v_min3_f32 %0, %0, |%1|, |%2|\n v_min3_f32 %0, %0, |%1|, |%2|\n v_min3_f32 %0, %0, |%1|, |%2|\n v_min3_f32 %0, %0, |%1|, |%2|\n"
                         : "+v"(a) : "v"(b), "v"(c));
        } else if (KIND == 4) {   // 8 independent v_pk_add_f32 (neg)
            asm volatile("v_pk_add_f32 %0, %0, %4 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %4 neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %2, %2, %4 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %4 neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %0, %0, %4 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %4 neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %2, %2, %4 neg_lo:[0,1] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %4 neg_lo:[0,1] neg_hi:[0,1]\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p3));
        } else if (KIND == 5) {   // two interleaved chains of 4 dependent v_fma
            asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         "v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3\n"
                         : "+v"(a), "+v"(b) : "v"(e), "v"(f));
        } else if (KIND == 6) {   // commit round trip: ds_add_f32 (3 hot addresses) then 4 broadcast ds_read_b128, wait
            __hip_atomic_fetch_add(&lds[(lane % 3) * 5], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            f4v x0 = *(const f4v*)(lds + 0), x1 = *(const f4v*)(lds + 4), x2 = *(const f4v*)(lds + 8), x3 = *(const f4v*)(lds + 12);
            a += x0.x + x1.y + x2.z + x3.w;
            asm volatile("" ::: "memory");
        } else if (KIND == 7) {   // 8 ds_read_b128 of a row (stride 36 floats per lane), wait
            const float* r = lds + ((lane + it) & 63) * 36 + zero;
            f4v x0 = *(const f4v*)(r + 0), x1 = *(const f4v*)(r + 4), x2 = *(const f4v*)(r + 8), x3 = *(const f4v*)(r + 12);
            f4v x4 = *(const f4v*)(r + 16), x5 = *(const f4v*)(r + 20), x6 = *(const f4v*)(r + 24), x7 = *(const f4v*)(r + 28);
            a += x0.x + x1.y + x2.z + x3.w + x4.x + x5.y + x6.z + x7.w;
            asm volatile("" ::: "memory");
        } else if (KIND == 8) {   // only 4 broadcast reads, wait (plain LDS latency)
            f4v x0 = *(const f4v*)(lds + 0 + zero), x1 = *(const f4v*)(lds + 4 + zero), x2 = *(const f4v*)(lds + 8 + zero), x3 = *(const f4v*)(lds + 12 + zero);
            a += x0.x + x1.y + x2.z + x3.w;
            asm volatile("" ::: "memory");
        } else if (KIND == 9) {   // compare -> ballot -> ff1 -> scalar -> vector use
            const unsigned long long m = ~__ballot(a >= b);
            int adv = m ? (int)__builtin_ctzll(m) : 64;
            if (adv == 0) { a += 3.0f; adv = 1; }
            acc += adv;
            a += (float)((lane < adv) ? 1 : 0);
        } else if (KIND == 10) {  // ds_add_f32 with every lane on its own address, then the reads
            __hip_atomic_fetch_add(&lds[lane], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            f4v x0 = *(const f4v*)(lds + 0), x1 = *(const f4v*)(lds + 4), x2 = *(const f4v*)(lds + 8), x3 = *(const f4v*)(lds + 12);
            a += x0.x + x1.y + x2.z + x3.w;
            asm volatile("" ::: "memory");
        } else if (KIND == 12) {  // ds_add_u32 3 hot addresses + reads
            __hip_atomic_fetch_add((unsigned*)&lds[(lane % 3) * 5], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            f4v x0 = *(const f4v*)(lds + 0), x1 = *(const f4v*)(lds + 4), x2 = *(const f4v*)(lds + 8), x3 = *(const f4v*)(lds + 12);
            a += x0.x + x1.y + x2.z + x3.w;
            asm volatile("" ::: "memory");
        } else if (KIND == 13) {  // ds_add_u32 64 addresses + reads
            __hip_atomic_fetch_add((unsigned*)&lds[lane], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            f4v x0 = *(const f4v*)(lds + 0), x1 = *(const f4v*)(lds + 4), x2 = *(const f4v*)(lds + 8), x3 = *(const f4v*)(lds + 12);
            a += x0.x + x1.y + x2.z + x3.w;
            asm volatile("" ::: "memory");
        } else if (KIND == 14) {  // ds_add_f32 3 hot addresses, only 16 lanes active, + reads
            if ((lane & 3) == 0) __hip_atomic_fetch_add(&lds[(lane % 3) * 5], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            f4v x0 = *(const f4v*)(lds + 0), x1 = *(const f4v*)(lds + 4), x2 = *(const f4v*)(lds + 8), x3 = *(const f4v*)(lds + 12);
            a += x0.x + x1.y + x2.z + x3.w;
            asm volatile("" ::: "memory");
        } else if (KIND == 15) {  // plain ds_write_b32 (64 addresses) + reads
            lds[lane + 64] = a;
            asm volatile("" ::: "memory");
            f4v x0 = *(const f4v*)(lds + 0), x1 = *(const f4v*)(lds + 4), x2 = *(const f4v*)(lds + 8), x3 = *(const f4v*)(lds + 12);
            a += x0.x + x1.y + x2.z + x3.w;
            asm volatile("" ::: "memory");
        } else if (KIND == 11) {  // ds_add_f32 all lanes on ONE address, then the reads
            __hip_atomic_fetch_add(&lds[7], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            f4v x0 = *(const f4v*)(lds + 0), x1 = *(const f4v*)(lds + 4), x2 = *(const f4v*)(lds + 8), x3 = *(const f4v*)(lds + 12);
            a += x0.x + x1.y + x2.z + x3.w;
            asm volatile("" ::: "memory");
        }
    }
    unsigned long long t1 = clock64();
    out[threadIdx.x + KIND * 64] = a + b + c + d + e + f + g + h + (float)(w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7) + p0.x + p1.y + p2.x + p3.y + acc;
    if (threadIdx.x == 0) cyc[KIND] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 64 * 32 * 4); hipMalloc(&cyc, 16 * 8);
    hipMemset(cyc, 0, 16 * 8);
#define RUN(K) hipLaunchKernelGGL(k<K>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f, 0); hipLaunchKernelGGL(k<K>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0f, 0);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15)
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"8 independent v_alignbit_b32", "8 dependent v_alignbit_b32", "8 independent v_min3_f32", "8 dependent v_min3_f32",
                           "8 independent v_pk_add_f32", "2 chains x 4 dependent v_fma_f32", "ds_add_f32 (3 hot addrs) + 4 bcast b128 reads", "8 ds_read_b128 row reads (stride 36)",
                           "4 bcast ds_read_b128", "cmp->ballot->ff1->branch->vector", "ds_add_f32 (64 addrs) + 4 bcast reads", "ds_add_f32 (1 addr) + 4 bcast reads", "ds_add_u32 (3 hot addrs) + 4 bcast reads", "ds_add_u32 (64 addrs) + 4 bcast reads", "ds_add_f32 (3 hot, 16 lanes) + 4 bcast reads", "ds_write_b32 + 4 bcast reads"};
    for (int i = 0; i < 16; i++) printf("%-48s %8.1f cycles / iteration\n", names[i], (double)h[i] / N);
    return 0;
}
