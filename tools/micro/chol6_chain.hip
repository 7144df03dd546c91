// Micro-benchmark: the 6x6 pivot Cholesky of k_band_chol (LDS -> registers -> LDS, every lane redundantly) on one wavefront, alone and with
// three other wavefronts of the workgroup busy (fp64 arithmetic / LDS reads), and the row solve + block update of half 1.
// hipcc --offload-arch=gfx950 -O3 tools/micro/chol6_chain.hip -o tools/micro/chol6_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cfloat>
__device__ __forceinline__ double fast_rsqrt(const double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}
__device__ __forceinline__ constexpr int tri6(const int i, const int j) { return i * (i + 1) / 2 + j; }
__device__ __forceinline__ bool band_chol6(const double* __restrict__ D, double a[21]) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) a[tri6(i, j)] = D[6 * i + j];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double p = a[tri6(j, j)];
        ok = ok && (p > 0.0) && (p <= DBL_MAX);
        const double r = fast_rsqrt(p);
        a[tri6(j, j)] = r;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) a[tri6(i, j)] *= r;
#pragma unroll
        for (int i = j + 1; i < 6; ++i)
#pragma unroll
            for (int k = j + 1; k <= i; ++k) a[tri6(i, k)] -= a[tri6(i, j)] * a[tri6(k, j)];
    }
    return ok;
}
// MODE 0: wave 0 alone (others exit); 1: others spin on fp64 fma; 2: others spin on LDS reads; 3: others wait at the barrier
template <int MODE>
__global__ __launch_bounds__(256) void bench(double* out, unsigned long long* t, int n) {
    __shared__ __attribute__((aligned(16))) double blk[2][36];
    __shared__ double junk[1024];
    __shared__ int flag;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid < 36) { const int i = tid / 6, j = tid % 6; blk[0][tid] = blk[1][tid] = (i == j ? 8.0 : 0.0) + 0.1 * (i + j); }
    for (int q = tid; q < 1024; q += 256) junk[q] = q & 255;
    if (tid == 0) flag = 0;
    __syncthreads();
    double acc = 0.0;
    if (wave == 0) {
        const unsigned long long w0 = wall_clock64();
        for (int it = 0; it < n; ++it) {
            double a[21];
            const bool ok = band_chol6(blk[it & 1], a);
            if (lane == 0) {
                // the next input depends on this output: a true chain through LDS, like the kernel's
                double* o = blk[(it + 1) & 1];
#pragma unroll
                for (int q = 0; q < 20; q += 2) reinterpret_cast<double2*>(o)[q >> 1] = make_double2(8.0 + a[q] * 1e-3, (q + 1 == 2 || q + 1 == 5 || q + 1 == 9 || q + 1 == 14 ? 8.0 : 0.0) + a[q + 1] * 1e-3);
                o[20] = 8.0 + a[20] * 1e-3;
                if (!ok) flag = 2;
            }
            if (MODE == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); }
            acc += a[3];
        }
        const unsigned long long w1 = wall_clock64();
        if (lane == 0) { t[0] = w1 - w0; flag = 1; }
    } else if (MODE == 1) {
        double x = 1.0000001, y = 0.9999999;
        while (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
#pragma unroll
            for (int u = 0; u < 64; ++u) x = __builtin_fma(x, y, y);
        }
        acc = x;
    } else if (MODE == 2) {
        double x = 3.0;
        while (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
#pragma unroll
            for (int u = 0; u < 16; ++u) { const double2 v = *reinterpret_cast<const double2*>(&junk[((int)x * 2 + 2 * lane) & 1022]); x = v.x + v.y * 1e-9; }
        }
        acc = x;
    } else if (MODE == 3) {
        for (int it = 0; it < n; ++it) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    out[tid] = acc;
}
template <int MODE>
void run(const char* name) {
    double* out; unsigned long long* t;
    (void)hipMalloc(&out, 1024 * 8); (void)hipMalloc(&t, 16);
    const int n = 2000;
    for (int rep = 0; rep < 2; ++rep) bench<MODE><<<1, 256>>>(out, t, n);
    (void)hipDeviceSynchronize();
    unsigned long long h[2]; (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%-64s %7.1f ns per factorisation\n", name, h[0] * 10.0 / n);
    (void)hipFree(out); (void)hipFree(t);
}
int main() {
    run<0>("6x6 Cholesky LDS -> LDS, wave 0 alone");
    run<1>("... three other waves of the workgroup in fp64 arithmetic");
    run<2>("... three other waves reading LDS");
    run<3>("... three other waves at the workgroup barrier, one per factor");
    return 0;
}
