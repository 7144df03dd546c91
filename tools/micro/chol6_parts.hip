// Micro-benchmark: what the parts of the 6x6 pivot Cholesky cost on one wavefront (registers only: no LDS in the loop).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ constexpr int tri6(const int i, const int j) { return i * (i + 1) / 2 + j; }
template <int MODE>
__device__ __forceinline__ double rs(const double x) {
    if (MODE == 2) return x * 0.125;                                   // no reciprocal root at all
    const double y0 = __builtin_amdgcn_rsq(x);
    if (MODE == 1) return y0;                                          // seed only
    const double e = __builtin_fma(-(x * y0), y0, 1.0);
    return __builtin_fma(y0 * e, __builtin_fma(0.375, e, 0.5), y0);
}
template <int MODE>
__global__ __launch_bounds__(64) void bench(double* out, unsigned long long* t, int n, double d0, double o0) {
    double a[21];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) a[tri6(i, j)] = i == j ? d0 : o0 * (i + j);
    double carry = 0.0;
    const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
    for (int it = 0; it < n; ++it) {
        if (MODE <= 2) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const double r = rs<MODE>(a[tri6(j, j)]);
                a[tri6(j, j)] = r;
#pragma unroll
                for (int i = j + 1; i < 6; ++i) a[tri6(i, j)] *= r;
#pragma unroll
                for (int i = j + 1; i < 6; ++i)
#pragma unroll
                    for (int k = j + 1; k <= i; ++k) a[tri6(i, k)] -= a[tri6(i, j)] * a[tri6(k, j)];
            }
            carry = a[20];
            // the next block depends on this one (a chain, as in the kernel): diagonal back to ~d0, off-diagonal small
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j) a[tri6(i, j)] = i == j ? d0 + carry * 1e-9 : o0 * (i + j);
        } else if (MODE == 3) {
            // 105 fp64 operations with the register pattern of a trailing update, no chain longer than 5
#pragma unroll
            for (int rep = 0; rep < 3; ++rep)
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int i = j + 1; i < 6; ++i)
#pragma unroll
                        for (int k = j + 1; k <= i; ++k) a[tri6(i, k)] = __builtin_fma(-a[tri6(i, j)], a[tri6(k, j)], a[tri6(i, k)]) ;
        }
    }
    const unsigned long long w1 = wall_clock64(), c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { t[0] = w1 - w0; t[1] = c1 - c0; }
    double s = carry;
#pragma unroll
    for (int q = 0; q < 21; ++q) s += a[q];
    out[threadIdx.x] = s;
}
template <int MODE>
void run(const char* name) {
    double* out; unsigned long long* t;
    (void)hipMalloc(&out, 1024 * 8); (void)hipMalloc(&t, 16);
    const int n = 2000;
    for (int rep = 0; rep < 2; ++rep) bench<MODE><<<1, 64>>>(out, t, n, 8.0, 0.01);
    (void)hipDeviceSynchronize();
    unsigned long long h[2]; (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%-72s %7.1f ns  %7.0f shader clocks\n", name, h[0] * 10.0 / n, (double)h[1] / n);
    (void)hipFree(out); (void)hipFree(t);
}
int main() {
    run<0>("6x6 Cholesky in registers (rsq seed + third-order step)");
    run<1>("... reciprocal root = the v_rsq_f64 seed alone");
    run<2>("... no reciprocal root (a multiply in its place)");
    run<3>("105 fma of the trailing-update pattern, short chains");
    return 0;
}
