// Latency probe (diagnostic, not product): shader clock vs the 100 MHz real-time counter, dependent fp64 FMA / rcp / rsq chains,
// LDS round trips, readlane, s_barrier — the numbers the single-workgroup kernels (k_band_chol) are designed around.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out, unsigned long long* t) {
    __shared__ double sm[1024];
    const int tid = threadIdx.x;
    double x = out[0] + tid * 1e-9, y = 1.000001;
    sm[tid] = x;
    __syncthreads();
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < 4096; ++i) x = fma(x, y, 1e-9);                 // dependent FMA chain
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3;
    for (int i = 0; i < 1024; ++i) { a0 = fma(a0, y, 1e-9); a1 = fma(a1, y, 1e-9); a2 = fma(a2, y, 1e-9); a3 = fma(a3, y, 1e-9); }   // 4 independent chains
    unsigned long long c2 = clock64();
    double r = a0 + a1 + a2 + a3;
    for (int i = 0; i < 1024; ++i) r = __builtin_amdgcn_rcp(r) + 1.5;   // rcp + add
    unsigned long long c3 = clock64();
    for (int i = 0; i < 1024; ++i) r = __builtin_amdgcn_rsq(r) + 1.5;
    unsigned long long c4 = clock64();
    int idx = tid;
    for (int i = 0; i < 1024; ++i) { sm[idx] = r; r = sm[(idx + 1) & 1023] + 1.0; }   // LDS write -> read (other address), dependent
    unsigned long long c5 = clock64();
    for (int i = 0; i < 1024; ++i) { int lo = __builtin_amdgcn_readlane(__double2loint(r), 3), hi = __builtin_amdgcn_readlane(__double2hiint(r), 3); r = __hiloint2double(hi, lo) + 1.0; }
    unsigned long long c6 = clock64();
    for (int i = 0; i < 256; ++i) { __syncthreads(); }
    unsigned long long c7 = clock64();
    for (int i = 0; i < 1024; ++i) { r = sm[(int)(r) & 1023] + 1.0; }    // dependent LDS read chain
    unsigned long long c8 = clock64(), w8 = wall_clock64();
    if (tid == 0) { t[0] = c1 - c0; t[1] = w1 - w0; t[2] = c2 - c1; t[3] = c3 - c2; t[4] = c4 - c3; t[5] = c5 - c4; t[6] = c6 - c5; t[7] = c7 - c6; t[8] = c8 - c7; t[9] = c8 - c0; t[10] = w8 - w0; }
    out[1 + tid] = r + x;
}
int main() {
    double* d; unsigned long long* t;
    hipMalloc(&d, 8 * 2048); hipMalloc(&t, 8 * 16);
    hipMemset(d, 0, 8 * 2048);
    for (int threads : { 64, 256 }) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(threads), 0, 0, d, t);
        unsigned long long h[16];
        hipMemcpy(h, t, 8 * 16, hipMemcpyDeviceToHost);
        const double mhz = (double)h[9] / ((double)h[10] * 10e-9) * 1e-6;
        printf("threads %d: clock64 ticks per 10 ns tick: %.2f (counter rate %.0f MHz)\n", threads, (double)h[9] / h[10], mhz);
        printf("  dependent fp64 FMA: %.1f ticks   4 independent chains: %.1f ticks per FMA\n", h[0] / 4096.0, h[2] / 4096.0);
        printf("  rcp+add: %.1f  rsq+add: %.1f  LDS write->read: %.1f  readlane_f64+add: %.1f  __syncthreads: %.1f  dependent LDS read+add: %.1f\n",
               h[3] / 1024.0, h[4] / 1024.0, h[5] / 1024.0, h[6] / 1024.0, h[7] / 256.0, h[8] / 1024.0);
        printf("  FMA chain wall: %.2f ns per FMA\n", h[1] * 10.0 / 4096.0);
    }
    return 0;
}
