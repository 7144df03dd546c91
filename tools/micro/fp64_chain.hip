// Micro-benchmark: what a DEPENDENT chain of fp64 instructions costs on one gfx950 wavefront (the banded Cholesky's pivot chain).
// hipcc --offload-arch=gfx950 -O3 tools/micro/fp64_chain.hip -o /tmp/fp64_chain && /tmp/fp64_chain
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned long long clk() { return __builtin_readcyclecounter(); }     // s_memtime: shader clock
__device__ __forceinline__ unsigned long long wall() { return wall_clock64(); }                  // 100 MHz
template <int MODE>
__global__ void chain(double* out, unsigned long long* t, double a, double b, int n) {
    __shared__ double lds[256];
    lds[threadIdx.x] = a + threadIdx.x;
    __syncthreads();
    double x = a, y = b, z = a + b, w = a - b;
    const unsigned long long c0 = clk(), w0 = wall();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {                     // dependent fma
#pragma unroll
            for (int u = 0; u < 16; ++u) x = __builtin_fma(x, y, y);
        } else if (MODE == 1) {              // 4 independent fma chains
#pragma unroll
            for (int u = 0; u < 4; ++u) { x = __builtin_fma(x, y, y); z = __builtin_fma(z, y, y); w = __builtin_fma(w, y, y); a = __builtin_fma(a, y, y); }
        } else if (MODE == 2) {              // dependent v_rsq_f64
#pragma unroll
            for (int u = 0; u < 16; ++u) x = __builtin_amdgcn_rsq(x) + y;
        } else if (MODE == 3) {              // dependent LDS round trip (broadcast read, address from the value)
#pragma unroll
            for (int u = 0; u < 16; ++u) x = lds[(int)x & 255];
        } else if (MODE == 4) {              // dependent mul
#pragma unroll
            for (int u = 0; u < 16; ++u) x = x * y;
        } else if (MODE == 5) {              // dependent f32 fma
            float xf = (float)x, yf = (float)y;
#pragma unroll
            for (int u = 0; u < 16; ++u) xf = __builtin_fmaf(xf, yf, yf);
            x = xf;
        } else if (MODE == 6) {              // LDS write then read by another lane of the wave
#pragma unroll
            for (int u = 0; u < 16; ++u) { lds[threadIdx.x] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); x = lds[threadIdx.x ^ 1]; }
        } else if (MODE == 7) {              // workgroup barrier (lgkmcnt only)
#pragma unroll
            for (int u = 0; u < 16; ++u) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
        }
    }
    const unsigned long long c1 = clk(), w1 = wall();
    if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
    out[threadIdx.x] = x + z + w + a;
}
template <int MODE>
void run(const char* name, int threads, int per_iter) {
    double* out; unsigned long long* t;
    hipMalloc(&out, 1024 * 8); hipMalloc(&t, 16);
    const int n = 4096;
    for (int rep = 0; rep < 2; ++rep) chain<MODE><<<1, threads>>>(out, t, MODE == 3 ? 3.0 : 1.0000001, 0.9999999, n);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    const double ops = (double)n * per_iter;
    printf("%-44s threads %4d: %7.2f shader-clk ticks / op, %7.2f ns / op\n", name, threads, h[0] / ops, h[1] * 10.0 / ops);
    hipFree(out); hipFree(t);
}
int main() {
    for (int threads : {64, 256}) {
        run<0>("dependent v_fma_f64", threads, 16);
        run<1>("4 independent v_fma_f64 chains (per fma)", threads, 16);
        run<4>("dependent v_mul_f64", threads, 16);
        run<2>("dependent v_rsq_f64 + add", threads, 16);
        run<5>("dependent v_fma_f32", threads, 16);
        run<3>("dependent LDS read", threads, 16);
        run<6>("LDS write -> other lane's read", threads, 16);
        run<7>("s_barrier (lgkmcnt)", threads, 16);
    }
    return 0;
}
