// Self-test of the lane-exchange primitives used by the reductions (DPP / v_permlane*_swap instead of ds_bpermute):
// every variant must equal __shfl_xor for its mask.  hipcc --offload-arch=gfx950 tools/shfl_selftest.hip -o /tmp/st && /tmp/st
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
template <int M> __device__ __forceinline__ unsigned xor_lane_u32(unsigned v, int lane, int variant) {
    if constexpr (M == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);
    else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);
    else if constexpr (M == 4) {
        if (variant == 0) { unsigned d = __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0xA, false); return __builtin_amdgcn_update_dpp(d, v, 0x12C, 0xF, 0x5, false); }
        else { unsigned d = __builtin_amdgcn_update_dpp(v, v, 0x124, 0xF, 0x5, false); return __builtin_amdgcn_update_dpp(d, v, 0x12C, 0xF, 0xA, false); }
    }
    else if constexpr (M == 8) return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);
    else if constexpr (M == 16) { v2u r = __builtin_amdgcn_permlane16_swap(v, v, false, false); return variant == 0 ? ((lane & 16) ? r.x : r.y) : ((lane & 16) ? r.y : r.x); }
    else { v2u r = __builtin_amdgcn_permlane32_swap(v, v, false, false); return variant == 0 ? ((lane & 32) ? r.x : r.y) : ((lane & 32) ? r.y : r.x); }
}
template <int M> __device__ void check(unsigned* out, int slot) {
    const int lane = threadIdx.x;
    const unsigned v = 1000u + 7u * lane;
    const unsigned want = __shfl_xor(v, M, 64);
    for (int variant = 0; variant < 2; ++variant) {
        const unsigned got = xor_lane_u32<M>(v, lane, variant);
        const unsigned long long ok = __ballot(got == want);
        if (lane == 0) out[2 * slot + variant] = (ok == ~0ull) ? 1u : 0u;
    }
}
// The select-free exchange of the halving reduce-scatters (ba_kernels.hip, halves_exchange_sum): after the swap / the two bank-masked
// DPP moves, register X + register Y must be (up ? hi : lo) + the partner's (up ? hi : lo).
template <int M> __device__ void check_halves(unsigned* out, int slot) {
    const int lane = threadIdx.x;
    const unsigned lo = 1000u + 7u * lane, hi = 500000u + 11u * lane;
    const bool up = (lane & M) != 0;
    const unsigned plo = __shfl_xor(lo, M, 64), phi = __shfl_xor(hi, M, 64);
    const unsigned want = up ? hi + phi : lo + plo;
    unsigned x, y;
    if constexpr (M == 32) { v2u r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false); x = r.x; y = r.y; }
    else if constexpr (M == 16) { v2u r = __builtin_amdgcn_permlane16_swap(lo, hi, false, false); x = r.x; y = r.y; }
    else if constexpr (M == 8) { x = __builtin_amdgcn_update_dpp(lo, hi, 0x128, 0xF, 0xC, false); y = __builtin_amdgcn_update_dpp(hi, lo, 0x128, 0xF, 0x3, false); }
    else { x = __builtin_amdgcn_update_dpp(lo, hi, 0x124, 0xF, 0xA, false); y = __builtin_amdgcn_update_dpp(hi, lo, 0x12C, 0xF, 0x5, false); }
    const unsigned long long ok = __ballot(x + y == want);
    if (lane == 0) out[slot] = (ok == ~0ull) ? 1u : 0u;
}
__global__ void k(unsigned* out) {
    check_halves<4>(out, 12); check_halves<8>(out, 13); check_halves<16>(out, 14); check_halves<32>(out, 15); check<1>(out, 0); check<2>(out, 1); check<4>(out, 2); check<8>(out, 3); check<16>(out, 4); check<32>(out, 5); }
int main() {
    unsigned* d; unsigned h[16];
    hipMalloc(&d, sizeof(h)); hipMemset(d, 0, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const int masks[6] = { 1, 2, 4, 8, 16, 32 };
    for (int i = 0; i < 6; ++i) std::printf("xor %2d: variant0 %s  variant1 %s\n", masks[i], h[2 * i] ? "OK" : "--", h[2 * i + 1] ? "OK" : "--");
    bool all = true;
    for (int i = 0; i < 4; ++i) { std::printf("halves exchange, mask %2d: %s\n", masks[2 + i], h[12 + i] ? "OK" : "--"); all = all && h[12 + i]; }
    for (int i = 0; i < 6; ++i) all = all && h[2 * i];
    return all ? 0 : 1;
}
