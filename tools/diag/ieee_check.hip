// Exhaustive check (all 2^32 float bit patterns) of lean correctly-rounded reciprocal / square root sequences against the
// compiler's IEEE expansions (-fhip-fp32-correctly-rounded-divide-sqrt).  Build: hipcc -O3 --offload-arch=gfx950
// -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -o ieee_check ieee_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float lean_rcp(float x)
{
    float y = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}

template <int STEPS>
__device__ __forceinline__ float lean_sqrt(float x)
{
    const float r = __builtin_amdgcn_rsqf(x);
    float g = x * r;
    const float h = 0.5f * r;
    for (int i = 0; i < STEPS; ++i) {
        const float e = __builtin_fmaf(-g, g, x);
        g = __builtin_fmaf(e, h, g);
    }
    return g;
}

__device__ __forceinline__ float sqrt_hw_fix(float x)
{
    // hardware sqrt (1 ulp) + residual-based choice among s-1ulp, s, s+1ulp without the denormal scaling
    float s = __builtin_amdgcn_sqrtf(x);
    const float lo = __int_as_float(__float_as_int(s) - 1), hi = __int_as_float(__float_as_int(s) + 1);
    const float rl = __builtin_fmaf(-lo, s, x), rh = __builtin_fmaf(-hi, s, x);
    s = rl <= 0.0f ? lo : s;
    s = rh > 0.0f ? hi : s;
    return s;
}

__device__ __forceinline__ bool same(float a, float b)
{
    return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
}

// counts[k]: mismatches of candidate k; first[k]: lowest mismatching bit pattern; range lo/hi of mismatching |x|
__global__ void check(unsigned long long* counts, unsigned* minabs, unsigned* maxabs)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long local[4] = { 0, 0, 0, 0 };
    unsigned mn[4] = { 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu }, mx[4] = { 0, 0, 0, 0 };
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((unsigned)i);
        const unsigned ax = (unsigned)i & 0x7fffffffu;
        const float ref_r = 1.0f / x, ref_s = sqrtf(x);
        const float c[4] = { lean_rcp(x), lean_sqrt<1>(x), lean_sqrt<2>(x), sqrt_hw_fix(x) };
        const float ref[4] = { ref_r, ref_s, ref_s, ref_s };
        for (int k = 0; k < 4; ++k) {
            if (!same(c[k], ref[k])) {
                ++local[k];
                mn[k] = ax < mn[k] ? ax : mn[k];
                mx[k] = ax > mx[k] ? ax : mx[k];
            }
        }
    }
    for (int k = 0; k < 4; ++k) {
        if (local[k]) {
            atomicAdd(&counts[k], local[k]);
            atomicMin(&minabs[k], mn[k]);
            atomicMax(&maxabs[k], mx[k]);
        }
    }
}

// mismatches restricted to x in [lo, hi] (positive normal range of interest)
__global__ void check_range(unsigned lo_bits, unsigned hi_bits, unsigned long long* counts, unsigned* first)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)lo_bits + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i <= hi_bits; i += stride) {
        const float x = __uint_as_float((unsigned)i);
        const float c[4] = { lean_rcp(x), lean_sqrt<1>(x), lean_sqrt<2>(x), sqrt_hw_fix(x) };
        const float ref[4] = { 1.0f / x, sqrtf(x), sqrtf(x), sqrtf(x) };
        for (int k = 0; k < 4; ++k)
            if (!same(c[k], ref[k])) { atomicAdd(&counts[k], 1ull); atomicMin(&first[k], (unsigned)i); }
        // negative x for the reciprocal
        const float nx = -x;
        if (!same(lean_rcp(nx), 1.0f / nx)) { atomicAdd(&counts[0], 1ull); atomicMin(&first[0], (unsigned)i); }
    }
}

int main()
{
    unsigned long long* counts;
    unsigned *mn, *mx;
    hipMalloc(&counts, 4 * sizeof(unsigned long long));
    hipMalloc(&mn, 4 * sizeof(unsigned));
    hipMalloc(&mx, 4 * sizeof(unsigned));
    hipMemset(counts, 0, 4 * sizeof(unsigned long long));
    hipMemset(mn, 0xff, 4 * sizeof(unsigned));
    hipMemset(mx, 0, 4 * sizeof(unsigned));
    check<<<256 * 8, 256>>>(counts, mn, mx);
    unsigned long long h[4];
    unsigned hmn[4], hmx[4];
    hipMemcpy(h, counts, sizeof h, hipMemcpyDeviceToHost);
    hipMemcpy(hmn, mn, sizeof hmn, hipMemcpyDeviceToHost);
    hipMemcpy(hmx, mx, sizeof hmx, hipMemcpyDeviceToHost);
    const char* names[4] = { "lean_rcp", "lean_sqrt<1>", "lean_sqrt<2>", "sqrt_hw_fix" };
    for (int k = 0; k < 4; ++k) {
        float a, b;
        std::memcpy(&a, &hmn[k], 4);
        std::memcpy(&b, &hmx[k], 4);
        std::printf("all inputs   %-13s mismatches %llu  |x| range of mismatches [%g (0x%08x), %g (0x%08x)]\n", names[k], h[k], a, hmn[k], b, hmx[k]);
    }
    // normal range 2^-60 .. 2^60
    const float lo = 8.67361738e-19f, hi = 1.15292150e18f;
    unsigned lob, hib;
    std::memcpy(&lob, &lo, 4);
    std::memcpy(&hib, &hi, 4);
    hipMemset(counts, 0, 4 * sizeof(unsigned long long));
    hipMemset(mn, 0xff, 4 * sizeof(unsigned));
    check_range<<<256 * 8, 256>>>(lob, hib, counts, mn);
    hipMemcpy(h, counts, sizeof h, hipMemcpyDeviceToHost);
    hipMemcpy(hmn, mn, sizeof hmn, hipMemcpyDeviceToHost);
    for (int k = 0; k < 4; ++k) std::printf("2^-60..2^60  %-13s mismatches %llu  first 0x%08x\n", names[k], h[k], hmn[k]);
    return 0;
}
