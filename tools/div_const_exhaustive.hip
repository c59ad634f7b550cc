// Exhaustive check of the 3-operation division by a CONSTANT divisor b with y = RN(1/b) precomputed (Markstein's sequence:
// q = RN(a*y); r = a - b*q exactly (fma); q' = RN(q + r*y)) against the correctly rounded quotient, computed in double and
// rounded to float (a/b in double is within 2^-53 relative of the exact quotient; a double-rounding tie would need the exact
// quotient within 2^-29 ulp of a float midpoint, which a quotient of two 24-bit significands cannot be unless it IS exact).
// Also checks the 5-operation form the kernels used before.  Divisors: k/2 for k = 2 .. 16384 (every half-width / half-height
// an image can have), numerators: every float with |a| in 2^-30 .. 2^40 for the listed sizes, 2^-8 .. 2^24 for all others.
//   build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/div_const_exhaustive.hip -o tools/bin/div_const_exhaustive
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ float div3(float a, float b, float y)
{
    float q = a * y;
    float r = fmaf(-b, q, a);
    return fmaf(r, y, q);
}
__device__ __forceinline__ float div5(float a, float b, float y)
{
    float q = a * y;
    float e = fmaf(-b, q, a);
    q = fmaf(e, y, q);
    e = fmaf(-b, q, a);
    return fmaf(e, y, q);
}

__global__ void k_check(float b, float y, int elo, int ehi, unsigned long long *bad3, unsigned long long *bad5, uint32_t *example)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t lo = (uint64_t)elo << 23, hi = (uint64_t)ehi << 23;
    unsigned long long n3 = 0, n5 = 0;
    for (uint64_t i = lo + blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += stride) {
        for (uint32_t sgn = 0; sgn < 2; ++sgn) {
            const float a = __uint_as_float((uint32_t)i | (sgn << 31));
            const float want = (float)((double)a / (double)b);
            if (__float_as_uint(div3(a, b, y)) != __float_as_uint(want)) { ++n3; *example = __float_as_uint(a); }
            if (__float_as_uint(div5(a, b, y)) != __float_as_uint(want)) ++n5;
        }
    }
    if (n3) atomicAdd(bad3, n3);
    if (n5) atomicAdd(bad5, n5);
}

int main()
{
    unsigned long long *d, h[2];
    uint32_t *ex, hex = 0;
    hipMalloc(&d, 16); hipMalloc(&ex, 4);
    hipMemset(d, 0, 16); hipMemset(ex, 0, 4);
    const int wide[] = {512, 640, 960, 1280, 384, 1024, 1536, 1920, 1080, 1242, 375, 48, 64, 96, 128, 3, 5, 7, 4095, 8191, 16383};
    int rc = 0;
    unsigned long long checked = 0;
    for (int k = 2; k <= 16384; ++k) {
        const float b = 0.5f * (float)k;
        const float y = (float)(1.0 / (double)b);        // RN(1/b): 1/b in double is far from a float midpoint for these b
        bool w = false;
        for (int v : wide) w |= (v == k);
        const int elo = 127 + (w ? -30 : -8), ehi = 127 + (w ? 40 : 24);
        hipLaunchKernelGGL(k_check, dim3(w ? 4096 : 1024), dim3(256), 0, 0, b, y, elo, ehi, d, d + 1, ex);
        checked += 2ull * (unsigned long long)(ehi - elo) << 23;
        if (w || k % 2048 == 0) {
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            hipMemcpy(&hex, ex, 4, hipMemcpyDeviceToHost);
            printf("divisors 1 .. %6.1f  numerators checked %.3e  mismatches 3-op %llu (example %08x)  5-op %llu\n", b, (double)checked, h[0], hex, h[1]);
            fflush(stdout);
        }
    }
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("TOTAL %.3e quotients: 3-op mismatches %llu, 5-op mismatches %llu\n", (double)checked, h[0], h[1]);
    rc = h[0] ? 1 : 0;
    return rc;
}
