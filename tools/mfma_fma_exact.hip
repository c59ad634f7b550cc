// Is the fp32 MFMA's multiply-add the same IEEE operation as v_fma_f32?  (Question behind DESIGN.md section 4 (c): could the 3x3 / 3x4
// contractions of Stage B - fma chains that must match the reference's torch.matmul bit for bit - move to the matrix cores?)
// v_mfma_f32_4x4x1_16b_f32: 16 blocks of a 4x4 outer product, D[i][j] = A[i] * B[j] + C[i][j]; lane l = 4 * block + j supplies A[l % 4]
// and B[l % 4] of its block and receives D[0..3][l % 4].  So D_v(lane) = A(lane 4*(l/4) + v) * B(l) + C_v(l): compared here with fmaf on
// operands drawn from the ranges of the path (coefficients 1e-7 .. 1e3, pixel coordinates 0 .. 2047, accumulators up to 1e6), plus
// signed zeros and values that cancel.
// build: hipcc -O2 --offload-arch=gfx950 -ffp-contract=off tools/mfma_fma_exact.hip -o tools/bin/mfma_fma_exact
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ uint32_t rng(uint32_t &s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }
__device__ float draw(uint32_t &s, int kind)
{
    const uint32_t r = rng(s);
    const float u = (float)(r >> 8) * (1.0f / 16777216.0f);
    switch (kind) {
    case 0: return (float)(r % 2048u);                                   // pixel coordinate
    case 1: { const float e = (float)((rng(s) % 34u)) - 23.0f; return ldexpf(1.0f + u, (int)e) * ((r & 1u) ? -1.0f : 1.0f); }   // coefficient
    case 2: return (u - 0.5f) * 2.0e6f;                                  // accumulator
    default: return (r & 3u) == 0 ? 0.0f : ((r & 3u) == 1 ? -0.0f : u);
    }
}

__global__ void k(unsigned long long *bad, unsigned long long *total, int iters, int ka, int kb, int kc, int cancel)
{
    uint32_t s = 0x9E3779B9u * (blockIdx.x * blockDim.x + threadIdx.x + 1u);
    unsigned long long nb = 0, nt = 0;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        const float a = draw(s, ka), b = draw(s, kb);
        f4 c = { draw(s, kc), draw(s, kc), draw(s, kc), draw(s, kc) };
        float av[4];
        for (int v = 0; v < 4; ++v) av[v] = __shfl(a, (lane & ~3) + v);
        if (cancel) for (int v = 0; v < 4; ++v) c[v] = -(av[v] * b) * (1.0f + (float)((int)(rng(s) % 5u) - 2) * 5.9604645e-8f);   // near-cancellation
        const f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
        for (int v = 0; v < 4; ++v) {
            const float e = fmaf(av[v], b, c[v]);
            nb += (__float_as_uint(e) != __float_as_uint(d[v]));
            ++nt;
        }
    }
    atomicAdd(bad, nb);
    atomicAdd(total, nt);
}

int main()
{
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    const char *names[] = { "pixel", "coefficient", "accumulator", "zeros/unit" };
    for (int cancel = 0; cancel < 2; ++cancel)
        for (int ka = 1; ka < 4; ka += 2)
            for (int kb = 0; kb < 4; ++kb)
                for (int kc = 1; kc < 4; ++kc) {
                    hipMemset(d, 0, 16);
                    hipLaunchKernelGGL(k, dim3(1024), dim3(256), 0, 0, d, d + 1, 400, ka, kb, kc, cancel);
                    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                    printf("A %-12s B %-12s C %-12s%s: %llu of %llu results differ from fmaf\n", names[ka], names[kb], names[kc], cancel ? " (C ~ -A*B)" : "", h[0], h[1]);
                }
    return 0;
}
