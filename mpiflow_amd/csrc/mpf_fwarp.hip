// mpf_fwarp.hip - order-preserving parallel forward splat for MI355X (gfx950) + the glue kernels of moving_obj.py.
//
// The reference's external/forward_warping/warping.c:6-33 is a serial raster scan whose result depends on visit
// order: at a target pixel the colour that survives is NOT the z-minimum but that of the LAST source (in raster order)
// whose z is smaller than the z of the source that visited the target immediately before it (1000 for the first);
// the collision byte records whether the last visitor found the target untouched (or left at the 1000 sentinel).
// A z-buffer with atomics therefore gives different images (SURVEY.md §7 hard part 3).  Parallel restatement used here:
//
//   1. key[i] = target(i) = idy[i]*w + idx[i] for every source pixel i (raster index)
//   2. stable LSD radix sort of (key, i) by key          -> each target's visitors, contiguous, in raster order
//   3. per sorted slot j: pred = slot j-1 if same target;  cond(j) = z[j] < (pred ? z[pred] : 1000)
//      winner(target) = max j with cond(j)                (atomicMax on a per-target word)
//   4. the last slot of each segment writes the 5 output bytes of its target
//
// Round 2 (189 us in 14 launches -> 53 us in 4 at 640 x 960, profiles/r2/forward_warp_kernels.txt):
//   * images up to 2^22 pixels: ONE stable pass on the HIGH bits of the target (keys + histogram fused, a per-digit column scan
//     instead of a single-workgroup scan over all counters - that scan alone was 34 us per pass -, stable scatter), then one
//     workgroup per bucket of 2^lb consecutive targets finishes the sort in LDS and resolves its targets (k_fw_bucket): no
//     second histogram / scan / scatter, no global winner array, no separate fill, mark and write launches;
//   * larger images: the general path - two or three passes with 8..11-bit digits, mark, write.
// Round 5 (5 launches -> 3 for the moving-object chain, no workgroup barrier left in the splat): the sort became a GATHER.  Targets come from a
// projection of raster-ordered sources, so the visitors of 256 consecutive targets sit in a handful of 64-source slabs; pass 1 records the
// [min, max] target of every slab and of every 1024-source tile beside the keys, and ONE WAVE per bucket of 256 targets collects its visitors
// from the slabs whose range touches the bucket - in raster order, so stability is free -, counting-sorts them by target in LDS and resolves
// them (k_fw_gather_resolve).  No histogram table, no column scan, no scatter, no second key / value arrays; pile-ups of any size are
// streamed through the wave's LDS in chunks that carry (last z, winner, collision) per target.  The round-2 path stays as fwarp_path 2.
// Integer/byte work, bandwidth-trivial (N = h*w <= a few million 4-byte keys, 2-3 radix passes): the design goal is
// bit-exact equality with the serial C, with bounded cost for pathological pile-ups (thousands of sources clamped onto
// one border pixel), which is what the global sort buys over per-target lists.
#include <string.h>
#include "mpf_common.h"
#include "mpf_math.h"

static int g_fw_prio = 0;       // mpf_tune("chain_prio", 0..3): s_setprio of the sort / resolve / mask kernels (they run underneath a pair launch)
#define MPF_FW_SETPRIO(p) do { if ((p) == 1) __builtin_amdgcn_s_setprio(1); else if ((p) == 2) __builtin_amdgcn_s_setprio(2); else if ((p) == 3) __builtin_amdgcn_s_setprio(3); } while (0)

#define RADIX_BITS_MAX 11              // digits of 8..11 bits: 20-bit keys (640 x 960 targets) sort in two passes, up to 33 bits in three
#define RADIX_MAX (1 << RADIX_BITS_MAX)
#define SORT_THREADS 256
#define SORT_ITEMS 8
#define SORT_TILE (SORT_THREADS * SORT_ITEMS)

// Device-side choice between the gather path and the radix path for caller-supplied targets (mpf_forward_warp / forward_warping), without a host round trip:
// pass 1 of the gather path adds up the bucket visits its slab ranges imply (work); every kernel of BOTH paths is launched, and each one returns at once when the
// other path owns the call.  work == nullptr: ungated (the moving-object chain, whose targets come from a projection and are parallax-bounded).
struct FwGate { const unsigned long long *work; unsigned long long thr; bool radix; };
__device__ __forceinline__ bool fw_gate_closed(const FwGate g) { return g.work && ((*g.work > g.thr) != g.radix); }

// ---- moving_obj.py:29-30 : depth = 1/(disp + 0.005), clamped to 100 --------------------------------------------------

// `p1.cpu().long()` (moving_obj.py:121) is an x86 cvttss2si: values that do not fit an int64 - NaN, +-inf, |v| >= 2^63, reachable
// when q.z + 1e-7 comes near 0 in Project3D - become INT64_MIN, which the following clamp turns into 0.  The GPU's conversion
// saturates instead (+huge -> INT64_MAX -> clamped to w-1), so the out-of-range case is spelled out.
__device__ __forceinline__ int64_t mpf_trunc_like_x86(float v)
{
    return (v >= -9223372036854775808.0f && v < 9223372036854775808.0f) ? (int64_t)v : INT64_MIN;
}


__global__ void __launch_bounds__(256)
k_disp_to_depth(const float *__restrict__ disp, int64_t N, float *__restrict__ depth)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float d = 1.0f / (disp[n] + 0.005f);
    depth[n] = (d > 100.0f) ? 100.0f : d;
}

extern "C" int mpf_disp_to_depth(const float *d_disp, int64_t N, float *d_depth, void *stream)
{
    MPF_REQUIRE(d_disp && d_depth && N >= 1, "mpf_disp_to_depth: bad argument");
    hipLaunchKernelGGL(k_disp_to_depth, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_disp, N, d_depth);
    return mpf_launch_status("k_disp_to_depth");
}

// ---- moving_obj.py:108-124, :153 --------------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
k_select_truncate(const float *__restrict__ p_static, const float *__restrict__ z_static, const float *__restrict__ p_obj,
                  const float *__restrict__ z_obj, const float *__restrict__ inst, int H, int W, float *__restrict__ p1,
                  float *__restrict__ z1, int64_t *__restrict__ safe_x, int64_t *__restrict__ safe_y, float *__restrict__ flow01)
{
    const int64_t N = (int64_t)H * W;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const bool sel = inst[n] > 0.0f;                                    // :108-112
    const float nx = sel ? p_obj[2 * n] : p_static[2 * n];
    const float ny = sel ? p_obj[2 * n + 1] : p_static[2 * n + 1];
    z1[n] = sel ? z_obj[n] : z_static[n];
    const float px = (nx + 1.0f) / 2.0f * (float)(W - 1);               // :115-117
    const float py = (ny + 1.0f) / 2.0f * (float)(H - 1);
    p1[2 * n] = px; p1[2 * n + 1] = py;
    int64_t tx = mpf_trunc_like_x86(px), ty = mpf_trunc_like_x86(py);   // .long() truncates toward zero, :121-122
    tx = tx > W - 1 ? W - 1 : tx; tx = tx < 0 ? 0 : tx;
    ty = ty > H - 1 ? H - 1 : ty; ty = ty < 0 ? 0 : ty;
    safe_x[n] = tx; safe_y[n] = ty;
    flow01[2 * n] = px - (float)(n % W);                                // :153
    flow01[2 * n + 1] = py - (float)(n / W);
}

extern "C" int mpf_select_truncate(const float *d_p_static, const float *d_z_static, const float *d_p_obj, const float *d_z_obj,
                                   const float *d_inst, int H, int W, float *d_p1, float *d_z1, int64_t *d_safe_x,
                                   int64_t *d_safe_y, float *d_flow01, void *stream)
{
    MPF_REQUIRE(d_p_static && d_z_static && d_p_obj && d_z_obj && d_inst && d_p1 && d_z1 && d_safe_x && d_safe_y && d_flow01 &&
                    H >= 1 && W >= 1, "mpf_select_truncate: bad argument");
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_select_truncate, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_p_static,
                       d_z_static, d_p_obj, d_z_obj, d_inst, H, W, d_p1, d_z1, d_safe_x, d_safe_y, d_flow01);
    return mpf_launch_status("k_select_truncate");
}

// ---- moving_obj.py:29-124 in one pass: depth from disparity, both projections, instance select, truncation -----------

struct MpfMoProj { float ik[9]; float Ps[12]; float Po[12]; };

MPF_DEV void mpf_project_point(const float *P, float X, float Y, float Z, int H, int W, float &nx, float &ny, float &z)
{
    float q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = mpf_row4_xyz1(P[4 * c], P[4 * c + 1], P[4 * c + 2], P[4 * c + 3], X, Y, Z);
    const float den = q[2] + 1e-7f;                                     // geometry.py:70
    float px = q[0] / den, py = q[1] / den;
    px = px / (float)(W - 1);                                           // geometry.py:73-74
    py = py / (float)(H - 1);
    nx = (px - 0.5f) * 2.0f;                                            // geometry.py:75
    ny = (py - 0.5f) * 2.0f;
    z = q[2];
}

// One source pixel n of moving_obj.py:29-124, :153: depth from disparity, back-projection, the static or the object projection, pixel
// units, truncation + clamp, flow.  Shared by the stand-alone kernel and by the chain's fused first sort pass (same IEEE op sequence).
struct MpfMoOut { float *p1, *z1; int64_t *safe_x, *safe_y; float *flow01; };

MPF_DEV uint32_t mpf_mo_pixel_v(const float disp_n, const MpfMoProj &m, const float inst_n, int H, int W, const int64_t n, const MpfMoOut &o);

MPF_DEV uint32_t mpf_mo_pixel(const float *__restrict__ disp, const MpfMoProj &m, const float *__restrict__ inst, int H, int W, const int64_t n,
                              const MpfMoOut &o)
{
    return mpf_mo_pixel_v(disp[n], m, inst[n], H, W, n, o);
}

// the same with the two inputs of pixel n already loaded (pass 1 of the gather path loads its four pixels' inputs before the first store)
MPF_DEV uint32_t mpf_mo_pixel_v(const float disp_n, const MpfMoProj &m, const float inst_n, int H, int W, const int64_t n, const MpfMoOut &o)
{
    const float fx = (float)(n % W), fy = (float)(n / W);
    float dep = 1.0f / (disp_n + 0.005f);                               // moving_obj.py:29-30
    dep = (dep > 100.0f) ? 100.0f : dep;
    float cam[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) cam[c] = dep * mpf_row3_xy1(m.ik[3 * c], m.ik[3 * c + 1], m.ik[3 * c + 2], fx, fy);   // geometry.py:42-43
    const bool sel = inst_n > 0.0f;                                     // moving_obj.py:108-112
    float nx, ny, z;
    if (sel) mpf_project_point(m.Po, cam[0], cam[1], cam[2], H, W, nx, ny, z);
    else     mpf_project_point(m.Ps, cam[0], cam[1], cam[2], H, W, nx, ny, z);
    o.z1[n] = z;
    const float px = (nx + 1.0f) / 2.0f * (float)(W - 1);               // :115-117
    const float py = (ny + 1.0f) / 2.0f * (float)(H - 1);
    o.p1[2 * n] = px; o.p1[2 * n + 1] = py;
    int64_t tx = mpf_trunc_like_x86(px), ty = mpf_trunc_like_x86(py);   // :121-122
    tx = tx > W - 1 ? W - 1 : tx; tx = tx < 0 ? 0 : tx;
    ty = ty > H - 1 ? H - 1 : ty; ty = ty < 0 ? 0 : ty;
    o.safe_x[n] = tx; o.safe_y[n] = ty;
    o.flow01[2 * n] = px - fx;                                          // :153
    o.flow01[2 * n + 1] = py - fy;
    return (uint32_t)(ty * W + tx);                                     // the forward warp's target (warping.c:15-16)
}

__global__ void __launch_bounds__(256)
k_moving_object_project(const float *__restrict__ disp, MpfMoProj m, const float *__restrict__ inst, int H, int W, MpfMoOut o)
{
    const int64_t N = (int64_t)H * W;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    (void)mpf_mo_pixel(disp, m, inst, H, W, n, o);
}

extern "C" int mpf_moving_object_project(const float *d_disp, const float *h_inv_k9, const float *h_P_static12, const float *h_P_obj12,
                                         const float *d_inst, int H, int W, float *d_p1, float *d_z1, int64_t *d_safe_x,
                                         int64_t *d_safe_y, float *d_flow01, void *stream)
{
    MPF_REQUIRE(d_disp && h_inv_k9 && h_P_static12 && h_P_obj12 && d_inst && d_p1 && d_z1 && d_safe_x && d_safe_y && d_flow01 &&
                    H >= 1 && W >= 1, "mpf_moving_object_project: bad argument");
    MpfMoProj m;
    memcpy(m.ik, h_inv_k9, sizeof(m.ik));
    memcpy(m.Ps, h_P_static12, sizeof(m.Ps));
    memcpy(m.Po, h_P_obj12, sizeof(m.Po));
    const int64_t N = (int64_t)H * W;
    const MpfMoOut o = { d_p1, d_z1, d_safe_x, d_safe_y, d_flow01 };
    hipLaunchKernelGGL(k_moving_object_project, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_disp, m, d_inst, H, W, o);
    return mpf_launch_status("k_moving_object_project");
}

// ---- sort -------------------------------------------------------------------------------------------------------

// Also clears the per-target winner word and, for the device entry point, the output image (unvisited targets read 0:
// moving_obj.py:123 zero-initialises `warped_arr`) - one pass over N instead of two separate fill launches.
__global__ void __launch_bounds__(256)
k_fw_keys(const int64_t *__restrict__ idx, const int64_t *__restrict__ idy, int h, int w, uint32_t *__restrict__ keys,
          uint32_t *__restrict__ vals, uint32_t *__restrict__ win, uint8_t *__restrict__ warped_to_clear, const FwGate gate = FwGate{nullptr, 0, true})
{
    if (fw_gate_closed(gate)) return;
    const int64_t N = (int64_t)h * w;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    if (win) win[n] = 0u;
    if (warped_to_clear) {
#pragma unroll
        for (int k = 0; k < 5; ++k) warped_to_clear[5 * n + k] = 0;
    }
    int64_t x = idx[n], y = idy[n];
    // the reference does no bounds check (caller pre-clamps, moving_obj.py:121-122); clamp instead of scribbling
    x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
    y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
    keys[n] = (uint32_t)(y * w + x);
    vals[n] = (uint32_t)n;
}

template <int BITS>
__global__ void __launch_bounds__(SORT_THREADS)
k_radix_hist(const uint32_t *__restrict__ keys, uint32_t N, int shift, uint32_t nb, uint32_t *__restrict__ hist, const FwGate gate = FwGate{nullptr, 0, true})
{
    constexpr int RADIX = 1 << BITS, DPT = RADIX / SORT_THREADS;
    __shared__ uint32_t h[RADIX];
    if (fw_gate_closed(gate)) return;
#pragma unroll
    for (int k = 0; k < DPT; ++k) h[threadIdx.x + k * SORT_THREADS] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * SORT_TILE;
#pragma unroll
    for (int it = 0; it < SORT_ITEMS; ++it) {
        const uint32_t i = base + it * SORT_THREADS + threadIdx.x;
        if (i < N) atomicAdd(&h[(keys[i] >> shift) & (RADIX - 1)], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DPT; ++k) {                            // tile-major table [tile][digit]: coalesced here and in the scatter
        const uint32_t d = threadIdx.x + k * SORT_THREADS;
        hist[(size_t)blockIdx.x * RADIX + d] = h[d];
    }
}

#ifdef MPF_WITNESS            // round 2's one-pass sort + per-bucket workgroups (mpf_tune("fwarp_path", 2)): a second witness of the splat, witness build only
// The first pass of the fast path: keys and the histogram of their high digit in one sweep over the sources (tile = SORT_TILE keys)
template <int BITS>
__global__ void __launch_bounds__(SORT_THREADS)
k_fw_keys_hist(const int64_t *__restrict__ idx, const int64_t *__restrict__ idy, int h, int w, uint32_t *__restrict__ keys,
               uint32_t *__restrict__ vals, uint32_t N, int shift, uint32_t *__restrict__ hist)
{
    constexpr int RADIX = 1 << BITS, DPT = RADIX / SORT_THREADS;
    __shared__ uint32_t hh[RADIX];
    const uint32_t nb = (N + SORT_TILE - 1) / SORT_TILE;
    for (uint32_t tile = blockIdx.x; tile < nb; tile += gridDim.x) {      // a workgroup walks several tiles when the grid is capped (mpf_tune("chain_grid"))
#pragma unroll
        for (int k = 0; k < DPT; ++k) hh[threadIdx.x + k * SORT_THREADS] = 0;
        __syncthreads();
        const uint32_t base = tile * SORT_TILE;
#pragma unroll
        for (int it = 0; it < SORT_ITEMS; ++it) {
            const uint32_t n = base + it * SORT_THREADS + threadIdx.x;
            if (n < N) {
                int64_t x = idx[n], y = idy[n];
                x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);              // as k_fw_keys: clamp instead of scribbling
                y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
                const uint32_t key = (uint32_t)(y * w + x);
                keys[n] = key;
                vals[n] = n;
                atomicAdd(&hh[(key >> shift) & (RADIX - 1)], 1u);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
            const uint32_t d = threadIdx.x + k * SORT_THREADS;
            hist[(size_t)tile * RADIX + d] = hh[d];
        }
    }
}

// The chain's first pass (mpf_moving_object_chain): the projection of moving_obj.py:29-124 computed HERE, per source pixel, instead of
// by a kernel of its own whose int64 targets this pass would read back - one launch and 16 N bytes of reads less.
template <int BITS>
__global__ void __launch_bounds__(SORT_THREADS)
k_mo_project_keys_hist(const float *__restrict__ disp, const MpfMoProj m, const float *__restrict__ inst, int h, int w, const MpfMoOut o,
                       uint32_t *__restrict__ keys, uint32_t *__restrict__ vals, uint32_t N, int shift, uint32_t *__restrict__ hist, const int prio)
{
    constexpr int RADIX = 1 << BITS, DPT = RADIX / SORT_THREADS;
    __shared__ uint32_t hh[RADIX];
    MPF_FW_SETPRIO(prio);
    const uint32_t nb = (N + SORT_TILE - 1) / SORT_TILE;
    for (uint32_t tile = blockIdx.x; tile < nb; tile += gridDim.x) {
#pragma unroll
        for (int k = 0; k < DPT; ++k) hh[threadIdx.x + k * SORT_THREADS] = 0;
        __syncthreads();
        const uint32_t base = tile * SORT_TILE;
        for (int it = 0; it < SORT_ITEMS; ++it) {
            const uint32_t n = base + it * SORT_THREADS + threadIdx.x;
            if (n < N) {
                const uint32_t key = mpf_mo_pixel(disp, m, inst, h, w, (int64_t)n, o);
                keys[n] = key;
                vals[n] = n;
                atomicAdd(&hh[(key >> shift) & (RADIX - 1)], 1u);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
            const uint32_t d = threadIdx.x + k * SORT_THREADS;
            hist[(size_t)tile * RADIX + d] = hh[d];
        }
    }
}

#endif   // MPF_WITNESS

// Offsets of the scatter, two cheap steps instead of one scan over all RADIX*nb counters (which took a single workgroup 34 us
// per pass - half of the whole forward warp): the exclusive offset of (digit d, tile b) is
//     base[d] + colprefix[b][d],   base[d] = sum of the totals of the digits below d,   colprefix[b][d] = sum over tiles < b of hist[.][d]
// k_radix_colscan: one wave per digit walks its column of the tile-major table (the only strided access of the sort), turns it into
//                  colprefix in place and writes the digit's total;
// k_radix_scatter: every workgroup rebuilds base[] from the RADIX totals in LDS (256 .. 2048 numbers) and reads its own row of
//                  colprefix with coalesced loads.
__global__ void __launch_bounds__(256)
k_radix_colscan(uint32_t *__restrict__ hist, uint32_t nb, uint32_t radix, uint32_t *__restrict__ totals, const int prio = 0, const FwGate gate = FwGate{nullptr, 0, true})
{
    if (fw_gate_closed(gate)) return;
    MPF_FW_SETPRIO(prio);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    for (uint32_t c = blockIdx.x * wpb + wave; c < radix; c += gridDim.x * wpb) {       // one wave per column
        uint32_t *col = hist + c;
        uint32_t carry = 0;
        for (uint32_t b0 = 0; b0 < nb; b0 += 64) {
            const uint32_t i = b0 + lane;
            const uint32_t v = i < nb ? col[(size_t)i * radix] : 0u;
            uint32_t inc = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(inc, off);
                if (lane >= (uint32_t)off) inc += o;
            }
            if (i < nb) col[(size_t)i * radix] = carry + inc - v;
            carry += __shfl(inc, 63);
        }
        if (lane == 0) totals[c] = carry;
    }
}

// Stable scatter of one tile.  Keys are visited in index order: iteration `it` covers 256 consecutive keys, wave k of
// the workgroup the k-th 64 of them, lane l the l-th.  rank-in-wave comes from BITS ballots (one per digit bit), waves
// are ordered through per-wave digit counts in LDS, iterations through a running per-digit cursor.
template <int BITS>
__global__ void __launch_bounds__(SORT_THREADS)
k_radix_scatter(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint32_t *__restrict__ keys_out,
                uint32_t *__restrict__ vals_out, uint32_t N, int shift, uint32_t nb, const uint32_t *__restrict__ offsets,
                const uint32_t *__restrict__ totals, const int prio = 0, const FwGate gate = FwGate{nullptr, 0, true})
{
    constexpr int RADIX = 1 << BITS, DPT = RADIX / SORT_THREADS, NW = SORT_THREADS / 64;
    if (fw_gate_closed(gate)) return;
    MPF_FW_SETPRIO(prio);
    __shared__ uint32_t running[RADIX];
    __shared__ uint32_t cnt[NW][RADIX];
    __shared__ uint32_t wave_tot[NW];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint32_t dbase[DPT];
    {   // base[d] = exclusive scan of the digit totals; thread t owns the DPT consecutive digits t*DPT ..
        uint32_t v[DPT], sum = 0;
#pragma unroll
        for (int k = 0; k < DPT; ++k) { v[k] = totals[tid * DPT + k]; sum += v[k]; }
        uint32_t inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= (uint32_t)off) inc += o;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        uint32_t base = inc - sum;
        for (uint32_t k = 0; k < wave; ++k) base += wave_tot[k];
#pragma unroll
        for (int k = 0; k < DPT; ++k) { dbase[k] = base; base += v[k]; }
    }
    for (uint32_t tile = blockIdx.x; tile < nb; tile += gridDim.x) {       // a workgroup walks several tiles when the grid is capped
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
            const uint32_t d = tid * DPT + k;
            running[d] = dbase[k] + offsets[(size_t)tile * RADIX + d];
        }
        const uint32_t base = tile * SORT_TILE;
        for (int it = 0; it < SORT_ITEMS; ++it) {
            const uint32_t i = base + it * SORT_THREADS + tid;
            const bool valid = i < N;
            const uint32_t key = valid ? keys_in[i] : 0xFFFFFFFFu;
            const uint32_t val = valid ? vals_in[i] : 0u;
            const uint32_t digit = (key >> shift) & (RADIX - 1);
#pragma unroll
            for (int k = 0; k < RADIX / 64; ++k) cnt[wave][lane + 64 * k] = 0;
            unsigned long long mask = __ballot(valid);
#pragma unroll
            for (int b = 0; b < BITS; ++b) {
                const bool bit = (digit >> b) & 1;
                const unsigned long long bal = __ballot(bit);
                mask &= bit ? bal : ~bal;
            }
            const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
            if (valid && rank == 0) cnt[wave][digit] = __popcll(mask);
            __syncthreads();
            if (valid) {
                uint32_t pos = running[digit] + rank;
                for (uint32_t k = 0; k < wave; ++k) pos += cnt[k][digit];
                keys_out[pos] = key;
                vals_out[pos] = val;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < DPT; ++k) {
                const uint32_t d = tid + k * SORT_THREADS;
                uint32_t add = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) add += cnt[w][d];
                running[d] += add;
            }
            __syncthreads();
        }
    }
}

template <int BITS>
static void radix_pass(const uint32_t *kin, const uint32_t *vin, uint32_t *kout, uint32_t *vout, uint32_t N, int shift, uint32_t nb,
                       uint32_t *hist, uint32_t *totals, hipStream_t st, const FwGate gate)
{
    hipLaunchKernelGGL((k_radix_hist<BITS>), dim3(nb), dim3(SORT_THREADS), 0, st, kin, N, shift, nb, hist, gate);
    hipLaunchKernelGGL(k_radix_colscan, dim3((1u << BITS) / 4u), dim3(256), 0, st, hist, nb, 1u << BITS, totals, g_fw_prio, gate);
    hipLaunchKernelGGL((k_radix_scatter<BITS>), dim3(nb), dim3(SORT_THREADS), 0, st, kin, vin, kout, vout, N, shift, nb, hist, totals, g_fw_prio, gate);
}

#ifdef MPF_WITNESS
// ---- buckets: finish the sort inside each high-digit bucket and resolve it, in ONE kernel ------------------------------------
// After ONE stable pass on the HIGH bits of the target (bucket = 2^LB consecutive targets, about an image row), a bucket's visitors
// are contiguous and in raster order.  One workgroup per bucket then (A) counts them per target, (B) scans the counts, (C) places
// them stably by target - the same ballot ranking as k_radix_scatter, with cursors in LDS and the bucket's slice of the second
// key/value arrays as destination - and (D) resolves every target of the bucket: per slot "z < z of the previous visitor of the
// same target (1000 for the first)", the last such slot per target through an LDS atomicMax, and the segment's last slot writes
// the 5 output bytes (warping.c:13-29).  Targets of the bucket nobody visited are cleared here when the caller wants a cleared
// image.  This replaces the second histogram / scan / scatter and the global mark / write passes (5 launches, 44 us at 640 x 960).
// Pile-ups stay bounded: a bucket of any size is walked in slabs of 256 by its workgroup.
#define FW_BUCKET_CAP 2048        // visitors of a bucket kept in LDS (sorted key, source index, z); larger buckets go through global memory

template <int LB>
__global__ void __launch_bounds__(SORT_THREADS)
k_fw_bucket(const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint32_t *__restrict__ keys_out,
            uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ totals, uint32_t N, const float *__restrict__ z,
            const uint8_t *__restrict__ src, uint8_t *__restrict__ warped, int zero_fill, const int prio = 0,
            const float *__restrict__ src_f = nullptr)
{
    constexpr int NT = 1 << LB, DPT = NT / SORT_THREADS, NW = SORT_THREADS / 64;
    MPF_FW_SETPRIO(prio);
    __shared__ uint32_t hcount[NT];            // visitors per target of the bucket (kept for the clears)
    __shared__ uint32_t cursor[NT];            // running insertion point per target, then: winner slot + 1 per target
    __shared__ uint32_t cnt[NW][NT];
    __shared__ uint32_t red[NW];
    __shared__ uint32_t skey[FW_BUCKET_CAP];   // the bucket's visitors sorted by (target, raster index): target, source index, z
    __shared__ uint32_t sval[FW_BUCKET_CAP];
    __shared__ float sz[FW_BUCKET_CAP];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t nbuckets = (uint32_t)(((uint64_t)N + NT - 1) >> LB);
    for (uint32_t b = blockIdx.x; b < nbuckets; b += gridDim.x) {              // a workgroup walks several buckets when the grid is capped
    // this bucket's slice [start, start + n) of the sorted-by-high-digit arrays: start = sum of the totals of the buckets below
    uint32_t part = 0;
    for (uint32_t k = tid; k < b; k += SORT_THREADS) part += totals[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) red[wave] = part;
#pragma unroll
    for (int k = 0; k < DPT; ++k) hcount[tid + k * SORT_THREADS] = 0;
    __syncthreads();
    uint32_t start = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) start += red[k];
    const uint32_t n = totals[b];
    const bool in_lds = n <= FW_BUCKET_CAP;     // uniform
    const uint32_t *kin = keys_in + start, *vin = vals_in + start;
    uint32_t *ko = keys_out + start, *vo = vals_out + start;
    // (A) visitors per target
    for (uint32_t j = tid; j < n; j += SORT_THREADS) atomicAdd(&hcount[kin[j] & (NT - 1)], 1u);
    __syncthreads();
    // (B) exclusive scan of hcount -> cursor (thread t owns DPT consecutive targets)
    {
        uint32_t v[DPT], sum = 0;
#pragma unroll
        for (int k = 0; k < DPT; ++k) { v[k] = hcount[tid * DPT + k]; sum += v[k]; }
        uint32_t inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(inc, off);
            if (lane >= (uint32_t)off) inc += o;
        }
        __syncthreads();
        if (lane == 63) red[wave] = inc;
        __syncthreads();
        uint32_t base = inc - sum;
        for (uint32_t k = 0; k < wave; ++k) base += red[k];
#pragma unroll
        for (int k = 0; k < DPT; ++k) { cursor[tid * DPT + k] = base; base += v[k]; }
    }
    __syncthreads();
    // (C) stable placement by target, slab by slab (the visitor's z travels with it)
    for (uint32_t j0 = 0; j0 < n; j0 += SORT_THREADS) {
        const uint32_t j = j0 + tid;
        const bool valid = j < n;
        const uint32_t key = valid ? kin[j] : 0xFFFFFFFFu;
        const uint32_t val = valid ? vin[j] : 0u;
        const float zv = (valid && in_lds) ? z[val] : 0.0f;
        const uint32_t digit = key & (NT - 1);
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) cnt[wave][lane + 64 * k] = 0;
        unsigned long long mask = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < LB; ++bit) {
            const bool on = (digit >> bit) & 1;
            const unsigned long long bal = __ballot(on);
            mask &= on ? bal : ~bal;
        }
        const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
        if (valid && rank == 0) cnt[wave][digit] = __popcll(mask);
        __syncthreads();
        if (valid) {
            uint32_t pos = cursor[digit] + rank;
            for (uint32_t k = 0; k < wave; ++k) pos += cnt[k][digit];
            if (in_lds) { skey[pos] = key; sval[pos] = val; sz[pos] = zv; }
            else { ko[pos] = key; vo[pos] = val; }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
            const uint32_t d = tid + k * SORT_THREADS;
            uint32_t add = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) add += cnt[w][d];
            cursor[d] += add;
        }
        __syncthreads();
    }
    // the slice is complete and sorted by (target, raster index); make the global form visible to the whole workgroup
    if (!in_lds) __threadfence_block();
#pragma unroll
    for (int k = 0; k < DPT; ++k) cursor[tid + k * SORT_THREADS] = 0;          // now: winner slot + 1 per target
    __syncthreads();
    auto key_at = [&](uint32_t j) -> uint32_t { return in_lds ? skey[j] : ko[j]; };
    auto val_at = [&](uint32_t j) -> uint32_t { return in_lds ? sval[j] : vo[j]; };
    auto z_at = [&](uint32_t j) -> float { return in_lds ? sz[j] : z[vo[j]]; };
    // (D) resolve
    for (uint32_t j = tid; j < n; j += SORT_THREADS) {
        const uint32_t t = key_at(j);
        const bool has_pred = (j > 0) && (key_at(j - 1) == t);
        const float zprev = has_pred ? z_at(j - 1) : 1000.0f;                   // dlut, warping.c:11, :29
        if (z_at(j) < zprev) atomicMax(&cursor[t & (NT - 1)], j + 1);           // warping.c:19
    }
    __syncthreads();
    for (uint32_t j = tid; j < n; j += SORT_THREADS) {
        const uint32_t t = key_at(j);
        if (j + 1 < n && key_at(j + 1) == t) continue;                          // only the last visitor of a target writes
        const bool has_pred = (j > 0) && (key_at(j - 1) == t);
        const float zprev = has_pred ? z_at(j - 1) : 1000.0f;
        uint8_t *o = warped + (size_t)t * 5;
        const uint32_t wj = cursor[t & (NT - 1)];
        if (wj) {                                                               // no visitor passed the z test: the colour bytes keep
            const uint32_t v = val_at(wj - 1);                                  // what they held (warping.c:19-21)
            if (src_f) {                                                        // the frame given as float [3,h,w] in 0..1: its uint8 BGR form, utils/utils.py:174-177
#pragma unroll
                for (int c = 0; c < 3; ++c) o[c] = mpf_to_u8(src_f[(size_t)(2 - c) * N + v]);
            } else {
                const uint8_t *sp = src + (size_t)v * 3;
                o[0] = sp[0]; o[1] = sp[1]; o[2] = sp[2];
            }
        } else if (zero_fill) {
            o[0] = o[1] = o[2] = 0;
        }
        o[3] = 1;                                                               // warping.c:23
        o[4] = (zprev == 1000.0f) ? 1 : 0;                                      // warping.c:24-27
    }
    if (zero_fill) {                                                            // targets nobody visited (moving_obj.py:123 zero-inits)
#pragma unroll
        for (int k = 0; k < DPT; ++k) {
            const uint32_t tl = tid + k * SORT_THREADS;
            const uint64_t t = ((uint64_t)b << LB) + tl;
            if (hcount[tl] == 0 && t < N) {
                uint8_t *o = warped + (size_t)t * 5;
                o[0] = o[1] = o[2] = o[3] = o[4] = 0;
            }
        }
    }
    __syncthreads();                                                           // the LDS tables are re-used by the next bucket
    }
}

#endif   // MPF_WITNESS

// ---- round 5: gather instead of sort ----------------------------------------------------------------------------------------------
#define FWG_TILE 1024             // sources per tile of pass 1 (one 256-thread workgroup, 4 per thread) = 16 slabs of 64
#define FWG_LB 8                  // a bucket = 256 consecutive targets, resolved by one wave
#define FWG_CAP 512               // visitors a wave keeps in LDS per chunk; a bucket with more is streamed through in chunks

MPF_DEV uint32_t fwg_wave_min(uint32_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = __shfl_xor(v, off); v = o < v ? o : v; }
    return v;
}
MPF_DEV uint32_t fwg_wave_max(uint32_t v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const uint32_t o = __shfl_xor(v, off); v = o > v ? o : v; }
    return v;
}

// Pass 1: the key (target) of every source - computed by the projection of moving_obj.py:29-124 (PROJ) or from the caller's idx / idy -, and the
// [min, max] key of every 64-source slab and every 1024-source tile.  Slabs / tiles beyond N get the empty range [0xFFFFFFFF, 0].
template <bool PROJ>
__global__ void __launch_bounds__(256)
k_fw_keys_ranges(const float *__restrict__ disp, const MpfMoProj m, const float *__restrict__ inst, const MpfMoOut o, const int64_t *__restrict__ idx,
                 const int64_t *__restrict__ idy, int h, int w, uint32_t N, uint32_t ntiles, uint32_t *__restrict__ keys, uint32_t *__restrict__ slab_min,
                 uint32_t *__restrict__ slab_max, uint32_t *__restrict__ tile_min, uint32_t *__restrict__ tile_max, unsigned long long *__restrict__ work = nullptr)
{
    __shared__ uint32_t red[4][3];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint32_t tmin = 0xFFFFFFFFu, tmax = 0u, visits = 0u;     // visits: buckets of 256 targets this wave's four slabs touch = what pass 2 will spend on them
        // the four pixels' inputs first: the outputs are plain (possibly aliasing) pointers, so a load behind a store would wait for it
        float dv[4], iv[4];
        int64_t xv[4], yv[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t n = tile * FWG_TILE + it * 256 + tid;
            const uint32_t nc = n < N ? n : N - 1;
            if constexpr (PROJ) { dv[it] = disp[nc]; iv[it] = inst[nc]; }
            else { xv[it] = idx[nc]; yv[it] = idy[nc]; }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const uint32_t n = tile * FWG_TILE + it * 256 + tid;
            uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
            if (n < N) {
                uint32_t key;
                if constexpr (PROJ) {
                    key = mpf_mo_pixel_v(dv[it], m, iv[it], h, w, (int64_t)n, o);
                } else {
                    int64_t x = xv[it], y = yv[it];
                    x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);                      // the reference does no bounds check (the caller pre-clamps, moving_obj.py:121-122)
                    y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
                    key = (uint32_t)(y * w + x);
                }
                keys[n] = key;
                kmin = kmax = key;
            }
            kmin = fwg_wave_min(kmin);
            kmax = fwg_wave_max(kmax);
            if (lane == 0) {
                const uint32_t slab = tile * 16 + it * 4 + wave;
                slab_min[slab] = kmin;
                slab_max[slab] = kmax;
            }
            tmin = kmin < tmin ? kmin : tmin;
            tmax = kmax > tmax ? kmax : tmax;
            if (kmin <= kmax) visits += (kmax >> FWG_LB) - (kmin >> FWG_LB) + 1u;
        }
        if (lane == 0) { red[wave][0] = tmin; red[wave][1] = tmax; red[wave][2] = visits; }
        __syncthreads();
        if (tid == 0) {
            uint32_t a = red[0][0], b = red[0][1];
#pragma unroll
            for (int k = 1; k < 4; ++k) { a = red[k][0] < a ? red[k][0] : a; b = red[k][1] > b ? red[k][1] : b; }
            tile_min[tile] = a;
            tile_max[tile] = b;
            if (work) atomicAdd(work, (unsigned long long)red[0][2] + red[1][2] + red[2][2] + red[3][2]);
        }
        __syncthreads();
    }
}

// Pass 2: ONE WAVE per bucket of 256 consecutive targets, four such waves per workgroup with nothing in common (private LDS slices; MPF_WAVE_SYNC is a
// compiler fence, not a barrier).  (1) the tiles, then the slabs, whose key range touches the bucket; (2) their keys, 64 at a time in raster order: the matching
// lanes append (source index, target, z) to the chunk; (3) per chunk: count per target, scan, stable placement by target (ballot ranking),
// then per sorted slot "z < z of the previous visitor of the same target" - the previous visitor of a chunk's first slot of a target is the
// carried last z (1000 for a target nobody visited yet: dlut, warping.c:11) -, the last such slot per target (LDS atomicMax) becomes the
// target's winner so far, the last slot of every target updates its carried z and its collision flag (warping.c:24-27);  (4) the 5 output
// bytes of every target of the bucket (warping.c:13-29), and on request the planes H = valid, M = 1 - (collision == valid)
// (moving_obj.py:133-142) the mask kernel would otherwise re-read from the 5-byte records.
// wave-level publication of LDS writes: a wave's LDS operations complete in order, so all that is needed is that the compiler neither reorders nor caches them
#define MPF_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define FWG_WAVES 4               // buckets per workgroup, one per wave: four waves on the four SIMDs of one CU take ONE 256-thread workgroup's slot from a chip-filling
                                  // launch on another stream; as single-wave workgroups each of them could cost such a slot on a different CU

__global__ void __launch_bounds__(64 * FWG_WAVES)
k_fw_gather_resolve(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ slab_min, const uint32_t *__restrict__ slab_max,
                    const uint32_t *__restrict__ tile_min, const uint32_t *__restrict__ tile_max, uint32_t N, uint32_t ntiles, const float *__restrict__ z,
                    const uint8_t *__restrict__ src, const float *__restrict__ src_f, uint8_t *__restrict__ warped, int zero_fill,
                    uint8_t *__restrict__ Hm, uint8_t *__restrict__ Mm, const FwGate gate = FwGate{nullptr, 0, false})
{
    constexpr int NT = 1 << FWG_LB;
    if (fw_gate_closed(gate)) return;                        // scattered targets: the radix path behind this launch owns the call
    __shared__ uint32_t g_src_[FWG_WAVES][FWG_CAP], g_tl_[FWG_WAVES][FWG_CAP];      // gathered, raster order
    __shared__ float g_z_[FWG_WAVES][FWG_CAP];
    __shared__ uint32_t s_src_[FWG_WAVES][FWG_CAP], s_tl_[FWG_WAVES][FWG_CAP];      // sorted by (target, raster index)
    __shared__ float s_z_[FWG_WAVES][FWG_CAP];
    __shared__ uint32_t hc_[FWG_WAVES][NT], cur_[FWG_WAVES][NT], winslot_[FWG_WAVES][NT], winsrc_[FWG_WAVES][NT], state_[FWG_WAVES][NT];
    __shared__ float zlast_[FWG_WAVES][NT];
    __shared__ uint32_t clist_[FWG_WAVES][512];             // candidate tiles of one batch of the tile table (ordered)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t *const g_src = g_src_[wave], *const g_tl = g_tl_[wave], *const s_src = s_src_[wave], *const s_tl = s_tl_[wave];
    float *const g_z = g_z_[wave], *const s_z = s_z_[wave], *const zlast = zlast_[wave];
    uint32_t *const hc = hc_[wave], *const cur = cur_[wave], *const winslot = winslot_[wave], *const winsrc = winsrc_[wave], *const state = state_[wave], *const clist = clist_[wave];
    const unsigned long long lt = (1ull << lane) - 1ull;
    const uint32_t nbuckets = (N + NT - 1) >> FWG_LB;

    // one chunk of `cnt` gathered visitors -> the carried per-target state
    auto process = [&](const uint32_t cnt) {
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) hc[lane + 64 * k] = 0;
        MPF_WAVE_SYNC();
        for (uint32_t j = lane; j < cnt; j += 64) atomicAdd(&hc[g_tl[j]], 1u);
        MPF_WAVE_SYNC();
        {   // exclusive scan: lane l owns targets 4l .. 4l+3
            uint32_t v[NT / 64], sum = 0;
#pragma unroll
            for (int k = 0; k < NT / 64; ++k) { v[k] = hc[lane * (NT / 64) + k]; sum += v[k]; }
            uint32_t inc = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = __shfl_up(inc, off);
                if (lane >= (uint32_t)off) inc += o;
            }
            uint32_t base = inc - sum;
#pragma unroll
            for (int k = 0; k < NT / 64; ++k) { cur[lane * (NT / 64) + k] = base; base += v[k]; }
        }
        MPF_WAVE_SYNC();
        for (uint32_t j0 = 0; j0 < cnt; j0 += 64) {           // stable placement, 64 visitors (in raster order) per round
            const uint32_t j = j0 + lane;
            const bool valid = j < cnt;
            const uint32_t d = valid ? g_tl[j] : 0u;
            unsigned long long mask = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < FWG_LB; ++bit) {
                const bool on = (d >> bit) & 1;
                const unsigned long long bal = __ballot(on);
                mask &= on ? bal : ~bal;
            }
            const uint32_t rank = __popcll(mask & lt);
            if (valid) {
                const uint32_t pos = cur[d] + rank;
                s_src[pos] = g_src[j]; s_tl[pos] = d; s_z[pos] = g_z[j];
            }
            MPF_WAVE_SYNC();
            if (valid && rank == 0) cur[d] += __popcll(mask);
            MPF_WAVE_SYNC();
        }
        for (uint32_t j = lane; j < cnt; j += 64) {           // z test against the previous visitor of the same target
            const uint32_t t = s_tl[j];
            const bool has_pred = (j > 0) && (s_tl[j - 1] == t);
            const float zprev = has_pred ? s_z[j - 1] : zlast[t];
            if (s_z[j] < zprev) atomicMax(&winslot[t], j + 1);                    // warping.c:19
        }
        MPF_WAVE_SYNC();
        for (uint32_t j = lane; j < cnt; j += 64) {           // the last visitor of every target in this chunk carries the state on
            const uint32_t t = s_tl[j];
            if (j + 1 < cnt && s_tl[j + 1] == t) continue;
            const bool has_pred = (j > 0) && (s_tl[j - 1] == t);
            const float zprev = has_pred ? s_z[j - 1] : zlast[t];
            state[t] = 1u | ((zprev == 1000.0f) ? 2u : 0u);                         // visited | collision byte (warping.c:24-27)
            const uint32_t wj = winslot[t];
            if (wj) { winsrc[t] = s_src[wj - 1] + 1u; winslot[t] = 0u; }
            zlast[t] = s_z[j];                                                       // dlut[y,x] = z, unconditionally (warping.c:29)
        }
        MPF_WAVE_SYNC();
    };

    for (uint32_t b = blockIdx.x * FWG_WAVES + wave; b < nbuckets; b += gridDim.x * FWG_WAVES) {
        const uint32_t lo = b << FWG_LB, hi = lo + (NT - 1);
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) {
            const uint32_t t = lane + 64 * k;
            zlast[t] = 1000.0f; winslot[t] = 0u; winsrc[t] = 0u; state[t] = 0u;
        }
        uint32_t cnt = 0;
        MPF_WAVE_SYNC();
        // candidate tiles, 512 table entries per batch (8 independent loads per lane in flight); typically a bucket sees 1-3 candidates
        for (uint32_t t0 = 0; t0 < ntiles; t0 += 512) {
            uint32_t mn[8], mx[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t tile = t0 + 64 * k + lane;
                const bool ok = tile < ntiles;
                mn[k] = ok ? tile_min[tile] : 0xFFFFFFFFu;
                mx[k] = ok ? tile_max[tile] : 0u;
            }
            uint32_t nct = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool hit = mn[k] <= hi && mx[k] >= lo;
                const unsigned long long bm = __ballot(hit);
                if (hit) clist[nct + __popcll(bm & lt)] = t0 + 64 * k + lane;
                nct += __popcll(bm);
            }
            MPF_WAVE_SYNC();
            for (uint32_t c0 = 0; c0 < nct; c0 += 4) {       // 4 candidate tiles = 64 slabs per round
                const uint32_t ci = c0 + (lane >> 4);
                const uint32_t slab = ci < nct ? clist[ci] * 16 + (lane & 15u) : 0u;
                const bool shit = ci < nct && slab_min[slab] <= hi && slab_max[slab] >= lo;
                unsigned long long sm = __ballot(shit);
                while (sm) {                                   // the touching slabs in raster order, EIGHT at a time (their 16 loads in flight together:
                                                               // a bucket of c3's white-noise box sees ~165 of them, each a dependent round trip under load)
                    uint32_t key[8];
                    float zz[8];
                    uint32_t nn[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        uint32_t n = 0xFFFFFFFFu;
                        if (sm) {
                            const int first = __ffsll((long long)sm) - 1;
                            sm &= sm - 1ull;
                            n = __shfl(slab, first) * 64 + lane;
                        }
                        const bool ok = n < N;
                        nn[q] = n;
                        key[q] = ok ? keys[n] : 0xFFFFFFFFu;
                        zz[q] = ok ? z[n] : 0.0f;
                    }
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        if (cnt > FWG_CAP - 256) {             // the next four slabs might not fit: fold the chunk into the carried state
                            MPF_WAVE_SYNC();
                            process(cnt);
                            cnt = 0;
                        }
#pragma unroll
                        for (int q = 4 * half; q < 4 * half + 4; ++q) {
                            const bool match = (key[q] >> FWG_LB) == b && nn[q] < N;
                            const unsigned long long mm = __ballot(match);
                            if (match) {
                                const uint32_t pos = cnt + __popcll(mm & lt);
                                g_src[pos] = nn[q]; g_tl[pos] = key[q] & (NT - 1); g_z[pos] = zz[q];
                            }
                            cnt += __popcll(mm);
                        }
                    }
                }
            }
            MPF_WAVE_SYNC();
        }
        MPF_WAVE_SYNC();
        if (cnt) process(cnt);
        // the 5 bytes of every target of the bucket
#pragma unroll
        for (int k = 0; k < NT / 64; ++k) {
            const uint32_t tl = lane + 64 * k;
            const uint64_t t = (uint64_t)lo + tl;
            if (t >= N) continue;
            const uint32_t st = state[tl];
            uint8_t *o = warped + (size_t)t * 5;
            if (st & 1u) {
                const uint32_t ws = winsrc[tl];
                if (ws) {                                                           // no visitor passed the z test: the colour bytes keep what they held
                    const uint32_t v = ws - 1u;
                    if (src_f) {                                                    // the frame given as float [3,h,w] in 0..1: its uint8 BGR form, utils/utils.py:174-177
#pragma unroll
                        for (int c = 0; c < 3; ++c) o[c] = mpf_to_u8(src_f[(size_t)(2 - c) * N + v]);
                    } else {
                        const uint8_t *sp = src + (size_t)v * 3;
                        o[0] = sp[0]; o[1] = sp[1]; o[2] = sp[2];
                    }
                } else if (zero_fill) {
                    o[0] = o[1] = o[2] = 0;
                }
                o[3] = 1;                                                           // warping.c:23
                o[4] = (st >> 1) & 1u;
            } else if (zero_fill) {                                                 // targets nobody visited (moving_obj.py:123 zero-inits)
                o[0] = o[1] = o[2] = o[3] = o[4] = 0;
            }
            if (Hm) {
                Hm[t] = (uint8_t)(st & 1u);
                Mm[t] = (uint8_t)((st & 1u) && !(st & 2u));                         // 1 - (collision == valid)
            }
        }
        MPF_WAVE_SYNC();
    }
}

// moving_obj.py:143-150 from the planes pass 2 wrote: M' = dilate3x3(M), P = (M' == M), H' = H * P.  Four pixels per thread.
__global__ void __launch_bounds__(256)
k_warp_masks_planes(const uint8_t *__restrict__ Hm, const uint8_t *__restrict__ Mm, int H, int W, uint8_t *__restrict__ Md, uint8_t *__restrict__ P,
                    uint8_t *__restrict__ Hp)
{
    const int W4 = (W + 3) >> 2;
    const int64_t total = (int64_t)H * W4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / W4), x0 = (int)(i - (int64_t)y * W4) * 4;
        uint8_t col[6] = {0, 0, 0, 0, 0, 0};                                        // column maxima over rows y-1..y+1 for x0-1 .. x0+4 (the border never wins the max)
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            const uint8_t *row = Mm + (size_t)yy * W;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int xx = x0 - 1 + k;
                if (xx >= 0 && xx < W) { const uint8_t v = row[xx]; col[k] = v > col[k] ? v : col[k]; }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = x0 + k;
            if (x >= W) break;
            const size_t n = (size_t)y * W + x;
            uint8_t md = col[k] > col[k + 1] ? col[k] : col[k + 1];
            md = col[k + 2] > md ? col[k + 2] : md;
            const uint8_t p = (uint8_t)(md == Mm[n]);
            Md[n] = md; P[n] = p; Hp[n] = (uint8_t)(Hm[n] * p);
        }
    }
}

// ---- resolve (the general path: images above 2^22 pixels) ------------------------------------------------------------------

__global__ void __launch_bounds__(256)
k_fw_mark(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, const float *__restrict__ z, uint32_t N,
          uint32_t *__restrict__ win, const FwGate gate = FwGate{nullptr, 0, true})
{
    if (fw_gate_closed(gate)) return;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const uint32_t t = keys[j];
    const bool has_pred = (j > 0) && (keys[j - 1] == t);
    const float zprev = has_pred ? z[vals[j - 1]] : 1000.0f;          // dlut, warping.c:11, :29
    if (z[vals[j]] < zprev) atomicMax(&win[t], j + 1);                // warping.c:19
}

__global__ void __launch_bounds__(256)
k_fw_write(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals, const float *__restrict__ z,
           const uint8_t *__restrict__ src, uint32_t N, const uint32_t *__restrict__ win, uint8_t *__restrict__ warped,
           const float *__restrict__ src_f = nullptr, const FwGate gate = FwGate{nullptr, 0, true})
{
    if (fw_gate_closed(gate)) return;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const uint32_t t = keys[j];
    if (j + 1 < N && keys[j + 1] == t) return;                        // only the last visitor of a target writes
    const bool has_pred = (j > 0) && (keys[j - 1] == t);
    const float zprev = has_pred ? z[vals[j - 1]] : 1000.0f;
    uint8_t *o = warped + (size_t)t * 5;
    const uint32_t wj = win[t];
    if (wj) {                                                         // no visitor ever passed the z test: colour bytes
        const uint32_t v = vals[wj - 1];                              // keep what they held (warping.c:19-21)
        if (src_f) {
#pragma unroll
            for (int c = 0; c < 3; ++c) o[c] = mpf_to_u8(src_f[(size_t)(2 - c) * N + v]);
        } else {
            const uint8_t *s = src + (size_t)v * 3;
            o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
        }
    }
    o[3] = 1;                                                         // warping.c:23
    o[4] = (zprev == 1000.0f) ? 1 : 0;                                // warping.c:24-27
}

#ifdef MPF_WITNESS
static int g_fw_path = 0;       // mpf_tune("fwarp_path", p): 0 = gather (round 5; images up to 2^24 pixels), 1 = the general multi-pass radix path (what larger
                                // images take), 2 = round 2's one-pass sort + per-bucket workgroups (kept for A/B and as a second witness in the tests).
                                // Process-global and not thread-safe, like every mpf_tune knob: set it before launching work, from one thread.

void mpf_fwarp_set_path(int v) { g_fw_path = v; }
#else
static constexpr int g_fw_path = 0;     // the product build: gather (<= 2^24 pixels) with the general radix path behind it; the other paths are witnesses
#endif
static long long g_fw_gate_thr = -1;   // mpf_tune("fwarp_gate", t): bucket visits above which caller-supplied targets take the radix path (-1 = default, one per source;
                                       // 0 = always radix behind the gate - same results either way: the tests run both sides of it)
void mpf_fwarp_set_gate(int v) { g_fw_gate_thr = v; }
static int g_fw_grid = 0;       // mpf_tune("chain_grid", g): cap the workgroup count of every sort / resolve / mask launch at g (0 = one per tile);
                                // fewer, longer-lived workgroups for runs underneath a chip-filling launch of another stream
void mpf_fwarp_set_grid(int v) { g_fw_grid = v < 0 ? 0 : v; }
static inline uint32_t fw_cap(uint32_t n) { return (g_fw_grid > 0 && n > (uint32_t)g_fw_grid) ? (uint32_t)g_fw_grid : n; }
void mpf_fwarp_set_prio(int v) { g_fw_prio = v < 0 ? 0 : (v > 3 ? 3 : v); }

static inline uint32_t fw_blocks(int64_t N) { return (uint32_t)((N + SORT_TILE - 1) / SORT_TILE); }

extern "C" size_t mpf_forward_warp_workspace(int h, int w)
{
    const int64_t N = (int64_t)h * w;
    if (N <= 0) return 0;
    const size_t a = ((size_t)N * 4 + 255) & ~(size_t)255;
    const size_t hs = (((size_t)RADIX_MAX * fw_blocks(N)) * 4 + 255) & ~(size_t)255;
    return 5 * a + hs + 4 * RADIX_MAX + 256;    // keysA, keysB, valsA, valsB, win, hist, digit totals, the gather / radix gate counter
}

// proj != nullptr (mpf_moving_object_chain): the targets are not given but computed - d_idx / d_idy / d_z are the projection's own outputs
struct FwProj { const float *disp; MpfMoProj m; const float *inst; MpfMoOut out; const float *src_f; };

static int fw_run(const uint8_t *d_src, const int64_t *d_idx, const int64_t *d_idy, const float *d_z, uint8_t *d_warped, int h,
                  int w, void *d_workspace, size_t workspace_bytes, void *stream, bool zero_fill, const FwProj *proj = nullptr,
                  uint8_t *planes_H = nullptr, uint8_t *planes_M = nullptr, bool *planes_written = nullptr)
{
    if (planes_written) *planes_written = false;
    const float *src_f = proj ? proj->src_f : nullptr;
    MPF_REQUIRE((d_src || src_f) && d_idx && d_idy && d_z && d_warped && d_workspace && h >= 1 && w >= 1, "mpf_forward_warp: bad argument");
    const int64_t N64 = (int64_t)h * w;
    MPF_REQUIRE(N64 < ((int64_t)1 << 31), "mpf_forward_warp: image too large");
    MPF_REQUIRE(workspace_bytes >= mpf_forward_warp_workspace(h, w), "mpf_forward_warp: workspace too small");
    MPF_REQUIRE((((uintptr_t)d_workspace) & 255) == 0, "mpf_forward_warp: workspace must be 256-byte aligned");
    const uint32_t N = (uint32_t)N64;
    hipStream_t st = (hipStream_t)stream;
    const size_t a = ((size_t)N * 4 + 255) & ~(size_t)255;
    uint8_t *ws = (uint8_t *)d_workspace;
    uint32_t *keys[2] = { (uint32_t *)ws, (uint32_t *)(ws + a) };
    uint32_t *vals[2] = { (uint32_t *)(ws + 2 * a), (uint32_t *)(ws + 3 * a) };
    uint32_t *win = (uint32_t *)(ws + 4 * a);
    uint32_t *hist = (uint32_t *)(ws + 5 * a);
    const uint32_t nb = fw_blocks(N);
    uint32_t *totals = hist + (((size_t)RADIX_MAX * nb + 63) & ~(size_t)63);
    unsigned long long *work = (unsigned long long *)(totals + RADIX_MAX);
    const uint32_t g256 = (N + 255) / 256;

    int bits = 0;
    while (bits < 32 && ((uint64_t)1 << bits) < (uint64_t)N) ++bits;
    FwGate gate = FwGate{nullptr, 0, true};                   // set by the gather branch for caller-supplied targets: the radix launches below then run conditionally
    if (bits <= 24 && g_fw_path == 0) {
        // round 5: gather instead of sort - keys + slab / tile ranges, then one wave per bucket of 256 targets (k_fw_gather_resolve)
        const uint32_t ntiles = (N + FWG_TILE - 1) / FWG_TILE, nslabs = ntiles * 16, nbuckets = (N + (1u << FWG_LB) - 1) >> FWG_LB;
        uint32_t *slab_min = keys[1], *slab_max = keys[1] + nslabs, *tile_min = slab_max + nslabs, *tile_max = tile_min + ntiles;   // 34 N / 1024 words of the second key array
        static const MpfMoProj no_proj = {};
        if (proj) {
            // the chain: targets are the projection of raster-ordered sources under one rigid motion - a slab's key range is bounded by the parallax, not by the caller
            hipLaunchKernelGGL((k_fw_keys_ranges<true>), dim3(fw_cap(ntiles)), dim3(256), 0, st, proj->disp, proj->m, proj->inst, proj->out, d_idx, d_idy, h, w, N,
                               ntiles, keys[0], slab_min, slab_max, tile_min, tile_max, (unsigned long long *)nullptr);
            hipLaunchKernelGGL(k_fw_gather_resolve, dim3(fw_cap((nbuckets + FWG_WAVES - 1) / FWG_WAVES)), dim3(64 * FWG_WAVES), 0, st, keys[0], slab_min, slab_max, tile_min, tile_max, N, ntiles, d_z, d_src, src_f,
                               d_warped, zero_fill ? 1 : 0, planes_H, planes_M, FwGate{nullptr, 0, false});
            if (planes_written) *planes_written = planes_H != nullptr;
            return mpf_launch_status("forward_warp kernels");
        }
        // caller-supplied targets (mpf_forward_warp / forward_warping): nothing bounds how far the targets of 64 consecutive sources are spread, and the gather
        // costs one slab read per (slab, bucket its range touches) - O(N^2 / 256) for scattered targets, where the radix path stays O(N) per pass.  Pass 1 adds
        // up those visits; beyond `thr` (one visit per source on average: <= 64 N key reads) the gather kernel returns at once and the radix launches behind it,
        // which otherwise return at once, do the work.  Decided on the device: no host round trip, the call stays asynchronous.
        MPF_HIP(hipMemsetAsync(work, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL((k_fw_keys_ranges<false>), dim3(fw_cap(ntiles)), dim3(256), 0, st, (const float *)nullptr, no_proj, (const float *)nullptr, MpfMoOut{}, d_idx,
                           d_idy, h, w, N, ntiles, keys[0], slab_min, slab_max, tile_min, tile_max, work);
        const unsigned long long thr = g_fw_gate_thr >= 0 ? (unsigned long long)g_fw_gate_thr : (unsigned long long)nslabs * 64ull;
        hipLaunchKernelGGL(k_fw_gather_resolve, dim3(fw_cap((nbuckets + FWG_WAVES - 1) / FWG_WAVES)), dim3(64 * FWG_WAVES), 0, st, keys[0], slab_min, slab_max, tile_min, tile_max, N, ntiles, d_z, d_src, src_f,
                           d_warped, zero_fill ? 1 : 0, planes_H, planes_M, FwGate{work, thr, false});
        gate = FwGate{work, thr, true};
    }
#ifdef MPF_WITNESS
    if (!gate.work && bits <= 2 * RADIX_BITS_MAX && g_fw_path == 2) {
        // the fast path: one stable pass on the high bits, then one workgroup per bucket of 2^lb targets sorts and resolves it
        const int lb = bits <= 16 ? 8 : (bits <= 18 ? 9 : (bits <= 20 ? 10 : 11));
        const int hb = bits > lb ? bits - lb : 1;                        // 1 .. 11 high bits (hb < 8: the pass still uses 8-bit digits)
        const int pb = hb < 8 ? 8 : hb;
        const uint32_t nbuckets = (uint32_t)(((uint64_t)N + ((uint64_t)1 << lb) - 1) >> lb);
#define MPF_FW_PASS1(PBv)                                                                                                          \
        if (proj) hipLaunchKernelGGL((k_mo_project_keys_hist<PBv>), dim3(fw_cap(nb)), dim3(SORT_THREADS), 0, st, proj->disp, proj->m, proj->inst, h, w, proj->out, keys[0], vals[0], N, lb, hist, g_fw_prio); \
        else hipLaunchKernelGGL((k_fw_keys_hist<PBv>), dim3(fw_cap(nb)), dim3(SORT_THREADS), 0, st, d_idx, d_idy, h, w, keys[0], vals[0], N, lb, hist);      \
        hipLaunchKernelGGL(k_radix_colscan, dim3(fw_cap((1u << PBv) / 4u)), dim3(256), 0, st, hist, nb, 1u << PBv, totals, g_fw_prio);                       \
        hipLaunchKernelGGL((k_radix_scatter<PBv>), dim3(fw_cap(nb)), dim3(SORT_THREADS), 0, st, keys[0], vals[0], keys[1], vals[1], N, lb, nb, hist, totals, g_fw_prio)
        switch (pb) {
        case 8: MPF_FW_PASS1(8); break;
        case 9: MPF_FW_PASS1(9); break;
        case 10: MPF_FW_PASS1(10); break;
        default: MPF_FW_PASS1(11); break;
        }
#undef MPF_FW_PASS1
#define MPF_FW_BUCKET(LBv) hipLaunchKernelGGL((k_fw_bucket<LBv>), dim3(fw_cap(nbuckets)), dim3(SORT_THREADS), 0, st, keys[1], vals[1], keys[0], vals[0], totals, \
                                              N, d_z, d_src, d_warped, zero_fill ? 1 : 0, g_fw_prio, src_f)
        switch (lb) {
        case 8: MPF_FW_BUCKET(8); break;
        case 9: MPF_FW_BUCKET(9); break;
        case 10: MPF_FW_BUCKET(10); break;
        default: MPF_FW_BUCKET(11); break;
        }
#undef MPF_FW_BUCKET
        return mpf_launch_status("forward_warp kernels");
    }
#endif
    if (proj) hipLaunchKernelGGL(k_moving_object_project, dim3(g256), dim3(256), 0, st, proj->disp, proj->m, proj->inst, h, w, proj->out);
    hipLaunchKernelGGL(k_fw_keys, dim3(g256), dim3(256), 0, st, d_idx, d_idy, h, w, keys[0], vals[0], win, zero_fill ? d_warped : (uint8_t *)nullptr, gate);
    // the fewest passes of 8..11-bit digits that cover the key: 640 x 960 (20 bits) -> 2 x 10, 1024 x 1536 (21 bits) -> 2 x 11
    if (bits < 1) bits = 1;
    const int passes = (bits + RADIX_BITS_MAX - 1) / RADIX_BITS_MAX;
    int dbits = (bits + passes - 1) / passes;
    if (dbits < 8) dbits = 8;
    int cur = 0;
    for (int p = 0; p < passes; ++p) {
        const int shift = p * dbits;
        switch (dbits) {
        case 8: radix_pass<8>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], N, shift, nb, hist, totals, st, gate); break;
        case 9: radix_pass<9>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], N, shift, nb, hist, totals, st, gate); break;
        case 10: radix_pass<10>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], N, shift, nb, hist, totals, st, gate); break;
        default: radix_pass<11>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], N, shift, nb, hist, totals, st, gate); break;
        }
        cur ^= 1;
    }
    hipLaunchKernelGGL(k_fw_mark, dim3(g256), dim3(256), 0, st, keys[cur], vals[cur], d_z, N, win, gate);
    hipLaunchKernelGGL(k_fw_write, dim3(g256), dim3(256), 0, st, keys[cur], vals[cur], d_z, d_src, N, win, d_warped, src_f, gate);
    return mpf_launch_status("forward_warp kernels");
}

extern "C" int mpf_forward_warp(const uint8_t *d_src, const int64_t *d_idx, const int64_t *d_idy, const float *d_z,
                                uint8_t *d_warped, int h, int w, void *d_workspace, size_t workspace_bytes, void *stream)
{
    return fw_run(d_src, d_idx, d_idy, d_z, d_warped, h, w, d_workspace, workspace_bytes, stream, true);
}

// ---- moving_obj.py:133-150 --------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
k_warp_masks(const uint8_t *__restrict__ warped, int H, int W, uint8_t *__restrict__ Hm, uint8_t *__restrict__ M,
             uint8_t *__restrict__ Md, uint8_t *__restrict__ P, uint8_t *__restrict__ Hp, const int prio = 0)
{
    MPF_FW_SETPRIO(prio);
    const int64_t N = (int64_t)H * W;
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(n % W), y = (int)(n / W);
    const uint8_t hv = warped[n * 5 + 3];
    const uint8_t m = (uint8_t)(1 - (warped[n * 5 + 4] == hv));       // M = 1 - (collision == valid)
    uint8_t md = 0;                                                    // cv2.dilate 3x3, border never wins the max
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                const int64_t k = (int64_t)yy * W + xx;
                const uint8_t mk = (uint8_t)(1 - (warped[k * 5 + 4] == warped[k * 5 + 3]));
                md = mk > md ? mk : md;
            }
        }
    Hm[n] = hv; M[n] = m; Md[n] = md;
    const uint8_t p = (uint8_t)(md == m);
    P[n] = p;
    Hp[n] = (uint8_t)(hv * p);
    }
}

extern "C" int mpf_warp_masks(const uint8_t *d_warped, int H, int W, uint8_t *d_Hm, uint8_t *d_M, uint8_t *d_Md, uint8_t *d_P,
                              uint8_t *d_Hp, void *stream)
{
    MPF_REQUIRE(d_warped && d_Hm && d_M && d_Md && d_P && d_Hp && H >= 1 && W >= 1, "mpf_warp_masks: bad argument");
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_warp_masks, dim3(fw_cap((unsigned)((N + 255) / 256))), dim3(256), 0, (hipStream_t)stream, d_warped, H, W, d_Hm,
                       d_M, d_Md, d_P, d_Hp, g_fw_prio);
    return mpf_launch_status("k_warp_masks");
}

// ---- moving_obj.py:29-150 as ONE call: projection (fused into the first sort pass), forward splat, masks ------------------------

extern "C" int mpf_moving_object_chain(const float *d_disp, const float *h_inv_k9, const float *h_P_static12, const float *h_P_obj12,
                                       const float *d_inst, const uint8_t *d_src_u8, const float *d_src_f32_3HW, int H, int W,
                                       const MpfMovingObjectOut *out, void *d_workspace, size_t workspace_bytes, void *stream)
{
    MPF_REQUIRE(d_disp && h_inv_k9 && h_P_static12 && h_P_obj12 && d_inst && out && H >= 1 && W >= 1, "mpf_moving_object_chain: bad argument");
    MPF_REQUIRE((d_src_u8 != nullptr) != (d_src_f32_3HW != nullptr), "mpf_moving_object_chain: give the source frame as uint8 [H,W,3] OR as float [3,H,W], not both");
    MPF_REQUIRE(out->d_p1 && out->d_z1 && out->d_safe_x && out->d_safe_y && out->d_flow01 && out->d_warped, "mpf_moving_object_chain: null output");
    const bool masks = out->d_Hm || out->d_M || out->d_Md || out->d_P || out->d_Hp;
    MPF_REQUIRE(!masks || (out->d_Hm && out->d_M && out->d_Md && out->d_P && out->d_Hp), "mpf_moving_object_chain: the five masks go together");
    FwProj pr;
    pr.disp = d_disp; pr.inst = d_inst;
    memcpy(pr.m.ik, h_inv_k9, sizeof(pr.m.ik));
    memcpy(pr.m.Ps, h_P_static12, sizeof(pr.m.Ps));
    memcpy(pr.m.Po, h_P_obj12, sizeof(pr.m.Po));
    pr.out = MpfMoOut{ out->d_p1, out->d_z1, out->d_safe_x, out->d_safe_y, out->d_flow01 };
    pr.src_f = d_src_f32_3HW;
    bool planes = false;
    const int rc = fw_run(d_src_u8, out->d_safe_x, out->d_safe_y, out->d_z1, out->d_warped, H, W, d_workspace, workspace_bytes, stream, true, &pr,
                          masks ? out->d_Hm : nullptr, masks ? out->d_M : nullptr, &planes);
    if (rc || !masks) return rc;
    if (!planes) return mpf_warp_masks(out->d_warped, H, W, out->d_Hm, out->d_M, out->d_Md, out->d_P, out->d_Hp, stream);
    const int64_t quads = (int64_t)H * ((W + 3) / 4);
    hipLaunchKernelGGL(k_warp_masks_planes, dim3(fw_cap((unsigned)((quads + 255) / 256))), dim3(256), 0, (hipStream_t)stream, out->d_Hm, out->d_M, H, W, out->d_Md, out->d_P,
                       out->d_Hp);
    return mpf_launch_status("k_warp_masks_planes");
}

// ---- the reference's FFI symbol (host pointers) -----------------------------------------------------------------

extern "C" int mpf_forward_warping_host(const void *src, const void *idx, const void *idy, const void *z, void *warped, int h, int w)
{
    MPF_REQUIRE(src && idx && idy && z && warped && h >= 1 && w >= 1, "forward_warping: bad argument");
    const size_t N = (size_t)h * w;
    const size_t wsb = mpf_forward_warp_workspace(h, w);
    uint8_t *d = nullptr;
    // one allocation: src | idx | idy | z | warped | workspace, each 256-byte aligned
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_src = 0, o_idx = o_src + al(N * 3), o_idy = o_idx + al(N * 8), o_z = o_idy + al(N * 8), o_w = o_z + al(N * 4),
                 o_ws = o_w + al(N * 5), total = o_ws + wsb;
    MPF_HIP(hipMalloc((void **)&d, total));
    int rc = 0;
    hipError_t e;
    do {
        if ((e = hipMemcpy(d + o_src, src, N * 3, hipMemcpyHostToDevice)) != hipSuccess) break;
        if ((e = hipMemcpy(d + o_idx, idx, N * 8, hipMemcpyHostToDevice)) != hipSuccess) break;
        if ((e = hipMemcpy(d + o_idy, idy, N * 8, hipMemcpyHostToDevice)) != hipSuccess) break;
        if ((e = hipMemcpy(d + o_z, z, N * 4, hipMemcpyHostToDevice)) != hipSuccess) break;
        // warping.c only touches the bytes of visited targets; everything else keeps the caller's contents
        if ((e = hipMemcpy(d + o_w, warped, N * 5, hipMemcpyHostToDevice)) != hipSuccess) break;
        rc = fw_run(d + o_src, (const int64_t *)(d + o_idx), (const int64_t *)(d + o_idy), (const float *)(d + o_z), d + o_w, h, w,
                    d + o_ws, wsb, nullptr, false);
        if (rc) break;
        if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) break;
        e = hipMemcpy(warped, d + o_w, N * 5, hipMemcpyDeviceToHost);
    } while (0);
    (void)hipFree(d);
    if (rc) return rc;
    if (e != hipSuccess) {
        mpf_set_error("forward_warping: %s", hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

extern "C" void forward_warping(const void *src, const void *idx, const void *idy, const void *z, void *warped, int h, int w)
{
    if (mpf_forward_warping_host(src, idx, idy, z, warped, h, w) != 0)
        fprintf(stderr, "libmpiflow_hip forward_warping: %s\n", mpf_last_error());
}
