// mpf_frames.hip - the output side of a pair: hole fill without host round trips and the
// PNG scanline filter, so that what leaves the GPU is ready for a GIL-free deflate on a writer thread.
//
// Hole filling (row A13): the reference calls cv2.inpaint(frame_mix, fill_mask, 3, INPAINT_NS), utils/utils.py:284-286 -
// third-party OpenCV arithmetic, parity unpinned.  What is here is NOT OpenCV's Navier-Stokes inpainting: it is a
// deterministic onion-peel fill, used when OpenCV is not installed.  Its inputs (frame_mix, fill_mask) are pinned exactly;
// its output is documented as a deviation in DESIGN.md.
#include "mpf_common.h"

namespace {

constexpr int OPEN = 0x7FFFFFFF;            // layer of a hole pixel nobody has reached yet
constexpr int QUEUED = 0x7FFFFFFE;          // ... that sits in the queue of the next pass

// layer[p] = 0 for known pixels, OPEN for holes; hole pixels that touch a known pixel form the queue of pass 1
__global__ __launch_bounds__(256) void k_fill_init(const uint8_t *__restrict__ hole, int H, int W, int *__restrict__ layer,
                                                  int *__restrict__ queue, int *__restrict__ count)
{
    const int64_t N = (int64_t)H * W, n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool front = false;
    if (n < N) {
        const bool h = hole[n] != 0;
        if (h) {
            const int y = (int)(n / W), x = (int)(n - (int64_t)y * W);
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if ((dx || dy) && yy >= 0 && yy < H && xx >= 0 && xx < W && !hole[(int64_t)yy * W + xx]) front = true;
                }
        }
        layer[n] = h ? (front ? QUEUED : OPEN) : 0;
    }
    const unsigned long long b = __ballot(front);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && b) base = atomicAdd(count, __popcll(b));
    base = __shfl(base, 0);
    if (front) queue[base + __popcll(b & ((1ull << lane) - 1ull))] = (int)n;
}

// Onion peel in ONE workgroup, frontier by frontier: pass k fills the queued pixels with the rounded mean of their
// neighbours of layer < k (the state after pass k-1: exactly what k successive full-image passes would compute), stamps
// them with layer k, and queues their still-open neighbours for pass k+1.  Work is proportional to the number of hole
// pixels, the passes are separated by workgroup barriers, nothing is reported to the host: the fill is one launch.
__global__ __launch_bounds__(1024) void k_fill_peel(uint8_t *img, uint8_t *hole, int H, int W, int *layer, int *queue_a, int *queue_b,
                                                    const int *count)
{
    __shared__ int n_next;
    volatile uint8_t *vimg = img;
    volatile int *vlayer = layer;
    int nq = *count;
    int *cur = queue_a, *nxt = queue_b;
    for (int pass = 1; nq > 0; ++pass) {
        if (threadIdx.x == 0) n_next = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < nq; i += blockDim.x) {
            const int p = cur[i];
            const int y = p / W, x = p - y * W;
            unsigned s0 = 0, s1 = 0, s2 = 0, cnt = 0;
            int open_nb[8], n_open = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if ((dx || dy) && yy >= 0 && yy < H && xx >= 0 && xx < W) {
                        const int k = yy * W + xx, lk = vlayer[k];
                        if (lk < pass) { s0 += vimg[3 * k]; s1 += vimg[3 * k + 1]; s2 += vimg[3 * k + 2]; ++cnt; }
                        else if (lk == OPEN) open_nb[n_open++] = k;
                    }
                }
            // cnt >= 1: p was queued by a neighbour filled in the previous pass (or touches a known pixel)
            vimg[3 * p] = (uint8_t)((s0 + cnt / 2) / cnt);
            vimg[3 * p + 1] = (uint8_t)((s1 + cnt / 2) / cnt);
            vimg[3 * p + 2] = (uint8_t)((s2 + cnt / 2) / cnt);
            vlayer[p] = pass;
            hole[p] = 0;
            for (int j = 0; j < n_open; ++j)
                if (atomicCAS(&layer[open_nb[j]], OPEN, QUEUED) == OPEN) nxt[atomicAdd(&n_next, 1)] = open_nb[j];
        }
        __threadfence();
        __syncthreads();
        nq = n_next;
        __syncthreads();
        int *t = cur; cur = nxt; nxt = t;
    }
}

// PNG scanlines with filter type 2 ("Up"): out[y] = (2, row[y] - row[y-1] mod 256), row[-1] = 0; BGR -> RGB on the way.
__global__ __launch_bounds__(256) void k_png_filter_up(const uint8_t *__restrict__ bgr, int H, int W, uint8_t *__restrict__ out)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const size_t pitch = 3 * (size_t)W + 1;
    const uint8_t *cur = bgr + ((size_t)y * W + x) * 3;
    uint8_t *o = out + y * pitch + 1 + 3 * (size_t)x;
    uint8_t p0 = 0, p1 = 0, p2 = 0;
    if (y > 0) {
        const uint8_t *up = cur - 3 * (size_t)W;
        p0 = up[0]; p1 = up[1]; p2 = up[2];
    }
    o[0] = (uint8_t)(cur[2] - p2);
    o[1] = (uint8_t)(cur[1] - p1);
    o[2] = (uint8_t)(cur[0] - p0);
    if (x == 0) out[y * pitch] = 2;
}

// per-pair statistics, deterministic: MPF_PAIR_STATS_SLICES workgroups reduce fixed contiguous slices (fixed per-thread
// strides, fp64 partial sums, LDS tree) into one row each: {sum |flow|, hole pixels, max |flow|, max(-flow component)}
__global__ __launch_bounds__(256) void k_pair_stats(const float *__restrict__ flow, const uint8_t *__restrict__ fill, int64_t N, double *__restrict__ out)
{
    __shared__ double red[4][256];
    const int64_t per = (N + gridDim.x - 1) / gridDim.x, n0 = (int64_t)blockIdx.x * per, n1 = n0 + per < N ? n0 + per : N;
    double s_mag = 0.0, s_hole = 0.0;
    float m_mag = -INFINITY, m_neg = -INFINITY;
    for (int64_t n = n0 + threadIdx.x; n < n1; n += blockDim.x) {
        const float2 f = reinterpret_cast<const float2 *>(flow)[n];
        const float mag = sqrtf(fmaf(f.y, f.y, f.x * f.x));
        s_mag += (double)mag;
        s_hole += fill[n] ? 1.0 : 0.0;
        m_mag = fmaxf(m_mag, mag);
        m_neg = fmaxf(m_neg, fmaxf(-f.x, -f.y));
    }
    red[0][threadIdx.x] = s_mag; red[1][threadIdx.x] = s_hole; red[2][threadIdx.x] = (double)m_mag; red[3][threadIdx.x] = (double)m_neg;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
            red[2][threadIdx.x] = fmax(red[2][threadIdx.x], red[2][threadIdx.x + w]);
            red[3][threadIdx.x] = fmax(red[3][threadIdx.x], red[3][threadIdx.x + w]);
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

}  // namespace

extern "C" int mpf_pair_stats(const float *d_flow_mix, const uint8_t *d_fill_mask, int H, int W, double *d_out, void *stream)
{
    MPF_REQUIRE(d_flow_mix && d_fill_mask && d_out && H >= 1 && W >= 1, "mpf_pair_stats: bad argument");
    MPF_REQUIRE((((uintptr_t)d_flow_mix) & 7) == 0, "mpf_pair_stats: flow must be 8-byte aligned");
    hipLaunchKernelGGL(k_pair_stats, dim3(MPF_PAIR_STATS_SLICES), dim3(256), 0, (hipStream_t)stream, d_flow_mix, d_fill_mask, (int64_t)H * W, d_out);
    return mpf_launch_status("k_pair_stats");
}

// ---- streaming probe: what HBM delivers to the plainest possible kernels on this box (bench.py reports it beside the 8 TB/s peak) ----
typedef float mpf_f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_stream_copy(const mpf_f4 *__restrict__ src, mpf_f4 *__restrict__ dst, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(&src[i]), &dst[i]);
}
__global__ void __launch_bounds__(256) k_stream_read(const mpf_f4 *__restrict__ src, float *__restrict__ sink, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const mpf_f4 v = __builtin_nontemporal_load(&src[i]);
        acc += (v.x + v.y) + (v.z + v.w);
    }
    if (acc == 12345.678f) *sink = acc;                      // keeps the loads alive; never true for the probe's data
}

extern "C" int mpf_stream_probe(const void *d_src, void *d_dst, size_t bytes, int mode, void *stream)
{
    MPF_REQUIRE(d_src && d_dst && bytes >= 16 && (bytes & 15) == 0 && (mode == 0 || mode == 1), "mpf_stream_probe: bad argument");
    MPF_REQUIRE((((uintptr_t)d_src | (uintptr_t)d_dst) & 15) == 0, "mpf_stream_probe: buffers must be 16-byte aligned");
    const size_t n4 = bytes / 16;
    const unsigned blocks = (unsigned)((n4 + 255) / 256 < 256u * 32u ? (n4 + 255) / 256 : 256u * 32u);   // 32 workgroups per CU, grid-stride
    if (mode == 0) hipLaunchKernelGGL(k_stream_read, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const mpf_f4 *)d_src, (float *)d_dst, n4);
    else hipLaunchKernelGGL(k_stream_copy, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const mpf_f4 *)d_src, (mpf_f4 *)d_dst, n4);
    return mpf_launch_status("k_stream_probe");
}

extern "C" size_t mpf_fill_holes_workspace(int H, int W)
{
    const size_t N = (size_t)(H > 0 ? H : 0) * (size_t)(W > 0 ? W : 0);
    return 3 * N * sizeof(int) + 256;               // layer map, two frontier queues, counter
}

extern "C" int mpf_fill_holes(uint8_t *d_img, uint8_t *d_hole, int H, int W, void *d_workspace, size_t workspace_bytes, void *stream)
{
    MPF_REQUIRE(d_img && d_hole && d_workspace && H >= 1 && W >= 1, "mpf_fill_holes: bad argument");
    MPF_REQUIRE((size_t)H * W < 0x7FFFFFFFull / 3, "mpf_fill_holes: image too large");
    MPF_REQUIRE(workspace_bytes >= mpf_fill_holes_workspace(H, W), "mpf_fill_holes: workspace too small (%zu < %zu)", workspace_bytes,
                mpf_fill_holes_workspace(H, W));
    const int64_t N = (int64_t)H * W;
    hipStream_t st = (hipStream_t)stream;
    int *layer = (int *)d_workspace, *qa = layer + N, *qb = qa + N, *count = qb + N;
    MPF_HIP(hipMemsetAsync(count, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_fill_init, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, d_hole, H, W, layer, qa, count);
    hipLaunchKernelGGL(k_fill_peel, dim3(1), dim3(1024), 0, st, d_img, d_hole, H, W, layer, qa, qb, count);
    return mpf_launch_status("k_fill_peel");
}

// Depth-ordered layer pick of the reference's older per-image module ("utils/utils copy.py":283-303): where BOTH
// rendered layers cover a target pixel (np.logical_and(mask, mask_dync): both masks NON-ZERO, not thresholded) and the object layer lies
// behind the background layer (depth > depth_dync), the merged frame takes the dynamic layer's pixel instead.  Everything else is
// Stage D's frame_mix (utils/utils.py:237-276; same quantisation, whitening and select as k_merge).
namespace {
__device__ __forceinline__ uint8_t frames_to_u8(float v)            // np.clip(np.round(x*255), 0, 255).astype(uint8), as mpf_math.h:mpf_to_u8
{
    return (uint8_t)fminf(fmaxf(rintf(v * 255.0f), 0.0f), 255.0f);
}

__global__ void __launch_bounds__(256)
k_merge_depth_ordered(const float *__restrict__ frame, const float *__restrict__ frame_dyn, const float *__restrict__ mask,
                      const float *__restrict__ mask_dyn, const float *__restrict__ depth, const float *__restrict__ depth_dyn, float th,
                      int64_t N, uint8_t *__restrict__ frame_mix_depth, uint8_t *__restrict__ depth_mask)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float m = mask[n], md = mask_dyn[n], z = depth[n], zd = depth_dyn[n];
    float fr[3], fd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        fr[c] = frame[c * N + n];
        fd[c] = frame_dyn[c * N + n];
    }
    const bool sel = m >= th;                                           // "utils copy.py":281-282
    const bool pick = (z > zd) && (m != 0.0f) && (md != 0.0f);          // :295, :301 (NaN masks are truthy, NaN depths compare false, as numpy)
#pragma unroll
    for (int c = 0; c < 3; ++c) {                                       // BGR
        const uint8_t a = (m < th) ? (uint8_t)255 : frames_to_u8(fr[2 - c]);       // :278
        const uint8_t b = (md < th) ? (uint8_t)255 : frames_to_u8(fd[2 - c]);      // :279
        frame_mix_depth[3 * n + c] = pick ? b : (sel ? a : b);          // :302-303
    }
    if (depth_mask) depth_mask[n] = pick ? 1 : 0;
}
}  // namespace

extern "C" int mpf_merge_depth_ordered(const float *d_frame, const float *d_frame_dyn, const float *d_mask, const float *d_mask_dyn,
                                       const float *d_depth, const float *d_depth_dyn, float thresh, int H, int W,
                                       uint8_t *d_frame_mix_depth, uint8_t *d_depth_mask, void *stream)
{
    MPF_REQUIRE(d_frame && d_frame_dyn && d_mask && d_mask_dyn && d_depth && d_depth_dyn && d_frame_mix_depth && H >= 1 && W >= 1,
                "mpf_merge_depth_ordered: bad argument");
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_merge_depth_ordered, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_frame, d_frame_dyn,
                       d_mask, d_mask_dyn, d_depth, d_depth_dyn, thresh, N, d_frame_mix_depth, d_depth_mask);
    return mpf_launch_status("k_merge_depth_ordered");
}

extern "C" int mpf_png_filter_up(const uint8_t *d_bgr, int H, int W, uint8_t *d_scanlines, void *stream)
{
    MPF_REQUIRE(d_bgr && d_scanlines && H >= 1 && W >= 1 && H <= 65535, "mpf_png_filter_up: bad argument");
    hipLaunchKernelGGL(k_png_filter_up, dim3((W + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, d_bgr, H, W, d_scanlines);
    return mpf_launch_status("k_png_filter_up");
}

// ---------------------------------------------------------------------------------------------------------------
// Input stage: uploaded uint8 buffers -> the float tensors the path consumes, resized.
//   image  : transforms.ToTensor()(PIL RGB) = u8 / 255 in fp32 (utils/utils.py:35-39), then
//            F.interpolate(size=(H,W), mode='bilinear', align_corners=True) (gen_3dphoto_dynamic_v2.py:86-87)
//   disp   : cv2.imread(path, 0) / 255 in float64, cast to fp32 (utils/utils.py:42-52), same resize (:88-89)
//   mask   : (ids == obj_index) as float (:101-103), same resize (:104-105)
// The resize reproduces ATen's CPU kernels bit for bit (established against torch 2.10 here, tests/golden/input_stage.npz):
//   scale = float(in-1)/(out-1); src = scale*dst; i0 = min(int(src), in-1); l1 = clamp(src - i0, 0, 1); l0 = 1 - l1;
//   i1 = i0 + (i0 < in-1)  (in == out: i0 = i1 = dst, l0 = 1, l1 = 0), then
//   out_H + out_W > 128 ("generic" kernel):  t_y = fma(v[y][x0], lx0, v[y][x1]*lx1);  out = fma(t_0, ly0, t_1*ly1)
//   out_H + out_W <= 128 (channels-last kernel): w_ab = ly_a*lx_b;  out = fma(v11,w11, fma(v10,w10, fma(v00,w00, v01*w01)))
// (ATen also takes the second form for 3-channel inputs when torch runs with ONE thread; the reference's host has more.)
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct AxisTap { int i0, i1; float l0, l1; };

__device__ __forceinline__ AxisTap axis_tap(int dst, int in, int out, float scale)
{
    AxisTap t;
    if (in == out) { t.i0 = t.i1 = dst; t.l0 = 1.0f; t.l1 = 0.0f; return t; }
    const float src = scale * (float)dst;
    t.i0 = min((int)src, in - 1);                       // src >= 0: trunc == floor
    t.l1 = fminf(fmaxf(src - (float)t.i0, 0.0f), 1.0f);
    t.l0 = 1.0f - t.l1;
    t.i1 = t.i0 + ((t.i0 < in - 1) ? 1 : 0);
    return t;
}

__device__ __forceinline__ float bilerp_ac(float v00, float v01, float v10, float v11, const AxisTap &ty, const AxisTap &tx, bool small)
{
    if (small) {
        float acc = v01 * (ty.l0 * tx.l1);
        acc = fmaf(v00, ty.l0 * tx.l0, acc);
        acc = fmaf(v10, ty.l1 * tx.l0, acc);
        return fmaf(v11, ty.l1 * tx.l1, acc);
    }
    const float t0 = fmaf(v00, tx.l0, v01 * tx.l1);
    const float t1 = fmaf(v10, tx.l0, v11 * tx.l1);
    return fmaf(t0, ty.l0, t1 * ty.l1);
}

__global__ __launch_bounds__(256) void k_prepare_inputs(const uint8_t *__restrict__ rgb, const uint8_t *__restrict__ disp,
                                                       const uint8_t *__restrict__ ids, int obj_index, int h, int w, int H, int W,
                                                       float sy, float sx, float *__restrict__ image_out, float *__restrict__ disp_out,
                                                       float *__restrict__ mask_out)
{
    const int64_t N = (int64_t)H * W, n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int Y = (int)(n / W), X = (int)(n - (int64_t)Y * W);
    const AxisTap ty = axis_tap(Y, h, H, sy), tx = axis_tap(X, w, W, sx);
    const bool small = (H + W) <= 128;
    const int64_t o00 = (int64_t)ty.i0 * w + tx.i0, o01 = (int64_t)ty.i0 * w + tx.i1, o10 = (int64_t)ty.i1 * w + tx.i0, o11 = (int64_t)ty.i1 * w + tx.i1;
    if (rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            image_out[(int64_t)c * N + n] = bilerp_ac((float)rgb[3 * o00 + c] / 255.0f, (float)rgb[3 * o01 + c] / 255.0f,
                                                      (float)rgb[3 * o10 + c] / 255.0f, (float)rgb[3 * o11 + c] / 255.0f, ty, tx, small);
    }
    if (disp)
        disp_out[n] = bilerp_ac((float)((double)disp[o00] / 255.0), (float)((double)disp[o01] / 255.0), (float)((double)disp[o10] / 255.0),
                                (float)((double)disp[o11] / 255.0), ty, tx, small);
    if (ids)
        mask_out[n] = bilerp_ac(ids[o00] == obj_index ? 1.0f : 0.0f, ids[o01] == obj_index ? 1.0f : 0.0f, ids[o10] == obj_index ? 1.0f : 0.0f,
                                ids[o11] == obj_index ? 1.0f : 0.0f, ty, tx, small);
}

}   // namespace

extern "C" int mpf_prepare_inputs(const uint8_t *d_rgb_u8, const uint8_t *d_disp_u8, const uint8_t *d_ids_u8, int obj_index, int h, int w,
                                  int H, int W, float *d_image, float *d_disp, float *d_mask, void *stream)
{
    MPF_REQUIRE(h >= 1 && w >= 1 && H >= 1 && W >= 1 && (int64_t)h * w < ((int64_t)1 << 31) && (int64_t)H * W < ((int64_t)1 << 31), "mpf_prepare_inputs: bad shape");
    MPF_REQUIRE((d_rgb_u8 == nullptr) == (d_image == nullptr) && (d_disp_u8 == nullptr) == (d_disp == nullptr) && (d_ids_u8 == nullptr) == (d_mask == nullptr),
                "mpf_prepare_inputs: every input goes with its output");
    MPF_REQUIRE(d_rgb_u8 || d_disp_u8 || d_ids_u8, "mpf_prepare_inputs: nothing to do");
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f;       // area_pixel_compute_scale, align_corners = True
    const float sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f;
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_prepare_inputs, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_rgb_u8, d_disp_u8, d_ids_u8,
                       obj_index, h, w, H, W, sy, sx, d_image, d_disp, d_mask);
    return mpf_launch_status("k_prepare_inputs");
}
