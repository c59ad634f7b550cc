// mpf_generic.hip - materialised-tensor kernels behind the individual utils/mpi and geometry.py function signatures,
// plus library-wide plumbing (error string, device info).
//
// These exist so that each reference function on the path (HomographySample.sample / .sample_inverse,
// get_src_xyz_from_plane_disparity, transform_G_xyz, plane_volume_rendering[_flow], BackprojectDepth, Project3D) is a
// drop-in on its own.  They are straightforward streaming kernels; the fused kernels in mpf_render.hip are the fast
// path that the pipeline entry points use.
#include <stdarg.h>
#include <string.h>
#include "mpf_common.h"
#include "mpf_math.h"

// ---- plumbing ---------------------------------------------------------------------------------------------------

static thread_local char g_err[512] = "";

void mpf_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *mpf_last_error(void) { return g_err; }
extern "C" int mpf_version(void) { return MPF_VERSION; }

// A HIP stream whose kernels may only be placed on a SUBSET of the compute units (hipExtStreamCreateWithCUMask): every `stride`-th CU of the
// device.  For latency-sized side work that runs underneath a chip-filling launch on another stream (the moving-object chain beside the pair
// launch): its workgroups then compete for slots on few CUs instead of taking a slot here and there on all of them.
extern "C" int mpf_stream_create_cu_subset(int stride, int offset, void **out_stream)
{
    // stride < 0: the COMPLEMENT - every CU except those of (|stride|, offset) - for the stream of the chip-filling launch, so that the side work's
    // CUs are its own (no workgroup-slot or issue-slot sharing at all)
    const bool invert = stride < 0;
    if (invert) stride = -stride;
    MPF_REQUIRE(out_stream && stride >= 1 && offset >= 0 && offset < stride, "mpf_stream_create_cu_subset: bad argument");
    int dev = 0;
    MPF_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    MPF_HIP(hipGetDeviceProperties(&p, dev));
    const int ncu = p.multiProcessorCount;
    uint32_t mask[32];
    memset(mask, 0, sizeof(mask));
    MPF_REQUIRE(ncu <= 1024, "mpf_stream_create_cu_subset: more compute units than the mask holds");
    int n = 0;
    for (int c = 0; c < ncu; ++c)
        if (((c % stride) == offset) != invert) { mask[c >> 5] |= 1u << (c & 31); ++n; }
    MPF_REQUIRE(n >= 1, "mpf_stream_create_cu_subset: empty CU set");
    hipStream_t st = nullptr;
    MPF_HIP(hipExtStreamCreateWithCUMask(&st, (uint32_t)((ncu + 31) / 32), mask));
    *out_stream = (void *)st;
    return 0;
}

extern "C" int mpf_stream_destroy(void *stream)
{
    MPF_REQUIRE(stream, "mpf_stream_destroy: null stream");
    MPF_HIP(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

extern "C" int mpf_device_info(int device, int *cu_count, size_t *hbm_bytes, char *arch, size_t arch_len)
{
    hipDeviceProp_t p;
    MPF_HIP(hipGetDeviceProperties(&p, device));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    if (arch && arch_len) {
        strncpy(arch, p.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return 0;
}

// ---- get_src_xyz_from_plane_disparity  (utils/mpi/mpi_rendering.py:213-239) -------------------------------------

__global__ void __launch_bounds__(256)
k_src_xyz(const float *__restrict__ params, int S, int H, int W, float *__restrict__ xyz)
{
    const int64_t N = (int64_t)H * W;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float fx = (float)(n % W), fy = (float)(n / W);
    const float rx = mpf_row3_xy1(params[0], params[1], params[2], fx, fy);
    const float ry = mpf_row3_xy1(params[3], params[4], params[5], fx, fy);
    const float rz = mpf_row3_xy1(params[6], params[7], params[8], fx, fy);
    for (int s = 0; s < S; ++s) {
        const float d = params[MPF_PARAMS_HEADER + MPF_PLANE_RECORD * s + 9];
        float *o = xyz + (int64_t)s * 3 * N + n;
        o[0] = rx * d; o[N] = ry * d; o[2 * N] = rz * d;
    }
}

extern "C" int mpf_src_xyz(const float *d_params, int S, int H, int W, float *d_xyz, void *stream)
{
    MPF_REQUIRE(d_params && d_xyz && S >= 1 && H >= 1 && W >= 1, "mpf_src_xyz: bad argument");
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_src_xyz, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_params, S, H, W, d_xyz);
    return mpf_launch_status("k_src_xyz");
}

// ---- transform_G_xyz  (utils/mpi/rendering_utils.py:4-23) ------------------------------------------------------

__global__ void __launch_bounds__(256)
k_transform_xyz(const float *__restrict__ params, const float *__restrict__ xyz, int S, int64_t N, float *__restrict__ out)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    if (n >= N) return;
    const float *p = xyz + (int64_t)s * 3 * N + n;
    const float X = p[0], Y = p[N], Z = p[2 * N];
    float *o = out + (int64_t)s * 3 * N + n;
    o[0] = mpf_row4_xyz1(params[9], params[10], params[11], params[12], X, Y, Z);
    o[N] = mpf_row4_xyz1(params[13], params[14], params[15], params[16], X, Y, Z);
    o[2 * N] = mpf_row4_xyz1(params[17], params[18], params[19], params[20], X, Y, Z);
}

extern "C" int mpf_transform_xyz(const float *d_params, const float *d_xyz, int S, int64_t N, float *d_out, void *stream)
{
    MPF_REQUIRE(d_params && d_xyz && d_out && S >= 1 && S < 65536 && N >= 1, "mpf_transform_xyz: bad argument");
    hipLaunchKernelGGL(k_transform_xyz, dim3((unsigned)((N + 255) / 256), S), dim3(256), 0, (hipStream_t)stream, d_params, d_xyz, S, N, d_out);
    return mpf_launch_status("k_transform_xyz");
}

// ---- HomographySample.sample  (utils/mpi/homography_sampler.py:124-158) -----------------------------------------

__global__ void __launch_bounds__(256)
k_homography_sample(const float *__restrict__ src, const float *__restrict__ params, int S, int C, int H, int W,
                    float *__restrict__ tgt, uint8_t *__restrict__ valid, float *__restrict__ flowB2A)
{
    const int64_t N = (int64_t)H * W;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    if (n >= N) return;
    const float fx = (float)(n % W), fy = (float)(n / W);
    const float *rec = params + MPF_PARAMS_HEADER + MPF_PLANE_RECORD * s;
    float qx = mpf_row3_xy1(rec[0], rec[1], rec[2], fx, fy);
    float qy = mpf_row3_xy1(rec[3], rec[4], rec[5], fx, fy);
    float qz = mpf_row3_xy1(rec[6], rec[7], rec[8], fx, fy);
    float u = qx / qz, v = qy / qz;
    if (flowB2A) {
        flowB2A[((int64_t)s * N + n) * 2] = u - fx;
        flowB2A[((int64_t)s * N + n) * 2 + 1] = v - fy;
    }
    if (valid) valid[(int64_t)s * N + n] = ((u < (float)W) && (u > -1.0f) && (v < (float)H) && (v > -1.0f)) ? 1 : 0;
    MpfTaps t = mpf_make_taps(u, v, W, H);
    const int o00 = t.y0 * W + t.x0;
    const int o01 = o00 + (t.e_in ? 1 : 0);
    const int o10 = o00 + (t.s_in ? W : 0);
    const int o11 = o10 + (t.e_in ? 1 : 0);
    for (int c = 0; c < C; ++c) {
        const float *pl = src + ((int64_t)s * C + c) * N;
        float a = pl[o00];
        float b = t.e_in ? pl[o01] : 0.0f;
        float cc = t.s_in ? pl[o10] : 0.0f;
        float d = (t.e_in && t.s_in) ? pl[o11] : 0.0f;
        tgt[((int64_t)s * C + c) * N + n] = mpf_bilerp(t, a, b, cc, d);
    }
}

extern "C" int mpf_homography_sample(const float *d_src, const float *d_params, int S, int C, int H, int W, float *d_tgt,
                                     uint8_t *d_valid, float *d_flowB2A, void *stream)
{
    MPF_REQUIRE(d_src && d_params && d_tgt && S >= 1 && S < 65536 && C >= 1 && H >= 1 && W >= 1, "mpf_homography_sample: bad argument");
    MPF_REQUIRE((int64_t)H * W < ((int64_t)1 << 31), "mpf_homography_sample: H*W too large");
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_homography_sample, dim3((unsigned)((N + 255) / 256), S), dim3(256), 0, (hipStream_t)stream, d_src,
                       d_params, S, C, H, W, d_tgt, d_valid, d_flowB2A);
    return mpf_launch_status("k_homography_sample");
}

// ---- HomographySample.sample_inverse  (utils/mpi/homography_sampler.py:197-218) ---------------------------------

__global__ void __launch_bounds__(256)
k_homography_flow(const float *__restrict__ params, int S, int H, int W, float *__restrict__ flow)
{
    const int64_t N = (int64_t)H * W;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int s = blockIdx.y;
    if (n >= N) return;
    const float fx = (float)(n % W), fy = (float)(n / W);
    const float *rec = params + MPF_PARAMS_HEADER + MPF_PLANE_RECORD * s;
    float qx = mpf_row3_xy1(rec[0], rec[1], rec[2], fx, fy);
    float qy = mpf_row3_xy1(rec[3], rec[4], rec[5], fx, fy);
    float qz = mpf_row3_xy1(rec[6], rec[7], rec[8], fx, fy);
    reinterpret_cast<float2 *>(flow)[(int64_t)s * N + n] = make_float2(qx / qz - fx, qy / qz - fy);
}

extern "C" int mpf_homography_flow(const float *d_params, int S, int H, int W, float *d_flow, void *stream)
{
    MPF_REQUIRE(d_params && d_flow && S >= 1 && S < 65536 && H >= 1 && W >= 1, "mpf_homography_flow: bad argument");
    MPF_REQUIRE((((uintptr_t)d_flow) & 7) == 0, "mpf_homography_flow: output must be 8-byte aligned");
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_homography_flow, dim3((unsigned)((N + 255) / 256), S), dim3(256), 0, (hipStream_t)stream, d_params, S, H, W, d_flow);
    return mpf_launch_status("k_homography_flow");
}

// ---- plane_volume_rendering / _flow / weighted_sum_mpi  (utils/mpi/mpi_rendering.py:62-154) ---------------------

template <int NL>
__global__ void __launch_bounds__(256)
k_volume_render(const float *__restrict__ rgb, const float *__restrict__ sigma, const float *__restrict__ xyz, int S,
                int64_t N, float *__restrict__ rgb_out, float *__restrict__ depth_out, float *__restrict__ tacc_out,
                float *__restrict__ weights_out, const float *__restrict__ extra_in, int E, float *__restrict__ extra_out,
                int hard)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double acc = 1.0;
    MpfCsum<NL> cw, cd, cc[3], ce[4];
    cw.init(); cd.init();
#pragma unroll
    for (int c = 0; c < 3; ++c) cc[c].init();
#pragma unroll
    for (int e = 0; e < 4; ++e) ce[e].init();
    float best_w = 0.0f;
    int best_s = 0;
    for (int s = 0; s < S; ++s) {
        const float *p = xyz + (int64_t)s * 3 * N + n;
        const float X = p[0], Y = p[N], Z = p[2 * N];
        float dist = 1e3f;
        if (s + 1 < S) {
            const float *q = p + 3 * N;
            dist = mpf_norm3(q[0] - X, q[N] - Y, q[2 * N] - Z);
        }
        float T = mpf_expf(-sigma[(int64_t)s * N + n] * dist);
        float alpha = 1.0f - T;
        float tacc = (float)acc;
        float w = tacc * alpha;
        acc *= (double)(T + 1e-6f);
        if (tacc_out) tacc_out[(int64_t)s * N + n] = tacc;
        if (weights_out) weights_out[(int64_t)s * N + n] = w;
        if (s == 0 || w > best_w) { best_w = w; best_s = s; }     // torch.max: first maximal index
        cw.push(w);
        if (rgb) {
#pragma unroll
            for (int c = 0; c < 3; ++c) cc[c].push(w * rgb[((int64_t)s * 3 + c) * N + n]);
        }
        cd.push(w * Z);
        if (!hard) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < E) ce[e].push(w * extra_in[((int64_t)s * E + e) * N + n]);
        }
        if (((s + 1) & 15) == 0) {
            cw.fold(s + 1); cd.fold(s + 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) cc[c].fold(s + 1);
#pragma unroll
            for (int e = 0; e < 4; ++e) ce[e].fold(s + 1);
        }
    }
    if (rgb_out && rgb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb_out[c * N + n] = cc[c].final();
    }
    if (depth_out) depth_out[n] = cd.final() / (cw.final() + 1e-5f);
    if (extra_out) {
        for (int e = 0; e < E && e < 4; ++e)
            // hard_flow (:126-130): sum of one-hot(argmax w) * flow == the arg-max plane's value (other addends are 0)
            extra_out[e * N + n] = hard ? extra_in[((int64_t)best_s * E + e) * N + n] : ce[e].final();
    }
}

extern "C" int mpf_volume_render(const float *d_rgb, const float *d_sigma, const float *d_xyz, int S, int64_t N,
                                 float *d_rgb_out, float *d_depth_out, float *d_tacc_out, float *d_weights_out,
                                 const float *d_extra_in, int E, float *d_extra_out, int hard, void *stream)
{
    MPF_REQUIRE(d_sigma && d_xyz && S >= 1 && S < 4096 && N >= 1, "mpf_volume_render: bad argument");
    MPF_REQUIRE(E >= 0 && E <= 4 && ((E == 0) || (d_extra_in && d_extra_out)), "mpf_volume_render: extra channels: 0..4 with both pointers");
    dim3 grid((unsigned)((N + 255) / 256)), block(256);
    if (S < 256)
        hipLaunchKernelGGL(k_volume_render<2>, grid, block, 0, (hipStream_t)stream, d_rgb, d_sigma, d_xyz, S, N, d_rgb_out,
                           d_depth_out, d_tacc_out, d_weights_out, d_extra_in, E, d_extra_out, hard);
    else
        hipLaunchKernelGGL(k_volume_render<3>, grid, block, 0, (hipStream_t)stream, d_rgb, d_sigma, d_xyz, S, N, d_rgb_out,
                           d_depth_out, d_tacc_out, d_weights_out, d_extra_in, E, d_extra_out, hard);
    return mpf_launch_status("k_volume_render");
}

// ---- weighted_sum_mpi with caller-supplied weights  (utils/mpi/mpi_rendering.py:142-154) ---------------------------

template <int NL>
__global__ void __launch_bounds__(256)
k_weighted_sum(const float *__restrict__ weights, const float *__restrict__ values, int S, int C, int64_t N,
               float *__restrict__ out)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (n >= N) return;
    MpfCsum<NL> acc;
    acc.init();
    for (int s = 0; s < S; ++s) {
        const float w = weights[(int64_t)s * N + n];
        acc.push(values ? w * values[((int64_t)s * C + c) * N + n] : w);   // torch.sum(weights * v, dim=1) / torch.sum(weights, dim=1)
        if (((s + 1) & 15) == 0) acc.fold(s + 1);
    }
    out[(int64_t)c * N + n] = acc.final();
}

extern "C" int mpf_weighted_sum(const float *d_weights, const float *d_values, int S, int C, int64_t N, float *d_out, void *stream)
{
    MPF_REQUIRE(d_weights && d_out && S >= 1 && S < 4096 && C >= 1 && C < 65536 && N >= 1, "mpf_weighted_sum: bad argument");
    dim3 grid((unsigned)((N + 255) / 256), C), block(256);
    if (S < 256)
        hipLaunchKernelGGL(k_weighted_sum<2>, grid, block, 0, (hipStream_t)stream, d_weights, d_values, S, C, N, d_out);
    else
        hipLaunchKernelGGL(k_weighted_sum<3>, grid, block, 0, (hipStream_t)stream, d_weights, d_values, S, C, N, d_out);
    return mpf_launch_status("k_weighted_sum");
}

// ---- alpha_composition and the use_alpha blend weights (utils/mpi/mpi_rendering.py:42-59, :36) ------------------------------
template <int NL>
__global__ void __launch_bounds__(256)
k_alpha_composite(const float *__restrict__ alpha, const float *__restrict__ values, int S, int C, int64_t N, float *__restrict__ out,
                  float *__restrict__ weights, float *__restrict__ cumprod_eps)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (n >= N) return;
    double keep = 1.0, keep_eps = 1.0;                       // torch.cumprod accumulates float tensors in double on the CPU
    MpfCsum<NL> acc;
    acc.init();
    for (int s = 0; s < S; ++s) {
        const float a = alpha[(int64_t)s * N + n];
        const float w = a * (float)keep;                     // alpha * [1, cumprod(1 - alpha)[:-1]]
        keep *= (double)(1.0f - a);
        if (c == 0) {
            if (weights) weights[(int64_t)s * N + n] = w;
            if (cumprod_eps) {
                keep_eps *= (double)((1.0f - a) + 1e-6f);
                cumprod_eps[(int64_t)s * N + n] = (float)keep_eps;
            }
        }
        if (values) {
            acc.push(values[((int64_t)s * C + c) * N + n] * w);
            if (((s + 1) & 15) == 0) acc.fold(s + 1);
        }
    }
    if (values && out) out[(int64_t)c * N + n] = acc.final();
}

extern "C" int mpf_alpha_composite(const float *d_alpha, const float *d_values, int S, int C, int64_t N, float *d_out, float *d_weights,
                                   float *d_cumprod_eps, void *stream)
{
    MPF_REQUIRE(d_alpha && S >= 1 && S < 4096 && N >= 1, "mpf_alpha_composite: bad argument");
    MPF_REQUIRE((d_values == nullptr) == (d_out == nullptr) && (!d_values || (C >= 1 && C < 65536)), "mpf_alpha_composite: values and out go together");
    dim3 grid((unsigned)((N + 255) / 256), d_values ? C : 1), block(256);
    if (S < 256)
        hipLaunchKernelGGL(k_alpha_composite<2>, grid, block, 0, (hipStream_t)stream, d_alpha, d_values, S, C, N, d_out, d_weights, d_cumprod_eps);
    else
        hipLaunchKernelGGL(k_alpha_composite<3>, grid, block, 0, (hipStream_t)stream, d_alpha, d_values, S, C, N, d_out, d_weights, d_cumprod_eps);
    return mpf_launch_status("k_alpha_composite");
}

// ---- BackprojectDepth + Project3D  (geometry.py:41-49, :63-76) --------------------------------------------------

struct MpfProj { float ik[9]; float P[12]; };

__global__ void __launch_bounds__(256)
k_backproject_project(const float *__restrict__ depth, MpfProj m, int H, int W, float *__restrict__ pix, float *__restrict__ z)
{
    const int64_t N = (int64_t)H * W;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float fx = (float)(n % W), fy = (float)(n / W);
    const float dep = depth[n];
    float cam[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) cam[c] = dep * mpf_row3_xy1(m.ik[3 * c], m.ik[3 * c + 1], m.ik[3 * c + 2], fx, fy);
    float q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = mpf_row4_xyz1(m.P[4 * c], m.P[4 * c + 1], m.P[4 * c + 2], m.P[4 * c + 3], cam[0], cam[1], cam[2]);
    const float den = q[2] + 1e-7f;
    float px = q[0] / den, py = q[1] / den;
    px = px / (float)(W - 1);
    py = py / (float)(H - 1);
    reinterpret_cast<float2 *>(pix)[n] = make_float2((px - 0.5f) * 2.0f, (py - 0.5f) * 2.0f);
    z[n] = q[2];
}

extern "C" int mpf_backproject_project(const float *d_depth, const float *h_inv_k9, const float *h_P12, int H, int W,
                                       float *d_pix, float *d_z, void *stream)
{
    MPF_REQUIRE(d_depth && h_inv_k9 && h_P12 && d_pix && d_z && H >= 1 && W >= 1, "mpf_backproject_project: bad argument");
    MPF_REQUIRE((((uintptr_t)d_pix) & 7) == 0, "mpf_backproject_project: pix must be 8-byte aligned");
    MpfProj m;
    memcpy(m.ik, h_inv_k9, sizeof(m.ik));
    memcpy(m.P, h_P12, sizeof(m.P));
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_backproject_project, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_depth, m, H, W, d_pix, d_z);
    return mpf_launch_status("k_backproject_project");
}

// ---- BackprojectDepth alone (geometry.py:41-49) and Project3D alone (geometry.py:63-76) -------------------------

__global__ void __launch_bounds__(256)
k_backproject(const float *__restrict__ depth, MpfProj m, int H, int W, float *__restrict__ cam)
{
    const int64_t N = (int64_t)H * W;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float fx = (float)(n % W), fy = (float)(n / W);
    const float dep = depth[n];
#pragma unroll
    for (int c = 0; c < 3; ++c) cam[c * N + n] = dep * mpf_row3_xy1(m.ik[3 * c], m.ik[3 * c + 1], m.ik[3 * c + 2], fx, fy);
    cam[3 * N + n] = 1.0f;
}

extern "C" int mpf_backproject(const float *d_depth, const float *h_inv_k9, int H, int W, float *d_cam_points, void *stream)
{
    MPF_REQUIRE(d_depth && h_inv_k9 && d_cam_points && H >= 1 && W >= 1, "mpf_backproject: bad argument");
    MpfProj m;
    memcpy(m.ik, h_inv_k9, sizeof(m.ik));
    memset(m.P, 0, sizeof(m.P));
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_backproject, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_depth, m, H, W, d_cam_points);
    return mpf_launch_status("k_backproject");
}

__global__ void __launch_bounds__(256)
k_project3d(const float *__restrict__ pts, MpfProj m, float eps, int H, int W, float *__restrict__ pix, float *__restrict__ z)
{
    const int64_t N = (int64_t)H * W;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float X = pts[n], Y = pts[N + n], Z = pts[2 * N + n], Wh = pts[3 * N + n];
    float q[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float a = m.P[4 * c] * X;
        a = fmaf(m.P[4 * c + 1], Y, a);
        a = fmaf(m.P[4 * c + 2], Z, a);
        a = fmaf(m.P[4 * c + 3], Wh, a);
        q[c] = a;
    }
    const float den = q[2] + eps;
    float px = q[0] / den, py = q[1] / den;
    px = px / (float)(W - 1);
    py = py / (float)(H - 1);
    reinterpret_cast<float2 *>(pix)[n] = make_float2((px - 0.5f) * 2.0f, (py - 0.5f) * 2.0f);
    z[n] = q[2];
}

extern "C" int mpf_project3d(const float *d_points_4N, const float *h_P12, float eps, int H, int W, float *d_pix, float *d_z, void *stream)
{
    MPF_REQUIRE(d_points_4N && h_P12 && d_pix && d_z && H >= 1 && W >= 1, "mpf_project3d: bad argument");
    MPF_REQUIRE((((uintptr_t)d_pix) & 7) == 0, "mpf_project3d: pix must be 8-byte aligned");
    MpfProj m;
    memset(m.ik, 0, sizeof(m.ik));
    memcpy(m.P, h_P12, sizeof(m.P));
    const int64_t N = (int64_t)H * W;
    hipLaunchKernelGGL(k_project3d, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_points_4N, m, eps, H, W, d_pix, d_z);
    return mpf_launch_status("k_project3d");
}
