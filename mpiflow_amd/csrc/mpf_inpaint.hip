// mpf_inpaint.hip - host-side hole filling with the reference's algorithms: OpenCV's cv2.inpaint, INPAINT_NS and INPAINT_TELEA.
//
// The reference finishes every pair on the HOST: cv2.inpaint(frame_mix, fill_mask, 3, cv2.INPAINT_NS) (utils/utils.py:284-286)
// and cv2.inpaint(im1_raw, 1 - H, 3, cv2.INPAINT_TELEA) (moving_obj.py:162).  Both are fast-marching methods: pixels are filled
// one at a time in the order a priority queue releases them (arrival time T, ties first-in-first-out), and each fill reads
// pixels filled before it - an inherently sequential front, so this step stays on the host here as well and runs on the
// generator's writer threads, overlapped with the GPU render of the following pairs (one call per frame, re-entrant, no
// globals; ctypes releases the GIL).  This file contains no device code; it lives in libmpiflow_hip.so because that is the
// C ABI the Python layer binds.
//
// What is implemented is OpenCV's published algorithm (modules/photo/src/inpaint.cpp; A. Telea 2004 for TELEA, the
// FMM-ordered isophote-weighted average OpenCV calls Navier-Stokes for NS) with OpenCV's order of operations and mix of
// float / double / int arithmetic, so the bytes are meant to equal cv2's.  OpenCV is third-party and not installed in
// the build image: parity with cv2 itself is UNPINNED until tests/test_inpaint.py has run next to a real cv2 (DESIGN.md
// section 7); it is checked byte for byte against the independent plain-C restatement in oracle/ by the tests.
//
// Differences in form (not in result) from OpenCV's code:
//   * the front queue is a binary heap ordered by (T, push sequence number) instead of a sorted linked list: the same
//     pop order (ascending T, FIFO among equal T) at O(log n) per operation instead of O(n)
//   * Telea's per-neighbour weight does not depend on the colour channel; it is computed once and applied to the channels
//     (OpenCV recomputes it per channel with identical operands) - each channel's sums still accumulate in the same order
//   * the distance weights depend on the offset only and are tabulated once per call
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "mpf_common.h"

namespace {

enum : uint8_t { KNOWN = 0, BAND = 1, INSIDE = 2, CHANGE = 3 };

struct FrontItem { float T; uint32_t seq; int32_t i, j; };
struct FrontLater {      // std heap functions build a max-heap: "less" = released later
    bool operator()(const FrontItem &a, const FrontItem &b) const { return a.T > b.T || (a.T == b.T && a.seq > b.seq); }
};
class Front {
    std::vector<FrontItem> h;
    uint32_t seq = 0;
public:
    void push(int i, int j, float T) { h.push_back(FrontItem{T, seq++, i, j}); std::push_heap(h.begin(), h.end(), FrontLater()); }
    bool pop(int &i, int &j)
    {
        if (h.empty()) return false;
        std::pop_heap(h.begin(), h.end(), FrontLater());
        i = h.back().i; j = h.back().j;
        h.pop_back();
        return true;
    }
};

struct Field {                    // the 1-pixel padded flag / arrival-time planes ("extended" coordinates = image + 1)
    int er, ec;
    std::vector<uint8_t> f;
    std::vector<float> t;
    uint8_t &F(int i, int j) { return f[(size_t)i * ec + j]; }
    float &T(int i, int j) { return t[(size_t)i * ec + j]; }
};

inline float eikonal2(Field &g, const std::vector<uint8_t> &flags, int i1, int j1, int i2, int j2)
{
    const double a11 = g.T(i1, j1), a22 = g.T(i2, j2), m12 = std::min(a11, a22);
    const bool in1 = flags[(size_t)i1 * g.ec + j1] == INSIDE, in2 = flags[(size_t)i2 * g.ec + j2] == INSIDE;
    double sol;
    if (!in1) {
        if (!in2) sol = (fabs(a11 - a22) >= 1.0) ? 1 + m12 : (a11 + a22 + sqrt((double)(2 - (a11 - a22) * (a11 - a22)))) * 0.5;
        else sol = 1 + a11;
    } else {
        sol = !in2 ? 1 + a22 : 1 + m12;
    }
    return (float)sol;
}

inline float arrival(Field &g, const std::vector<uint8_t> &flags, int i, int j)
{
    const float a = eikonal2(g, flags, i - 1, j, i, j - 1), b = eikonal2(g, flags, i + 1, j, i, j - 1);
    const float c = eikonal2(g, flags, i - 1, j, i, j + 1), d = eikonal2(g, flags, i + 1, j, i, j + 1);
    return std::min(std::min(a, b), std::min(c, d));
}

// grey dilation with a (2r+1)-cross (r = 1 only) or a (2r+1)^2 box; pixels outside the array never win.  The box is separable:
// a running maximum along the rows, then along the columns.
void dilate(const std::vector<uint8_t> &src, std::vector<uint8_t> &dst, int er, int ec, int r, bool cross)
{
    if (cross) {
        for (int i = 0; i < er; ++i)
            for (int j = 0; j < ec; ++j) {
                uint8_t m = src[(size_t)i * ec + j];
                if (i > 0) m = std::max(m, src[(size_t)(i - 1) * ec + j]);
                if (i + 1 < er) m = std::max(m, src[(size_t)(i + 1) * ec + j]);
                if (j > 0) m = std::max(m, src[(size_t)i * ec + j - 1]);
                if (j + 1 < ec) m = std::max(m, src[(size_t)i * ec + j + 1]);
                dst[(size_t)i * ec + j] = m;
            }
        return;
    }
    std::vector<uint8_t> tmp((size_t)er * ec);
    for (int i = 0; i < er; ++i)
        for (int j = 0; j < ec; ++j) {
            uint8_t m = 0;
            for (int x = std::max(j - r, 0); x <= std::min(j + r, ec - 1); ++x) m = std::max(m, src[(size_t)i * ec + x]);
            tmp[(size_t)i * ec + j] = m;
        }
    for (int i = 0; i < er; ++i)
        for (int j = 0; j < ec; ++j) {
            uint8_t m = 0;
            for (int y = std::max(i - r, 0); y <= std::min(i + r, er - 1); ++y) m = std::max(m, tmp[(size_t)y * ec + j]);
            dst[(size_t)i * ec + j] = m;
        }
}

void zero_border(std::vector<uint8_t> &a, int er, int ec)
{
    for (int j = 0; j < ec; ++j) a[j] = a[(size_t)(er - 1) * ec + j] = 0;
    for (int i = 0; i < er; ++i) a[(size_t)i * ec] = a[(size_t)i * ec + ec - 1] = 0;
}

inline uint8_t sat8(long v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

struct Offset { int dk, dl; float w_ns, w_telea; };     // neighbourhood offsets within the radius, raster order, with their distance weights

template <int C>
void fill(const uint8_t *img, const uint8_t *mask_in, int rows, int cols, int range, int method, uint8_t *out)
{
    Field g;
    g.er = rows + 2; g.ec = cols + 2;
    const int er = g.er, ec = g.ec;
    const size_t en = (size_t)er * ec;
    memcpy(out, img, (size_t)rows * cols * C);
    std::vector<uint8_t> hole(en, KNOWN), band(en, 0);
    size_t nhole = 0;
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j)
            if (mask_in[(size_t)i * cols + j]) { hole[(size_t)(i + 1) * ec + j + 1] = INSIDE; ++nhole; }
    if (!nhole) return;
    g.t.assign(en, 1.0e6f);
    dilate(hole, band, er, ec, 1, true);
    for (size_t n = 0; n < en; ++n) band[n] = (uint8_t)(band[n] > hole[n] ? band[n] - hole[n] : 0);
    zero_border(band, er, ec);
    Front front;
    for (int i = 0; i < er; ++i)
        for (int j = 0; j < ec; ++j)
            if (band[(size_t)i * ec + j]) { front.push(i, j, 0.0f); g.t[(size_t)i * ec + j] = 0.0f; }

    if (method == MPF_INPAINT_TELEA) {          // signed distance: negative arrival times in the ring outside the hole
        std::vector<uint8_t> ring(en, 0);
        dilate(hole, ring, er, ec, range, false);
        Front outer;
        for (int i = 0; i < er; ++i)
            for (int j = 0; j < ec; ++j)
                if (band[(size_t)i * ec + j]) outer.push(i, j, 0.0f);
        for (size_t n = 0; n < en; ++n) {
            const uint8_t v = (uint8_t)(ring[n] > hole[n] ? ring[n] - hole[n] : 0);
            ring[n] = (uint8_t)(v > band[n] ? v - band[n] : 0);
        }
        zero_border(ring, er, ec);
        int ii, jj;
        while (outer.pop(ii, jj)) {
            ring[(size_t)ii * ec + jj] = CHANGE;
            const int ni[4] = {ii - 1, ii, ii + 1, ii}, nj[4] = {jj, jj - 1, jj, jj + 1};
            for (int q = 0; q < 4; ++q) {
                const int i = ni[q], j = nj[q];
                if (i <= 0 || j <= 0 || i > er - 1 || j > ec - 1 || ring[(size_t)i * ec + j] != INSIDE) continue;
                const float d = arrival(g, ring, i, j);
                g.T(i, j) = d;
                ring[(size_t)i * ec + j] = BAND;
                outer.push(i, j, d);
            }
        }
        for (size_t n = 0; n < en; ++n)
            if (ring[n] == CHANGE) g.t[n] = -g.t[n];
    }

    std::vector<Offset> offs;
    for (int dk = -range; dk <= range; ++dk)
        for (int dl = -range; dl <= range; ++dl) {
            if (dk * dk + dl * dl > range * range) continue;
            Offset o;
            o.dk = dk; o.dl = dl;
            const float len2 = (float)dl * (float)dl + (float)dk * (float)dk;        // same value for r and -r
            o.w_ns = 1 / (len2 * len2 + 1);
            o.w_telea = (dk || dl) ? (float)(1. / (len2 * sqrt((double)len2))) : 0.0f;
            offs.push_back(o);
        }
    std::vector<uint8_t> &f = hole;                  // KNOWN / INSIDE, BAND once filled: the flags the fill rules test
    const bool tiny = rows < 2 || cols < 2;                  // only then can OpenCV's index arithmetic leave the image (it reads out of bounds there)
    auto px = [&](int r, int c, int ch) -> int {
        if (__builtin_expect(tiny, 0)) {
            r = r < 0 ? 0 : (r > rows - 1 ? rows - 1 : r);
            c = c < 0 ? 0 : (c > cols - 1 ? cols - 1 : c);
        }
        return out[((size_t)r * cols + c) * C + ch];
    };
    // 3 channels: the fill rules work on a float copy of the image, one 16-byte vector per pixel, kept in step with `out`
    std::vector<v4f> fimg;
    if (C == 3) {
        fimg.resize((size_t)rows * cols);
        for (size_t n = 0; n < (size_t)rows * cols; ++n) fimg[n] = v4f{(float)out[3 * n], (float)out[3 * n + 1], (float)out[3 * n + 2], 0.0f};
    }
    auto px3 = [&](int r, int c) -> v4f {                                   // the pixel as floats (0..255: exact, differences and |.| too)
        if (__builtin_expect(tiny, 0)) {
            r = r < 0 ? 0 : (r > rows - 1 ? rows - 1 : r);
            c = c < 0 ? 0 : (c > cols - 1 ? cols - 1 : c);
        }
        return fimg[(size_t)r * cols + c];
    };
    auto vabs = [](v4f a) -> v4f { return __builtin_elementwise_abs(a); };
    const v4f vzero = {0.0f, 0.0f, 0.0f, 0.0f};
    int ii, jj;
    while (front.pop(ii, jj)) {
        f[(size_t)ii * ec + jj] = KNOWN;
        const int ni[4] = {ii - 1, ii, ii + 1, ii}, nj[4] = {jj, jj - 1, jj, jj + 1};
        for (int q = 0; q < 4; ++q) {
            const int i = ni[q], j = nj[q];
            if (i <= 0 || j <= 0 || i > er - 1 || j > ec - 1 || f[(size_t)i * ec + j] != INSIDE) continue;
            const float dist = arrival(g, f, i, j);
            g.T(i, j) = dist;
            auto inside = [&](int a, int b) { return f[(size_t)a * ec + b] == INSIDE; };
            if (method == MPF_INPAINT_TELEA) {
                float gTx, gTy;
                if (!inside(i, j + 1)) gTx = !inside(i, j - 1) ? (float)(g.T(i, j + 1) - g.T(i, j - 1)) * 0.5f : (float)(g.T(i, j + 1) - g.T(i, j));
                else gTx = !inside(i, j - 1) ? (float)(g.T(i, j) - g.T(i, j - 1)) : 0.0f;
                if (!inside(i + 1, j)) gTy = !inside(i - 1, j) ? (float)(g.T(i + 1, j) - g.T(i - 1, j)) * 0.5f : (float)(g.T(i + 1, j) - g.T(i, j));
                else gTy = !inside(i - 1, j) ? (float)(g.T(i, j) - g.T(i - 1, j)) : 0.0f;
                if constexpr (C == 3) {
                    // Telea's rule with the three channels of a neighbour in one vector (the weight is shared by the channels)
                    v4f Ia = vzero, Jx = vzero, Jy = vzero;
                    float sw = 1.0e-20f;
                    for (const Offset &o : offs) {
                        const int k = i + o.dk, l = j + o.dl;
                        if (!(k > 0 && l > 0 && k < er - 1 && l < ec - 1) || inside(k, l)) continue;
                        const int km = k - 1 + (k == 1), kp = k - 1 - (k == er - 2), lm = l - 1 + (l == 1), lp = l - 1 - (l == ec - 2);
                        const float ry = (float)(i - k), rx = (float)(j - l);
                        const float lev = (float)(1. / (1 + fabsf(g.T(k, l) - g.T(i, j))));
                        float dir = rx * gTx + ry * gTy;
                        if (fabsf(dir) <= 0.01) dir = 0.000001f;
                        const float w = fabsf(o.w_telea * lev * dir);
                        const bool e_in = inside(k, l + 1), w_in = inside(k, l - 1), s_in = inside(k + 1, l), n_in = inside(k - 1, l);
                        v4f gIx, gIy;
                        if (!e_in) gIx = !w_in ? (px3(km, lp + 1) - px3(km, lm - 1)) * 2.0f : px3(km, lp + 1) - px3(km, lm);
                        else gIx = !w_in ? px3(km, lp) - px3(km, lm - 1) : vzero;
                        if (!s_in) gIy = !n_in ? (px3(kp + 1, lm) - px3(km - 1, lm)) * 2.0f : px3(kp + 1, lm) - px3(km, lm);
                        else gIy = !n_in ? px3(kp, lm) - px3(km - 1, lm) : vzero;
                        Ia += w * px3(km, lm);
                        Jx -= w * (gIx * rx);
                        Jy -= w * (gIy * ry);
                        sw += w;
                    }
                    const v4f sat = Ia / sw + (Jx + Jy) / (__builtin_elementwise_sqrt(Jx * Jx + Jy * Jy) + 1.0e-20f) + 0.5f;
                    v4f filled = vzero;
                    for (int c = 0; c < 3; ++c) {
                        const uint8_t v = sat8(lrintf(sat[c]));
                        out[((size_t)(i - 1) * cols + (j - 1)) * 3 + c] = v;
                        filled[c] = (float)v;
                    }
                    fimg[(size_t)(i - 1) * cols + (j - 1)] = filled;
                } else {
                float Ia[C], Jx[C], Jy[C], s[C];
                for (int c = 0; c < C; ++c) { Ia[c] = Jx[c] = Jy[c] = 0.0f; s[c] = 1.0e-20f; }
                for (const Offset &o : offs) {
                    const int k = i + o.dk, l = j + o.dl;
                    if (!(k > 0 && l > 0 && k < er - 1 && l < ec - 1) || inside(k, l)) continue;
                    const int km = k - 1 + (k == 1), kp = k - 1 - (k == er - 2), lm = l - 1 + (l == 1), lp = l - 1 - (l == ec - 2);
                    const float ry = (float)(i - k), rx = (float)(j - l);
                    const float lev = (float)(1. / (1 + fabsf(g.T(k, l) - g.T(i, j))));
                    float dir = rx * gTx + ry * gTy;
                    if (fabsf(dir) <= 0.01) dir = 0.000001f;
                    const float w = fabsf(o.w_telea * lev * dir);
                    const bool e_in = inside(k, l + 1), w_in = inside(k, l - 1), s_in = inside(k + 1, l), n_in = inside(k - 1, l);
                    for (int c = 0; c < C; ++c) {
                        float gIx, gIy;
                        if (!e_in) gIx = !w_in ? (float)(px(km, lp + 1, c) - px(km, lm - 1, c)) * 2.0f : (float)(px(km, lp + 1, c) - px(km, lm, c));
                        else gIx = !w_in ? (float)(px(km, lp, c) - px(km, lm - 1, c)) : 0.0f;
                        if (!s_in) gIy = !n_in ? (float)(px(kp + 1, lm, c) - px(km - 1, lm, c)) * 2.0f : (float)(px(kp + 1, lm, c) - px(km, lm, c));
                        else gIy = !n_in ? (float)(px(kp, lm, c) - px(km - 1, lm, c)) : 0.0f;
                        Ia[c] += w * (float)px(km, lm, c);
                        Jx[c] -= w * (float)(gIx * rx);
                        Jy[c] -= w * (float)(gIy * ry);
                        s[c] += w;
                    }
                }
                for (int c = 0; c < C; ++c) {
                    const float sat = Ia[c] / s[c] + (Jx[c] + Jy[c]) / (sqrtf(Jx[c] * Jx[c] + Jy[c] * Jy[c]) + 1.0e-20f) + 0.5f;
                    out[((size_t)(i - 1) * cols + (j - 1)) * C + c] = sat8(lrintf(sat));
                }
                }
            } else if constexpr (C == 3) {
                // Navier-Stokes rule, the three channels of a neighbour in one 4-lane vector: the neighbour tests are shared, and the
                // division and the square root each channel needs per neighbour (what this loop spends its time on) become one divps
                // and one sqrtps.  Lane-wise IEEE operations in the scalar code's order, so the bytes are the same (tests/test_inpaint.py
                // holds it to the plain-C restatement): 18.0 -> 10.4 ms per 384 x 1280 frame with 33 500 hole pixels on the GPU box's host (tools/bench_inpaint_threads.py).
                v4f Ia = {0.0f, 0.0f, 0.0f, 0.0f}, sw = {1.0e-20f, 1.0e-20f, 1.0e-20f, 1.0e-20f};
                for (const Offset &o : offs) {
                    const int k = i + o.dk, l = j + o.dl;
                    if (!(k > 0 && l > 0 && k < er - 1 && l < ec - 1) || inside(k, l)) continue;
                    const int km = k - 1 + (k == 1), kp = k - 1 - (k == er - 2), lm = l - 1 + (l == 1), lp = l - 1 - (l == ec - 2);
                    const float ry = (float)(k - i), rx = (float)(l - j);
                    const float r2 = rx * rx + ry * ry;
                    const bool e_in = inside(k, l + 1), w_in = inside(k, l - 1), s_in = inside(k + 1, l), n_in = inside(k - 1, l);
                    const v4f ctr = px3(km, lm);
                    v4f gIx, gIy;
                    if (!s_in) {
                        const v4f a = px3(kp + 1, lm), b = px3(kp, lm);
                        gIx = !n_in ? vabs(a - b) + vabs(b - px3(km - 1, lm)) : vabs(a - b) * 2.0f;
                    } else {
                        gIx = !n_in ? vabs(px3(kp, lm) - px3(km - 1, lm)) * 2.0f : vzero;
                    }
                    if (!e_in) {
                        const v4f a = px3(km, lp + 1);
                        gIy = !w_in ? vabs(a - ctr) + vabs(ctr - px3(km, lm - 1)) : vabs(a - ctr) * 2.0f;
                    } else {
                        gIy = !w_in ? vabs(ctr - px3(km, lm - 1)) * 2.0f : vzero;
                    }
                    gIx = -gIx;
                    const v4f num = rx * gIx + ry * gIy;
                    // fabsf(dir) <= 0.01 compares in double; 0.01f is the largest float below 0.01, so the float comparison decides the same
                    const v4i small = vabs(num) <= 0.01f;
                    const v4f q = vabs(num / __builtin_elementwise_sqrt(r2 * (gIx * gIx + gIy * gIy)));
                    const v4f dir = small ? v4f{0.000001f, 0.000001f, 0.000001f, 0.000001f} : q;
                    const v4f w = o.w_ns * dir;
                    Ia += w * ctr;
                    sw += w;
                }
                v4f filled = {0.0f, 0.0f, 0.0f, 0.0f};
                for (int c = 0; c < 3; ++c) {
                    const uint8_t v = sat8(lrint((double)Ia[c] / sw[c]));
                    out[((size_t)(i - 1) * cols + (j - 1)) * 3 + c] = v;
                    filled[c] = (float)v;
                }
                fimg[(size_t)(i - 1) * cols + (j - 1)] = filled;
            } else {
                float Ia[C], s[C];
                for (int c = 0; c < C; ++c) { Ia[c] = 0.0f; s[c] = 1.0e-20f; }
                for (const Offset &o : offs) {
                    const int k = i + o.dk, l = j + o.dl;
                    if (!(k > 0 && l > 0 && k < er - 1 && l < ec - 1) || inside(k, l)) continue;
                    const int km = k - 1 + (k == 1), kp = k - 1 - (k == er - 2), lm = l - 1 + (l == 1), lp = l - 1 - (l == ec - 2);
                    const float ry = (float)(k - i), rx = (float)(l - j);
                    const float r2 = rx * rx + ry * ry;
                    const bool e_in = inside(k, l + 1), w_in = inside(k, l - 1), s_in = inside(k + 1, l), n_in = inside(k - 1, l);
                    for (int c = 0; c < C; ++c) {
                        float gIx, gIy;
                        if (!s_in) gIx = !n_in ? (float)(abs(px(kp + 1, lm, c) - px(kp, lm, c)) + abs(px(kp, lm, c) - px(km - 1, lm, c)))
                                               : (float)(abs(px(kp + 1, lm, c) - px(kp, lm, c))) * 2.0f;
                        else gIx = !n_in ? (float)(abs(px(kp, lm, c) - px(km - 1, lm, c))) * 2.0f : 0.0f;
                        if (!e_in) gIy = !w_in ? (float)(abs(px(km, lp + 1, c) - px(km, lm, c)) + abs(px(km, lm, c) - px(km, lm - 1, c)))
                                               : (float)(abs(px(km, lp + 1, c) - px(km, lm, c))) * 2.0f;
                        else gIy = !w_in ? (float)(abs(px(km, lm, c) - px(km, lm - 1, c))) * 2.0f : 0.0f;
                        gIx = -gIx;
                        float dir = rx * gIx + ry * gIy;
                        if (fabsf(dir) <= 0.01) dir = 0.000001f;
                        else dir = fabsf((rx * gIx + ry * gIy) / sqrtf(r2 * (gIx * gIx + gIy * gIy)));
                        const float w = o.w_ns * dir;
                        Ia[c] += w * (float)px(km, lm, c);
                        s[c] += w;
                    }
                }
                for (int c = 0; c < C; ++c) out[((size_t)(i - 1) * cols + (j - 1)) * C + c] = sat8(lrint((double)Ia[c] / s[c]));
            }
            f[(size_t)i * ec + j] = BAND;
            front.push(i, j, dist);
        }
    }
}

}   // namespace

extern "C" int mpf_inpaint_host(const uint8_t *img, const uint8_t *mask, int H, int W, int C, double radius, int method,
                                uint8_t *out)
{
    MPF_REQUIRE(img && mask && out, "mpf_inpaint_host: null pointer");
    MPF_REQUIRE(H >= 1 && W >= 1 && (int64_t)H * W < ((int64_t)1 << 30), "mpf_inpaint_host: bad shape %d x %d", H, W);
    MPF_REQUIRE(C == 1 || C == 3, "mpf_inpaint_host: 1 or 3 channels (got %d)", C);
    MPF_REQUIRE(method == MPF_INPAINT_NS || method == MPF_INPAINT_TELEA, "mpf_inpaint_host: method must be MPF_INPAINT_NS or MPF_INPAINT_TELEA");
    int range = (int)lrint(radius);
    range = std::max(1, std::min(range, 100));            // cvInpaint: cvRound, then clamped to [1, 100]
    try {
        if (C == 3) fill<3>(img, mask, H, W, range, method, out);
        else fill<1>(img, mask, H, W, range, method, out);
    } catch (const std::bad_alloc &) {
        mpf_set_error("mpf_inpaint_host: out of host memory");
        return MPF_ERR_UNSUPPORTED;
    }
    return 0;
}
