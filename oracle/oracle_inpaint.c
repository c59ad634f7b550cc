/* oracle_inpaint.c - CPU restatement of cv2.inpaint (INPAINT_NS and INPAINT_TELEA) and of cv2.dilate(3x3 ones).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED.  The arithmetic restated here is THIRD-PARTY: the reference calls
 *     cv2.inpaint(frame_mix, fill_mask, 3, cv2.INPAINT_NS)       utils/utils.py:284-286
 *     cv2.inpaint(im1_raw, 1 - H, 3, cv2.INPAINT_TELEA)          moving_obj.py:162
 *     cv2.dilate(M, np.ones((3,3)))                              moving_obj.py:144-145
 * from opencv-python (pinned 4.4.0.40 in the reference's README.md:38), which is neither vendored under /root/reference
 * nor installed in this image, and the reference holds no test vector for it.  What follows restates OpenCV's
 * published algorithm - modules/photo/src/inpaint.cpp: A. Telea, "An image inpainting technique based on the fast
 * marching method" (J. Graphics Tools 9(1), 2004) for INPAINT_TELEA; the same fast-marching front driving the
 * isophote-weighted average OpenCV calls "Navier-Stokes" (after Bertalmio, Bertozzi, Sapiro, CVPR 2001) for INPAINT_NS -
 * step by step in OpenCV's own order of operations and precisions (float / double / int as OpenCV mixes them), so that
 * it should agree with cv2 byte for byte; tests/test_inpaint.py compares against the real cv2 whenever it is importable
 * and skips otherwise.  Until that comparison has run somewhere, this file is pinned only by hand-checked properties
 * (tests/test_inpaint.py) and parity for rows A13 / N2 stays "unpinned" (DESIGN.md section 7).
 *
 * Structure of OpenCV's routine (all on a 1-pixel zero-padded copy of the mask, "e" = extended coordinates = image + 1):
 *   flags f: KNOWN 0, BAND 1, INSIDE 2, CHANGE 3;  t = 1e6 everywhere
 *   mask_e = INSIDE where mask != 0;  band = dilate(mask_e, 3x3 cross) - mask_e, border cleared;  t[band] = 0
 *   queue: doubly linked list kept sorted by T, ties in insertion order (FIFO); seeded with the band in raster order
 *   TELEA only: signed distance outside the hole - out = dilate(mask_e, (2r+1)^2 box) - mask_e - band, FMM from the band
 *               over `out`, then t = -t there
 *   main loop: pop (ii,jj) -> KNOWN; for its 4-neighbours (up, left, down, right) still INSIDE: t = eikonal update,
 *              colour = weighted combination of the non-INSIDE pixels within the radius, -> BAND, push(t)
 *
 * Precision note.  OpenCV's source calls sqrt() / fabs() unqualified on float expressions; the file is C++ and pulls in
 * <math.h> through the C API headers, where libstdc++ declares the float overloads in the global namespace, so those calls
 * are FLOAT operations unless the argument was cast to double in the source (it is in FastMarching_solve and in Telea's
 * distance weight).  This restatement follows that reading: sqrtf / fabsf on float arguments, double only where OpenCV
 * casts.  Built without FMA contraction (as the x86-64 baseline opencv-python wheels are).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* Which overloads OpenCV's unqualified sqrt() / fabs() calls on FLOAT arguments resolved to when the wheel was compiled decides
 * bytes at three sites (marked READING below), and cannot be settled without OpenCV at hand:
 *   reading 0 (default, the precision note above): the float overloads  -> sqrtf / fabsf, the surrounding sums stay float;
 *   reading 1: ::sqrt(double) / ::fabs(double) (what a translation unit sees that only has <cmath>, not <math.h>) -> the argument
 *              is promoted, and so is every operation the result feeds until the explicit (float) cast.
 * tests/golden/inpaint_reading_exhibit.npz holds inputs on which the two readings give different bytes; one run of the real
 * cv2.inpaint on them (tests/golden/make_cv2_golden.py, tests/test_inpaint.py) decides. */
static int g_reading = 0;
void orc_inpaint_set_reading(int r) { g_reading = r ? 1 : 0; }

#define KNOWN 0
#define BAND 1
#define INSIDE 2
#define CHANGE 3

/* ---- the priority queue: OpenCV's CvPriorityQueueFloat, a sorted doubly linked list over a node pool ------------------ */
typedef struct Node { float T; int i, j; struct Node *prev, *next; } Node;
typedef struct { Node *mem, *empty, *head, *tail; int num, in; } Queue;

static int q_init(Queue *q, const uint8_t *f, int rows, int cols)
{
    int num = 0;
    for (int i = 0; i < rows * cols; ++i) num += f[i] != 0;
    q->num = num; q->in = 0; q->mem = NULL;
    if (num <= 0) return 0;
    q->mem = (Node *)malloc((size_t)(num + 2) * sizeof(Node));
    if (!q->mem) return 0;
    Node *mem = q->mem;
    q->head = mem; mem[0].i = mem[0].j = -1; mem[0].prev = NULL; mem[0].next = mem + 1; mem[0].T = -FLT_MAX;
    q->empty = mem + 1;
    int i;
    for (i = 1; i <= num; ++i) { mem[i].prev = mem + i - 1; mem[i].next = mem + i + 1; mem[i].i = -1; mem[i].T = FLT_MAX; }
    q->tail = mem + i; q->tail->i = q->tail->j = -1; q->tail->prev = mem + i - 1; q->tail->next = NULL; q->tail->T = FLT_MAX;
    return 1;
}

static int q_push(Queue *q, int i, int j, float T)
{
    Node *tmp = q->empty, *add = q->empty;
    if (q->empty == q->tail) return 0;
    while (tmp->prev->T > T) tmp = tmp->prev;          /* behind every entry with T' <= T: FIFO among equals */
    if (tmp != q->empty) {
        add->prev->next = add->next; add->next->prev = add->prev;
        q->empty = add->next;
        add->prev = tmp->prev; add->next = tmp;
        add->prev->next = add; add->next->prev = add;
    } else {
        q->empty = q->empty->next;
    }
    add->i = i; add->j = j; add->T = T;
    q->in++;
    return 1;
}

static int q_pop(Queue *q, int *i, int *j)
{
    Node *tmp = q->head->next;
    if (q->empty == tmp) return 0;
    *i = tmp->i; *j = tmp->j;
    tmp->prev->next = tmp->next; tmp->next->prev = tmp->prev;
    tmp->prev = q->empty->prev; tmp->next = q->empty;
    tmp->prev->next = tmp; tmp->next->prev = tmp;
    q->empty = tmp;
    q->in--;
    return 1;
}

static int q_add(Queue *q, const uint8_t *f, int rows, int cols)
{
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j)
            if (f[i * cols + j] != 0 && !q_push(q, i, j, 0.0f)) return 0;
    return 1;
}

/* ---- helpers ------------------------------------------------------------------------------------------------------------ */
#define F(i, j) f[(i) * ecols + (j)]
#define TT(i, j) t[(i) * ecols + (j)]
static int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }
/* in range whenever rows, cols >= 2 (OpenCV's own index arithmetic); the clamp only keeps 1-pixel-wide images in bounds */
#define PIX(r, c, ch) out[((size_t)clampi((r), rows - 1) * cols + clampi((c), cols - 1)) * C + (ch)]

static float vlen2(float x, float y) { return x * x + y * y; }                  /* VectorLength: the SQUARED length */
static float vdot(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }

static uint8_t sat_u8_i(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static uint8_t sat_u8_f(float v) { return sat_u8_i((int)lrintf(v)); }          /* saturate_cast<uchar>(float): cvRound = to nearest even */
static uint8_t sat_u8_d(double v) { return sat_u8_i((int)lrint(v)); }

/* FastMarching_solve: eikonal update from the two neighbours (i1,j1), (i2,j2) */
static float fmm_solve(int i1, int j1, int i2, int j2, const uint8_t *f, const float *t, int ecols)
{
    double sol, a11 = TT(i1, j1), a22 = TT(i2, j2), m12 = a11 < a22 ? a11 : a22;
    if (F(i1, j1) != INSIDE) {
        if (F(i2, j2) != INSIDE) {
            if (fabs(a11 - a22) >= 1.0) sol = 1 + m12;
            else sol = (a11 + a22 + sqrt((double)(2 - (a11 - a22) * (a11 - a22)))) * 0.5;
        } else {
            sol = 1 + a11;
        }
    } else if (F(i2, j2) != INSIDE) {
        sol = 1 + a22;
    } else {
        sol = 1 + m12;
    }
    return (float)sol;
}

static float min4(float a, float b, float c, float d)
{
    a = a < b ? a : b; c = c < d ? c : d;
    return a < c ? a : c;
}

static float fmm_dist(int i, int j, const uint8_t *f, const float *t, int ecols)
{
    return min4(fmm_solve(i - 1, j, i, j - 1, f, t, ecols), fmm_solve(i + 1, j, i, j - 1, f, t, ecols),
                fmm_solve(i - 1, j, i, j + 1, f, t, ecols), fmm_solve(i + 1, j, i, j + 1, f, t, ecols));
}

/* icvCalcFMM(negate = true): distances of the ring outside the hole, stored negative */
static void calc_fmm_outside(uint8_t *f, float *t, Queue *q, int erows, int ecols)
{
    int ii = 0, jj = 0;
    while (q_pop(q, &ii, &jj)) {
        F(ii, jj) = CHANGE;
        for (int k = 0; k < 4; ++k) {
            int i = ii + (k == 0 ? -1 : (k == 2 ? 1 : 0)), j = jj + (k == 1 ? -1 : (k == 3 ? 1 : 0));
            if (i <= 0 || j <= 0 || i > erows - 1 || j > ecols - 1) continue;     /* the border ring is never INSIDE */
            if (F(i, j) == INSIDE) {
                float dist = fmm_dist(i, j, f, t, ecols);
                TT(i, j) = dist;
                F(i, j) = BAND;
                q_push(q, i, j, dist);
            }
        }
    }
    for (int i = 0; i < erows * ecols; ++i)
        if (f[i] == CHANGE) { f[i] = KNOWN; t[i] = -t[i]; }
}

static void dilate_to(const uint8_t *src, uint8_t *dst, int erows, int ecols, int ry, int rx, int cross)
{
    for (int i = 0; i < erows; ++i)
        for (int j = 0; j < ecols; ++j) {
            uint8_t m = 0;
            for (int di = -ry; di <= ry; ++di)
                for (int dj = -rx; dj <= rx; ++dj) {
                    if (cross && di != 0 && dj != 0) continue;
                    int y = i + di, x = j + dj;
                    if (y < 0 || x < 0 || y >= erows || x >= ecols) continue;      /* outside = the minimum: never wins */
                    if (src[y * ecols + x] > m) m = src[y * ecols + x];
                }
            dst[i * ecols + j] = m;
        }
}

static void clear_border(uint8_t *a, int erows, int ecols)
{
    for (int j = 0; j < ecols; ++j) a[j] = a[(erows - 1) * ecols + j] = 0;
    for (int i = 0; i < erows; ++i) a[i * ecols] = a[i * ecols + ecols - 1] = 0;
}

/* ---- the two fill rules -------------------------------------------------------------------------------------------------- */

/* icvTeleaInpaintFMM: one pixel (i,j) in extended coordinates */
static void telea_pixel(int i, int j, const uint8_t *f, const float *t, uint8_t *out, int rows, int cols, int C, int range)
{
    const int erows = rows + 2, ecols = cols + 2;
    float gTx, gTy;
    if (F(i, j + 1) != INSIDE) {
        if (F(i, j - 1) != INSIDE) gTx = (float)(TT(i, j + 1) - TT(i, j - 1)) * 0.5f;
        else gTx = (float)(TT(i, j + 1) - TT(i, j));
    } else {
        if (F(i, j - 1) != INSIDE) gTx = (float)(TT(i, j) - TT(i, j - 1));
        else gTx = 0;
    }
    if (F(i + 1, j) != INSIDE) {
        if (F(i - 1, j) != INSIDE) gTy = (float)(TT(i + 1, j) - TT(i - 1, j)) * 0.5f;
        else gTy = (float)(TT(i + 1, j) - TT(i, j));
    } else {
        if (F(i - 1, j) != INSIDE) gTy = (float)(TT(i, j) - TT(i - 1, j));
        else gTy = 0;
    }
    for (int color = 0; color < C; ++color) {
        float Jx = 0, Jy = 0, Ia = 0, s = 1.0e-20f;
        for (int k = i - range; k <= i + range; ++k) {
            const int km = k - 1 + (k == 1), kp = k - 1 - (k == erows - 2);
            for (int l = j - range; l <= j + range; ++l) {
                const int lm = l - 1 + (l == 1), lp = l - 1 - (l == ecols - 2);
                if (!(k > 0 && l > 0 && k < erows - 1 && l < ecols - 1)) continue;
                if (F(k, l) == INSIDE || (l - j) * (l - j) + (k - i) * (k - i) > range * range) continue;
                const float ry = (float)(i - k), rx = (float)(j - l);
                const float dst = (float)(1. / (vlen2(rx, ry) * sqrt((double)vlen2(rx, ry))));
                const float lev = g_reading ? (float)(1. / (1 + fabs((double)(TT(k, l) - TT(i, j)))))      /* READING: 1 + |dt| summed in double */
                                            : (float)(1. / (1 + fabsf(TT(k, l) - TT(i, j))));              /*          ... or as a FLOAT sum */
                float dir = vdot(rx, ry, gTx, gTy);
                if (fabsf(dir) <= 0.01) dir = 0.000001f;
                const float w = fabsf(dst * lev * dir);
                float gIx, gIy;
                if (F(k, l + 1) != INSIDE) {
                    if (F(k, l - 1) != INSIDE) gIx = (float)(PIX(km, lp + 1, color) - PIX(km, lm - 1, color)) * 2.0f;
                    else gIx = (float)(PIX(km, lp + 1, color) - PIX(km, lm, color));
                } else {
                    if (F(k, l - 1) != INSIDE) gIx = (float)(PIX(km, lp, color) - PIX(km, lm - 1, color));
                    else gIx = 0;
                }
                if (F(k + 1, l) != INSIDE) {
                    if (F(k - 1, l) != INSIDE) gIy = (float)(PIX(kp + 1, lm, color) - PIX(km - 1, lm, color)) * 2.0f;
                    else gIy = (float)(PIX(kp + 1, lm, color) - PIX(km, lm, color));
                } else {
                    if (F(k - 1, l) != INSIDE) gIy = (float)(PIX(kp, lm, color) - PIX(km - 1, lm, color));
                    else gIy = 0;
                }
                Ia += (float)w * (float)(PIX(km, lm, color));
                Jx -= (float)w * (float)(gIx * rx);
                Jy -= (float)w * (float)(gIy * ry);
                s += w;
            }
        }
        const float sat = g_reading ? (float)((Ia / s + (Jx + Jy) / (sqrt((double)(Jx * Jx + Jy * Jy)) + 1.0e-20f) + 0.5f))   /* READING */
                                    : (float)((Ia / s + (Jx + Jy) / (sqrtf(Jx * Jx + Jy * Jy) + 1.0e-20f) + 0.5f));
        PIX(i - 1, j - 1, color) = sat_u8_f(sat);
    }
}

/* icvNSInpaintFMM: one pixel */
static void ns_pixel(int i, int j, const uint8_t *f, uint8_t *out, int rows, int cols, int C, int range)
{
    const int erows = rows + 2, ecols = cols + 2;
    for (int color = 0; color < C; ++color) {
        float Ia = 0, s = 1.0e-20f;
        for (int k = i - range; k <= i + range; ++k) {
            const int km = k - 1 + (k == 1), kp = k - 1 - (k == erows - 2);
            for (int l = j - range; l <= j + range; ++l) {
                const int lm = l - 1 + (l == 1), lp = l - 1 - (l == ecols - 2);
                if (!(k > 0 && l > 0 && k < erows - 1 && l < ecols - 1)) continue;
                if (F(k, l) == INSIDE || (l - j) * (l - j) + (k - i) * (k - i) > range * range) continue;
                const float ry = (float)(k - i), rx = (float)(l - j);
                const float dst = 1 / (vlen2(rx, ry) * vlen2(rx, ry) + 1);
                float gIx, gIy;
                if (F(k + 1, l) != INSIDE) {
                    if (F(k - 1, l) != INSIDE)
                        gIx = (float)(abs(PIX(kp + 1, lm, color) - PIX(kp, lm, color)) + abs(PIX(kp, lm, color) - PIX(km - 1, lm, color)));
                    else gIx = (float)(abs(PIX(kp + 1, lm, color) - PIX(kp, lm, color))) * 2.0f;
                } else {
                    if (F(k - 1, l) != INSIDE) gIx = (float)(abs(PIX(kp, lm, color) - PIX(km - 1, lm, color))) * 2.0f;
                    else gIx = 0;
                }
                if (F(k, l + 1) != INSIDE) {
                    if (F(k, l - 1) != INSIDE)
                        gIy = (float)(abs(PIX(km, lp + 1, color) - PIX(km, lm, color)) + abs(PIX(km, lm, color) - PIX(km, lm - 1, color)));
                    else gIy = (float)(abs(PIX(km, lp + 1, color) - PIX(km, lm, color))) * 2.0f;
                } else {
                    if (F(k, l - 1) != INSIDE) gIy = (float)(abs(PIX(km, lm, color) - PIX(km, lm - 1, color))) * 2.0f;
                    else gIy = 0;
                }
                gIx = -gIx;
                float dir = vdot(rx, ry, gIx, gIy);
                if (fabsf(dir) <= 0.01) dir = 0.000001f;
                else if (g_reading) dir = (float)fabs(vdot(rx, ry, gIx, gIy) / sqrt((double)(vlen2(rx, ry) * vlen2(gIx, gIy))));   /* READING */
                else dir = fabsf(vdot(rx, ry, gIx, gIy) / sqrtf(vlen2(rx, ry) * vlen2(gIx, gIy)));
                const float w = dst * dir;
                Ia += (float)w * (float)(PIX(km, lm, color));
                s += w;
            }
        }
        PIX(i - 1, j - 1, color) = sat_u8_d((double)Ia / s);
    }
}

/* cvInpaint.  img u8 [rows, cols, C] (C = 1 or 3), mask u8 [rows, cols] (non-zero = to fill), out u8 [rows, cols, C].
 * method: 0 = INPAINT_NS, 1 = INPAINT_TELEA (cv2's values).  radius is rounded and clamped to [1, 100] as OpenCV does.
 * returns 0, or -1 on allocation failure. */
int orc_inpaint(const uint8_t *img, const uint8_t *mask_in, int rows, int cols, int C, double radius, int method, uint8_t *out)
{
    int range = (int)lrint(radius);
    range = range < 1 ? 1 : (range > 100 ? 100 : range);
    const int erows = rows + 2, ecols = cols + 2;
    const size_t en = (size_t)erows * ecols;
    memcpy(out, img, (size_t)rows * cols * C);
    uint8_t *mask = (uint8_t *)calloc(en, 1), *band = (uint8_t *)calloc(en, 1), *ring = NULL;
    float *t = (float *)malloc(en * sizeof(float));
    if (!mask || !band || !t) { free(mask); free(band); free(t); return -1; }
    for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j)
            if (mask_in[(size_t)i * cols + j] != 0) mask[(size_t)(i + 1) * ecols + j + 1] = INSIDE;
    clear_border(mask, erows, ecols);
    for (size_t n = 0; n < en; ++n) t[n] = 1.0e6f;
    dilate_to(mask, band, erows, ecols, 1, 1, 1);
    Queue heap, outq;
    outq.mem = NULL;
    int rc = 0;
    if (!q_init(&heap, band, erows, ecols)) goto done;              /* empty mask: the copy is the result */
    for (size_t n = 0; n < en; ++n) band[n] = (uint8_t)(band[n] > mask[n] ? band[n] - mask[n] : 0);
    clear_border(band, erows, ecols);
    if (!q_add(&heap, band, erows, ecols)) goto done;
    for (size_t n = 0; n < en; ++n) if (band[n]) t[n] = 0.0f;
    uint8_t *f = mask;                                              /* OpenCV hands `mask` (KNOWN / INSIDE) to the fill routines */
    if (method == 1) {
        ring = (uint8_t *)calloc(en, 1);
        if (!ring) { rc = -1; goto done; }
        dilate_to(mask, ring, erows, ecols, range, range, 0);
        for (size_t n = 0; n < en; ++n) ring[n] = (uint8_t)(ring[n] > mask[n] ? ring[n] - mask[n] : 0);
        if (!q_init(&outq, ring, erows, ecols)) goto done;
        if (!q_add(&outq, band, erows, ecols)) goto done;
        for (size_t n = 0; n < en; ++n) ring[n] = (uint8_t)(ring[n] > band[n] ? ring[n] - band[n] : 0);
        clear_border(ring, erows, ecols);
        calc_fmm_outside(ring, t, &outq, erows, ecols);
    }
    {
        int ii = 0, jj = 0;
        while (q_pop(&heap, &ii, &jj)) {
            F(ii, jj) = KNOWN;
            for (int k = 0; k < 4; ++k) {
                int i = ii + (k == 0 ? -1 : (k == 2 ? 1 : 0)), j = jj + (k == 1 ? -1 : (k == 3 ? 1 : 0));
                if (i <= 0 || j <= 0 || i > erows - 1 || j > ecols - 1) continue;
                if (F(i, j) != INSIDE) continue;
                const float dist = fmm_dist(i, j, f, t, ecols);
                TT(i, j) = dist;
                if (method == 1) telea_pixel(i, j, f, t, out, rows, cols, C, range);
                else ns_pixel(i, j, f, out, rows, cols, C, range);
                F(i, j) = BAND;
                q_push(&heap, i, j, dist);
            }
        }
    }
done:
    free(heap.mem); free(outq.mem); free(mask); free(band); free(ring); free(t);
    return rc;
}

/* cv2.dilate(img, np.ones((3,3))) on one u8 channel: 3x3 maximum, pixels outside the image never win (moving_obj.py:144-145) */
void orc_dilate3x3(const uint8_t *img, int rows, int cols, uint8_t *out)
{
    dilate_to(img, out, rows, cols, 1, 1, 0);
}
