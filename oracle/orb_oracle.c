/*
 * orb_oracle.c -- CPU restatement (ORACLE) of ORBextractor::operator() and the ORBmatcher
 * Hamming routines of hwb0314/VI-ORB-SLAM-ICRA2018.  See orb_oracle.h for the status header
 * ("parity unpinned" at the OpenCV boundary) and the rule that only tests / smoke / the
 * cpu_baseline leg of bench.py may use this file.
 *
 * Build: gcc -O3 -march=native -ffp-contract=off (CMakeLists.txt:10-11 of the reference uses
 * -O3 -march=native; contraction is disabled so that x*b + y*a is evaluated as two roundings,
 * the canonical choice of SURVEY.md Appendix A7).
 *
 * All "ref:" citations are relative to /root/reference.  "cv2.4:" marks behaviour of the
 * external OpenCV 2.4.x library restated from its published algorithm (file named for
 * orientation only; the source is not available in this environment).
 */
#include "orb_oracle.h"
#include "orb_pattern_data.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PATCH_SIZE 31        /* ref: src/ORBextractor.cc:74 */
#define HALF_PATCH_SIZE 15   /* ref: src/ORBextractor.cc:75 */
#define EDGE_THRESHOLD 19    /* ref: src/ORBextractor.cc:76 */

/* ------------------------------------------------------------------------------------------
 * OpenCV scalar helpers
 * ---------------------------------------------------------------------------------------- */

/* cv2.4: cvRound(double) = _mm_cvtsd_si32 => round half to even in the default MXCSR mode. */
int orbo_cvround(double v) { return (int)lrint(v); }

static int cv_floor(double v)
{
    int i = (int)v;
    return i - (i > v);
}

static int cv_ceil(double v)
{
    int i = (int)v;
    return i + (i < v);
}

static uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* cv2.4: saturate_cast<short>(float) = saturate(cvRound(v)). */
static short sat_s16_from_float(float v)
{
    int i = orbo_cvround((double)v);
    return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i));
}

/* cv2.4 (core/mathfuncs.cpp): fastAtan2, degrees in [0,360), float32 arithmetic, 7th-order odd
 * polynomial.  Called at ref: src/ORBextractor.cc:105. */
float orbo_fast_atan2(float y, float x)
{
    static const float rad2deg = (float)(180.0 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * rad2deg;
    const float p3 = -0.3258083974640975f * rad2deg;
    const float p5 = 0.1555786518463281f * rad2deg;
    const float p7 = -0.04432655554792128f * rad2deg;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ------------------------------------------------------------------------------------------
 * E0: constructor arithmetic.  ref: src/ORBextractor.cc:412-472
 * ---------------------------------------------------------------------------------------- */
int orbo_params_init(orbo_params *p, int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                     int minThFAST)
{
    if (!p || nlevels < 1 || nlevels > ORBO_MAX_LEVELS || nfeatures < 1) return -1;
    memset(p, 0, sizeof(*p));
    p->nfeatures = nfeatures;
    p->nlevels = nlevels;
    p->iniThFAST = iniThFAST;
    p->minThFAST = minThFAST;
    p->scaleFactor = (double)scaleFactor; /* member is double: include/ORBextractor.h:116 */

    /* :417-425  float * double -> double -> float */
    p->mvScaleFactor[0] = 1.0f;
    p->mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        p->mvScaleFactor[i] = (float)((double)p->mvScaleFactor[i - 1] * p->scaleFactor);
        p->mvLevelSigma2[i] = p->mvScaleFactor[i] * p->mvScaleFactor[i];
    }
    /* :427-433 */
    for (int i = 0; i < nlevels; i++) {
        p->mvInvScaleFactor[i] = 1.0f / p->mvScaleFactor[i];
        p->mvInvLevelSigma2[i] = 1.0f / p->mvLevelSigma2[i];
    }
    /* :437-448 */
    float factor = (float)(1.0 / p->scaleFactor); /* 1.0f / double */
    float nDesired =
        (float)nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        p->mnFeaturesPerLevel[level] = orbo_cvround((double)nDesired);
        sum += p->mnFeaturesPerLevel[level];
        nDesired *= factor;
    }
    p->mnFeaturesPerLevel[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;

    /* :456-471  end of each row of the radius-15 disc */
    int vmax = cv_floor((double)(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1));
    int vmin = cv_ceil((double)(HALF_PATCH_SIZE * sqrtf(2.f) / 2));
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (int v = 0; v <= vmax; ++v) p->umax[v] = orbo_cvround(sqrt(hp2 - v * v));
    for (int v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (p->umax[v0] == p->umax[v0 + 1]) ++v0;
        p->umax[v] = v0;
        ++v0;
    }
    return 0;
}

/* ref: src/ORBextractor.cc:1132-1133 -- size from the ORIGINAL image, float multiply. */
void orbo_level_size(const orbo_params *p, int cols, int rows, int level, int *w, int *h)
{
    float scale = p->mvInvScaleFactor[level];
    *w = orbo_cvround((double)((float)cols * scale));
    *h = orbo_cvround((double)((float)rows * scale));
}

/* ------------------------------------------------------------------------------------------
 * E2: cv::resize(..., INTER_LINEAR) for 8UC1.  cv2.4: imgproc/imgwarp.cpp
 *   scale = 1/(dsize/ssize) (double); f = (float)((d+0.5)*scale-0.5); s = floor(f); f -= s;
 *   x: s<0 -> (0, f=0); s>=sw-1 -> (sw-1, f=0); weights = saturate_cast<short>(w*2048);
 *   horizontal: S[s]*a0 + S[s+1]*a1 (int); vertical rows clamped to [0,sh-1];
 *   out = (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
 * ---------------------------------------------------------------------------------------- */
void orbo_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw,
                           int dh, int dstride)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * (size_t)dw);
    int *row0 = (int *)malloc(sizeof(int) * (size_t)dw);
    int *row1 = (int *)malloc(sizeof(int) * (size_t)dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor((double)fx);
        fx -= sx;
        if (sx < 0) {
            fx = 0;
            sx = 0;
        }
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) {
                fx = 0;
                sx = sw - 1;
            }
        }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_s16_from_float((1.f - fx) * 2048);
        ialpha[2 * dx + 1] = sat_s16_from_float(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor((double)fy);
        fy -= sy;
        short b0 = sat_s16_from_float((1.f - fy) * 2048);
        short b1 = sat_s16_from_float(fy * 2048);
        int sy0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);
        int sy1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
        const uint8_t *S0 = src + (size_t)sy0 * sstride;
        const uint8_t *S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            if (dx < xmax) {
                row0[dx] = S0[sx] * ialpha[2 * dx] + S0[sx + 1] * ialpha[2 * dx + 1];
                row1[dx] = S1[sx] * ialpha[2 * dx] + S1[sx + 1] * ialpha[2 * dx + 1];
            } else {
                row0[dx] = S0[sx] * 2048;
                row1[dx] = S1[sx] * 2048;
            }
        }
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs);
    free(ialpha);
    free(row0);
    free(row1);
}

/* ------------------------------------------------------------------------------------------
 * E6: cv::GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101) for 8UC1.
 * ref call site: src/ORBextractor.cc:1103-1104.  cv2.4: imgproc/smooth.cpp + filter.cpp:
 *   float kernel g[i] = exp(-(i-3)^2/8)/sum; fixed point k = cvRound(g*256) = {18,34,49,55,...}
 *   (sum 257, not renormalised); row pass int32; column pass brings the sum back by 2^16.
 *   On x86-64 (SSE2 always present) the column pass of the 32s->8u symmetric filter runs in
 *   float for x < width - width%4:  s = r0*f0; s += (r[+k]+r[-k])*f[k], f = k/65536, then
 *   cvtps2dq (round half to EVEN) and saturating packs; the last width%4 pixels use the integer
 *   cast (sum + 32768) >> 16 (round half UP).  Both are restated literally.
 * ---------------------------------------------------------------------------------------- */
static int reflect101(int p, int len)
{
    while (p < 0 || p >= len) {
        if (p < 0)
            p = -p;
        else
            p = 2 * len - 2 - p;
    }
    return p;
}

static void gaussian_kernel7_fixed(int k[7])
{
    /* cv2.4 getGaussianKernel(7, 2, CV_32F) then convertTo(CV_32S, 256). */
    double t[7], sum = 0;
    float cf[7];
    const double scale2X = -0.5 / (2.0 * 2.0);
    for (int i = 0; i < 7; i++) {
        double x = i - 3.0;
        t[i] = exp(scale2X * x * x);
        cf[i] = (float)t[i];
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) {
        cf[i] = (float)(cf[i] * sum);
        k[i] = orbo_cvround((double)(cf[i] * 256.f));
    }
}

void orbo_gaussian_blur7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst, int dstride)
{
    int k[7];
    gaussian_kernel7_fixed(k);
    float f[4];
    for (int i = 0; i < 4; i++) f[i] = (float)((double)k[3 + i] * (1. / 65536));
    int *rows = (int *)malloc(sizeof(int) * (size_t)w * (size_t)h);
    int *pad = (int *)malloc(sizeof(int) * (size_t)(w + 6));
    for (int y = 0; y < h; y++) {
        const uint8_t *S = src + (size_t)y * sstride;
        int *R = rows + (size_t)y * w;
        for (int x = -3; x < w + 3; x++) pad[x + 3] = S[reflect101(x, w)];
        for (int x = 0; x < w; x++)
            R[x] = k[0] * (pad[x] + pad[x + 6]) + k[1] * (pad[x + 1] + pad[x + 5]) +
                   k[2] * (pad[x + 2] + pad[x + 4]) + k[3] * pad[x + 3];
    }
    free(pad);
    const int wvec = w - (w % 4);
    for (int y = 0; y < h; y++) {
        const int *R[7];
        for (int j = -3; j <= 3; j++) R[j + 3] = rows + (size_t)reflect101(y + j, h) * w;
        uint8_t *D = dst + (size_t)y * dstride;
        for (int x = 0; x < wvec; x++) {
            float s = (float)R[3][x] * f[0];
            s = s + (float)(R[4][x] + R[2][x]) * f[1];
            s = s + (float)(R[5][x] + R[1][x]) * f[2];
            s = s + (float)(R[6][x] + R[0][x]) * f[3];
            int v = (int)lrintf(s);
            v = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
            D[x] = sat_u8(v);
        }
        for (int x = wvec; x < w; x++) {
            int s = k[3] * R[3][x];
            for (int j = 1; j <= 3; j++) s += k[3 + j] * (R[3 + j][x] + R[3 - j][x]);
            D[x] = sat_u8((s + 32768) >> 16);
        }
    }
    free(rows);
}

/* ------------------------------------------------------------------------------------------
 * E3f: cv::FAST(img, keypoints, threshold, nonmaxSuppression=true), TYPE_9_16.
 * cv2.4: features2d/fast.cpp + fast_score.cpp.
 *   ring = 16-pixel Bresenham circle of radius 3; p is a corner iff >= 9 contiguous ring pixels
 *   are all < p - t or all > p + t.  Score = max(t, max over the 16 9-arcs of the arc-min of
 *   (p - q), same of (q - p)) - 1, stored as uchar.  NMS: strictly greater than the 8
 *   neighbours' scores, where non-corners and positions outside rows/cols [3, n-3) count 0.
 *   Output KeyPoint(x, y, 7, -1, score) in raster order.
 * ---------------------------------------------------------------------------------------- */
static const int RING_DX[16] = {0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1};
static const int RING_DY[16] = {3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3};

static int fast_is_corner(const uint8_t *p, int stride, int th)
{
    int v = p[0];
    int q[25];
    for (int k = 0; k < 25; k++) q[k] = p[RING_DY[k & 15] * stride + RING_DX[k & 15]];
    int lo = v - th, hi = v + th, run = 0;
    for (int k = 0; k < 25; k++) {
        if (q[k] < lo) {
            if (++run > 8) return 1;
        } else
            run = 0;
    }
    run = 0;
    for (int k = 0; k < 25; k++) {
        if (q[k] > hi) {
            if (++run > 8) return 1;
        } else
            run = 0;
    }
    return 0;
}

int orbo_fast_corner_score(const uint8_t *p, int stride, int th)
{
    int v = p[0];
    int d[25];
    for (int k = 0; k < 25; k++) d[k] = v - p[RING_DY[k & 15] * stride + RING_DX[k & 15]];
    int a0 = th;
    for (int s = 0; s < 16; s++) { /* arc s..s+8: all darker than centre by more than a0? */
        int m = d[s];
        for (int j = 1; j < 9; j++) m = d[s + j] < m ? d[s + j] : m;
        if (m > a0) a0 = m;
    }
    int b0 = -a0;
    for (int s = 0; s < 16; s++) {
        int m = d[s];
        for (int j = 1; j < 9; j++) m = d[s + j] > m ? d[s + j] : m;
        if (m < b0) b0 = m;
    }
    return -b0 - 1;
}

int orbo_fast9_16(const uint8_t *img, int stride, int w, int h, int th, orbo_cand *out, int cap)
{
    if (w < 7 || h < 7) return 0;
    th = th < 0 ? 0 : (th > 255 ? 255 : th);
    /* class table: bit0 = darker than centre by more than th, bit1 = brighter (cv2.4 keeps the
     * same 512-entry table so that most pixels are rejected after two lookups). */
    uint8_t tab[512];
    for (int i = -255; i <= 255; i++) tab[i + 255] = (uint8_t)(i < -th ? 1 : (i > th ? 2 : 0));
    int off[16];
    for (int k = 0; k < 16; k++) off[k] = RING_DY[k] * stride + RING_DX[k];
    uint8_t *score = (uint8_t *)calloc((size_t)w * (size_t)h, 1);
    for (int y = 3; y < h - 3; y++) {
        const uint8_t *row = img + (size_t)y * stride;
        uint8_t *srow = score + (size_t)y * w;
        for (int x = 3; x < w - 3; x++) {
            const uint8_t *p = row + x;
            const uint8_t *t = tab + 255 - p[0];
            /* a 9-arc contains at least one pixel of every opposite pair */
            int d = t[p[off[0]]] | t[p[off[8]]];
            if (!d) continue;
            d &= t[p[off[4]]] | t[p[off[12]]];
            if (!d) continue;
            d &= t[p[off[2]]] | t[p[off[10]]];
            d &= t[p[off[6]]] | t[p[off[14]]];
            if (!d) continue;
            d &= t[p[off[1]]] | t[p[off[9]]];
            d &= t[p[off[3]]] | t[p[off[11]]];
            d &= t[p[off[5]]] | t[p[off[13]]];
            d &= t[p[off[7]]] | t[p[off[15]]];
            if (!d) continue;
            if (fast_is_corner(p, stride, th)) {
                int sc = orbo_fast_corner_score(p, stride, th);
                srow[x] = (uint8_t)sc; /* stored as uchar in cv2.4 */
            }
        }
    }
    /* NMS.  A corner always has score >= th; with th >= 1 "score != 0" identifies corners, and
     * for th == 0 a zero-score corner can never be a strict maximum anyway. */
    int n = 0;
    for (int y = 3; y < h - 3; y++) {
        const uint8_t *s0 = score + (size_t)y * w;
        for (int x = 3; x < w - 3; x++) {
            const uint8_t *s = s0 + x;
            int sc = s[0];
            if (sc == 0) continue;
            if (sc > s[-1] && sc > s[1] && sc > s[-w - 1] && sc > s[-w] && sc > s[-w + 1] &&
                sc > s[w - 1] && sc > s[w] && sc > s[w + 1]) {
                if (n >= cap) {
                    free(score);
                    return -3;
                }
                out[n].x = x;
                out[n].y = y;
                out[n].score = sc;
                n++;
            }
        }
    }
    free(score);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * E3: per-level cell loop.  ref: src/ORBextractor.cc:767-831
 * ---------------------------------------------------------------------------------------- */
int orbo_level_candidates(const uint8_t *img, int w, int h, int stride, int iniTh, int minTh,
                          orbo_cand *out, int cap)
{
    const float W = 30;
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = w - EDGE_THRESHOLD + 3, maxBorderY = h - EDGE_THRESHOLD + 3;
    const float width = (float)(maxBorderX - minBorderX);
    const float height = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width / W), nRows = (int)(height / W);
    if (nCols < 1 || nRows < 1) return -2; /* reference divides by zero here */
    const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
    int n = 0;
    const int cellcap = (wCell + 6) * (hCell + 6);
    orbo_cand *cell = (orbo_cand *)malloc(sizeof(orbo_cand) * (size_t)cellcap);
    for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
            const float iniX = (float)(minBorderX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBorderX - 6) continue;
            if (maxX > maxBorderX) maxX = (float)maxBorderX;
            const uint8_t *sub = img + (size_t)(int)iniY * stride + (int)iniX;
            int sw = (int)maxX - (int)iniX, sh = (int)maxY - (int)iniY;
            int nc = orbo_fast9_16(sub, stride, sw, sh, iniTh, cell, cellcap);
            if (nc == 0) nc = orbo_fast9_16(sub, stride, sw, sh, minTh, cell, cellcap);
            if (nc < 0 || n + nc > cap) {
                free(cell);
                return -3;
            }
            for (int c = 0; c < nc; c++) {
                out[n].x = cell[c].x + j * wCell;
                out[n].y = cell[c].y + i * hCell;
                out[n].score = cell[c].score;
                n++;
            }
        }
    }
    free(cell);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * E4: quadtree distribution.  ref: src/ORBextractor.cc:483-539 (DivideNode), :541-765.
 * Index-based doubly linked list instead of std::list; behaviour per Appendix A4/C of the
 * survey: children are pushed to the list front in order n1..n4, parents erased in place;
 * final phase expands the largest nodes first; ties on size are broken by creation sequence
 * (canonical replacement for the reference's pointer compare).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int ulx, uly, brx, bry; /* UL and BR corners; UR=(brx,uly), BL=(ulx,bry) */
    int first, count;       /* points live in pts[first .. first+count) of the node's arena */
    int *pts;
    int nomore;
    int prev, next;
    int seq;
} qnode;

typedef struct {
    qnode *nodes;
    int nnodes, capnodes;
    int head, tail, size;
    int seq;
} qlist;

static int ql_new(qlist *L)
{
    if (L->nnodes == L->capnodes) {
        L->capnodes = L->capnodes ? L->capnodes * 2 : 256;
        L->nodes = (qnode *)realloc(L->nodes, sizeof(qnode) * (size_t)L->capnodes);
    }
    int id = L->nnodes++;
    memset(&L->nodes[id], 0, sizeof(qnode));
    L->nodes[id].prev = L->nodes[id].next = -1;
    L->nodes[id].seq = L->seq++;
    return id;
}

static void ql_push_front(qlist *L, int id)
{
    L->nodes[id].prev = -1;
    L->nodes[id].next = L->head;
    if (L->head >= 0) L->nodes[L->head].prev = id;
    L->head = id;
    if (L->tail < 0) L->tail = id;
    L->size++;
}

static void ql_push_back(qlist *L, int id)
{
    L->nodes[id].next = -1;
    L->nodes[id].prev = L->tail;
    if (L->tail >= 0) L->nodes[L->tail].next = id;
    L->tail = id;
    if (L->head < 0) L->head = id;
    L->size++;
}

static int ql_erase(qlist *L, int id) /* returns next */
{
    int p = L->nodes[id].prev, n = L->nodes[id].next;
    if (p >= 0)
        L->nodes[p].next = n;
    else
        L->head = n;
    if (n >= 0)
        L->nodes[n].prev = p;
    else
        L->tail = p;
    L->size--;
    free(L->nodes[id].pts);
    L->nodes[id].pts = NULL;
    return n;
}

/* ref: :483-539.  Creates up to four children; child[k] = -1 when empty. */
static void q_divide(qlist *L, int id, const orbo_cand *c, int child[4])
{
    qnode P = L->nodes[id];
    const int halfX = (int)ceilf((float)(P.brx - P.ulx) / 2);
    const int halfY = (int)ceilf((float)(P.bry - P.uly) / 2);
    const int midx = P.ulx + halfX, midy = P.uly + halfY;
    int box[4][4] = {{P.ulx, P.uly, midx, midy},
                     {midx, P.uly, P.brx, midy},
                     {P.ulx, midy, midx, P.bry},
                     {midx, midy, P.brx, P.bry}};
    int *buf[4], cnt[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; k++) buf[k] = (int *)malloc(sizeof(int) * (size_t)(P.count ? P.count : 1));
    for (int i = 0; i < P.count; i++) {
        int pi = P.pts[i];
        float px = (float)c[pi].x, py = (float)c[pi].y;
        int k;
        if (px < (float)midx)
            k = (py < (float)midy) ? 0 : 2;
        else
            k = (py < (float)midy) ? 1 : 3;
        buf[k][cnt[k]++] = pi;
    }
    for (int k = 0; k < 4; k++) {
        if (cnt[k] == 0) {
            free(buf[k]);
            child[k] = -1;
            continue;
        }
        int nid = ql_new(L);
        qnode *n = &L->nodes[nid];
        n->ulx = box[k][0];
        n->uly = box[k][1];
        n->brx = box[k][2];
        n->bry = box[k][3];
        n->pts = buf[k];
        n->count = cnt[k];
        n->nomore = (cnt[k] == 1);
        child[k] = nid;
    }
}

typedef struct {
    int size, seq, id;
} qsz;

static int qsz_cmp(const void *a, const void *b)
{
    const qsz *x = (const qsz *)a, *y = (const qsz *)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0); /* canonical pointer order */
}

int orbo_distribute_octtree(const orbo_cand *c, int ncand, int width, int height, int N,
                            int *out_idx, int cap)
{
    if (height <= 0 || width <= 0) return -2;
    /* :545-547 */
    const int nIni = (int)roundf((float)width / (float)height);
    if (nIni < 1) return -2; /* reference divides by zero */
    const float hX = (float)width / nIni;
    qlist L;
    memset(&L, 0, sizeof(L));
    L.head = L.tail = -1;
    int *root = (int *)malloc(sizeof(int) * (size_t)nIni);
    for (int i = 0; i < nIni; i++) { /* :554-565 */
        int id = ql_new(&L);
        qnode *n = &L.nodes[id];
        n->ulx = (int)(hX * (float)i);
        n->uly = 0;
        n->brx = (int)(hX * (float)(i + 1));
        n->bry = height;
        n->pts = (int *)malloc(sizeof(int) * (size_t)(ncand ? ncand : 1));
        ql_push_back(&L, id);
        root[i] = id;
    }
    for (int i = 0; i < ncand; i++) { /* :568-572 */
        int r = (int)((float)c[i].x / hX);
        if (r < 0) r = 0;
        if (r >= nIni) r = nIni - 1; /* out-of-range is UB in the reference; never hit for x<width */
        qnode *n = &L.nodes[root[r]];
        n->pts[n->count++] = i;
    }
    free(root);
    for (int it = L.head; it >= 0;) { /* :574-587 */
        if (L.nodes[it].count == 1) {
            L.nodes[it].nomore = 1;
            it = L.nodes[it].next;
        } else if (L.nodes[it].count == 0)
            it = ql_erase(&L, it);
        else
            it = L.nodes[it].next;
    }

    int finish = 0;
    qsz *vs = NULL, *vprev = NULL;
    int nvs = 0, capvs = 0, nprev = 0, capprev = 0;
#define VS_PUSH(sz_, id_)                                               \
    do {                                                                \
        if (nvs == capvs) {                                             \
            capvs = capvs ? capvs * 2 : 256;                            \
            vs = (qsz *)realloc(vs, sizeof(qsz) * (size_t)capvs);       \
        }                                                               \
        vs[nvs].size = (sz_);                                           \
        vs[nvs].id = (id_);                                             \
        vs[nvs].seq = L.nodes[(id_)].seq;                               \
        nvs++;                                                          \
    } while (0)

    while (!finish) { /* :596-741 */
        int prevSize = L.size;
        int nToExpand = 0;
        nvs = 0;
        for (int it = L.head; it >= 0;) {
            if (L.nodes[it].nomore) {
                it = L.nodes[it].next;
                continue;
            }
            int ch[4];
            q_divide(&L, it, c, ch);
            for (int k = 0; k < 4; k++) {
                if (ch[k] < 0) continue;
                ql_push_front(&L, ch[k]);
                if (L.nodes[ch[k]].count > 1) {
                    nToExpand++;
                    VS_PUSH(L.nodes[ch[k]].count, ch[k]);
                }
            }
            it = ql_erase(&L, it);
        }
        if (L.size >= N || L.size == prevSize) {
            finish = 1;
        } else if (L.size + nToExpand * 3 > N) {
            while (!finish) { /* :675-739 */
                prevSize = L.size;
                if (nvs > capprev) {
                    capprev = nvs;
                    vprev = (qsz *)realloc(vprev, sizeof(qsz) * (size_t)capprev);
                }
                memcpy(vprev, vs, sizeof(qsz) * (size_t)nvs);
                nprev = nvs;
                nvs = 0;
                qsort(vprev, (size_t)nprev, sizeof(qsz), qsz_cmp);
                for (int j = nprev - 1; j >= 0; j--) {
                    int ch[4];
                    q_divide(&L, vprev[j].id, c, ch);
                    for (int k = 0; k < 4; k++) {
                        if (ch[k] < 0) continue;
                        ql_push_front(&L, ch[k]);
                        if (L.nodes[ch[k]].count > 1) VS_PUSH(L.nodes[ch[k]].count, ch[k]);
                    }
                    ql_erase(&L, vprev[j].id);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) finish = 1;
            }
        }
    }
#undef VS_PUSH
    /* :743-764  best response per node, first wins ties; output in list order */
    int n = 0, rc = 0;
    for (int it = L.head; it >= 0; it = L.nodes[it].next) {
        qnode *q = &L.nodes[it];
        int best = q->pts[0];
        float maxResponse = (float)c[best].score;
        for (int k = 1; k < q->count; k++) {
            if ((float)c[q->pts[k]].score > maxResponse) {
                best = q->pts[k];
                maxResponse = (float)c[best].score;
            }
        }
        if (n >= cap) {
            rc = -3;
            break;
        }
        out_idx[n++] = best;
    }
    for (int i = 0; i < L.nnodes; i++) free(L.nodes[i].pts);
    free(L.nodes);
    free(vs);
    free(vprev);
    return rc < 0 ? rc : n;
}

/* ------------------------------------------------------------------------------------------
 * E5: intensity-centroid orientation.  ref: src/ORBextractor.cc:79-106
 * ---------------------------------------------------------------------------------------- */
float orbo_ic_angle(const uint8_t *img, int stride, int x, int y, const int *umax)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t *center = img + (size_t)y * stride + x;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orbo_fast_atan2((float)m_01, (float)m_10);
}

/* ------------------------------------------------------------------------------------------
 * E7: steered BRIEF.  ref: src/ORBextractor.cc:109-149.  cos/sin are libm's float versions,
 * as in the reference (`(float)cos(angle)` with a float argument under `using namespace std`).
 * ---------------------------------------------------------------------------------------- */
void orbo_brief(const uint8_t *blurred, int stride, int x, int y, float angle_deg, uint8_t desc[32])
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = angle_deg * factorPI;
    float a = cosf(angle), b = sinf(angle);
    const uint8_t *center = blurred + (size_t)y * stride + x;
    const signed char *pat = ORB_PATTERN_I8;
    for (int i = 0; i < 32; ++i) {
        int val = 0;
        for (int k = 0; k < 8; k++, pat += 4) {
            float x0 = (float)pat[0], y0 = (float)pat[1], x1 = (float)pat[2], y1 = (float)pat[3];
            int r0 = orbo_cvround((double)(x0 * b + y0 * a)), c0 = orbo_cvround((double)(x0 * a - y0 * b));
            int r1 = orbo_cvround((double)(x1 * b + y1 * a)), c1 = orbo_cvround((double)(x1 * a - y1 * b));
            int t0 = center[r0 * stride + c0], t1 = center[r1 * stride + c1];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------------------------------
 * E1: operator().  ref: src/ORBextractor.cc:1045-1153
 * ---------------------------------------------------------------------------------------- */
struct orbo_extractor {
    orbo_params P;
    int img_w, img_h;
    int lw[ORBO_MAX_LEVELS], lh[ORBO_MAX_LEVELS];
    uint8_t *pyr[ORBO_MAX_LEVELS];
    uint8_t *blur[ORBO_MAX_LEVELS];
    orbo_cand *cands[ORBO_MAX_LEVELS];
    int ncands[ORBO_MAX_LEVELS];
    orbo_keypoint *lkps[ORBO_MAX_LEVELS];
    int nlkps[ORBO_MAX_LEVELS];
};

orbo_extractor *orbo_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                            int minThFAST)
{
    orbo_extractor *e = (orbo_extractor *)calloc(1, sizeof(*e));
    if (!e) return NULL;
    if (orbo_params_init(&e->P, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) != 0) {
        free(e);
        return NULL;
    }
    return e;
}

static void orbo_free_frame(orbo_extractor *e)
{
    for (int l = 0; l < ORBO_MAX_LEVELS; l++) {
        free(e->pyr[l]);
        free(e->blur[l]);
        free(e->cands[l]);
        free(e->lkps[l]);
        e->pyr[l] = e->blur[l] = NULL;
        e->cands[l] = NULL;
        e->lkps[l] = NULL;
        e->ncands[l] = e->nlkps[l] = 0;
    }
}

void orbo_destroy(orbo_extractor *e)
{
    if (!e) return;
    orbo_free_frame(e);
    free(e);
}

const orbo_params *orbo_get_params(const orbo_extractor *e) { return &e->P; }

int orbo_extract(orbo_extractor *e, const uint8_t *img, int w, int h, int stride,
                 orbo_keypoint *kps, uint8_t *desc, int cap)
{
    if (!e || !img || w <= 0 || h <= 0 || stride < w) return -1;
    const orbo_params *P = &e->P;
    orbo_free_frame(e);
    e->img_w = w;
    e->img_h = h;
    /* E2 ComputePyramid :1128-1153.  The 19-px border is never read downstream (survey A2). */
    for (int l = 0; l < P->nlevels; l++) {
        orbo_level_size(P, w, h, l, &e->lw[l], &e->lh[l]);
        if (e->lw[l] < 1 || e->lh[l] < 1) return -2;
        e->pyr[l] = (uint8_t *)malloc((size_t)e->lw[l] * (size_t)e->lh[l]);
        if (l == 0)
            for (int y = 0; y < h; y++) memcpy(e->pyr[0] + (size_t)y * w, img + (size_t)y * stride, (size_t)w);
        else
            orbo_resize_linear_u8(e->pyr[l - 1], e->lw[l - 1], e->lh[l - 1], e->lw[l - 1], e->pyr[l],
                                  e->lw[l], e->lh[l], e->lw[l]);
    }
    /* E3/E4 ComputeKeyPointsOctTree :767-855 */
    for (int l = 0; l < P->nlevels; l++) {
        const int lw = e->lw[l], lh = e->lh[l];
        int ccap = (lw * lh) / 4 + 64;
        e->cands[l] = (orbo_cand *)malloc(sizeof(orbo_cand) * (size_t)ccap);
        int nc = orbo_level_candidates(e->pyr[l], lw, lh, lw, P->iniThFAST, P->minThFAST, e->cands[l], ccap);
        if (nc < 0) return nc;
        e->ncands[l] = nc;
        const int minB = EDGE_THRESHOLD - 3;
        const int regw = (lw - EDGE_THRESHOLD + 3) - minB, regh = (lh - EDGE_THRESHOLD + 3) - minB;
        int *sel = (int *)malloc(sizeof(int) * (size_t)(nc + 1));
        int ns = orbo_distribute_octtree(e->cands[l], nc, regw, regh, P->mnFeaturesPerLevel[l], sel, nc + 1);
        if (ns < 0) {
            free(sel);
            return ns;
        }
        e->lkps[l] = (orbo_keypoint *)malloc(sizeof(orbo_keypoint) * (size_t)(ns + 1));
        e->nlkps[l] = ns;
        const int scaledPatchSize = (int)(PATCH_SIZE * P->mvScaleFactor[l]); /* :836 */
        for (int i = 0; i < ns; i++) {
            const orbo_cand *c = &e->cands[l][sel[i]];
            orbo_keypoint *k = &e->lkps[l][i];
            k->x = (float)c->x + (float)minB; /* :843-844 */
            k->y = (float)c->y + (float)minB;
            k->size = (float)scaledPatchSize;
            k->angle = -1.f;
            k->response = (float)c->score;
            k->octave = l;
            k->class_id = -1;
        }
        free(sel);
    }
    /* E5 :852-854 */
    for (int l = 0; l < P->nlevels; l++)
        for (int i = 0; i < e->nlkps[l]; i++) {
            orbo_keypoint *k = &e->lkps[l][i];
            k->angle = orbo_ic_angle(e->pyr[l], e->lw[l], orbo_cvround((double)k->x),
                                     orbo_cvround((double)k->y), P->umax);
        }
    /* E6/E7/E8 :1076-1122 */
    int n = 0;
    for (int l = 0; l < P->nlevels; l++) n += e->nlkps[l];
    if (n > cap) return -3;
    int off = 0;
    for (int l = 0; l < P->nlevels; l++) {
        const int nl = e->nlkps[l];
        if (nl == 0) continue;
        const int lw = e->lw[l], lh = e->lh[l];
        e->blur[l] = (uint8_t *)malloc((size_t)lw * (size_t)lh);
        orbo_gaussian_blur7_u8(e->pyr[l], lw, lh, lw, e->blur[l], lw);
        const float scale = P->mvScaleFactor[l];
        for (int i = 0; i < nl; i++) {
            orbo_keypoint k = e->lkps[l][i];
            orbo_brief(e->blur[l], lw, orbo_cvround((double)k.x), orbo_cvround((double)k.y), k.angle,
                       desc + (size_t)(off + i) * 32);
            if (l != 0) {
                k.x *= scale;
                k.y *= scale;
            }
            kps[off + i] = k;
        }
        off += nl;
    }
    return n;
}

const uint8_t *orbo_pyramid_level(const orbo_extractor *e, int level, int *w, int *h, int *stride)
{
    if (!e || level < 0 || level >= e->P.nlevels) return NULL;
    if (w) *w = e->lw[level];
    if (h) *h = e->lh[level];
    if (stride) *stride = e->lw[level];
    return e->pyr[level];
}

const uint8_t *orbo_blurred_level(const orbo_extractor *e, int level, int *w, int *h, int *stride)
{
    if (!e || level < 0 || level >= e->P.nlevels) return NULL;
    if (w) *w = e->lw[level];
    if (h) *h = e->lh[level];
    if (stride) *stride = e->lw[level];
    return e->blur[level];
}

int orbo_level_cands(const orbo_extractor *e, int level, const orbo_cand **c)
{
    if (!e || level < 0 || level >= e->P.nlevels) return -1;
    if (c) *c = e->cands[level];
    return e->ncands[level];
}

int orbo_level_keypoints(const orbo_extractor *e, int level, const orbo_keypoint **k)
{
    if (!e || level < 0 || level >= e->P.nlevels) return -1;
    if (k) *k = e->lkps[level];
    return e->nlkps[level];
}

/* ------------------------------------------------------------------------------------------
 * M0: ORBmatcher::DescriptorDistance.  ref: src/ORBmatcher.cc:1675-1691 (SWAR popcount over
 * 8 x 32-bit words of a XOR b).
 * ---------------------------------------------------------------------------------------- */
int orbo_descriptor_distance(const uint8_t *a, const uint8_t *b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t wa, wb;
        memcpy(&wa, a + 4 * i, 4);
        memcpy(&wb, b + 4 * i, 4);
        uint32_t v = wa ^ wb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0x0F0F0F0Fu) * 0x01010101u) >> 24);
    }
    return dist;
}

/* M3 shape: bestDist=256, bestIdx=-1, bestDist2=256; strict '<' updates
 * (ref: src/ORBmatcher.cc:205-226 and the other search routines listed in SURVEY 8a M3). */
void orbo_knn2(const uint8_t *q, int nq, const uint8_t *db, int ndb, int32_t *best_idx,
               int32_t *best_d, int32_t *second_d)
{
    for (int i = 0; i < nq; i++) {
        int b1 = 256, b2 = 256, bi = -1;
        for (int j = 0; j < ndb; j++) {
            int d = orbo_descriptor_distance(q + (size_t)i * 32, db + (size_t)j * 32);
            if (d < b1) {
                b2 = b1;
                b1 = d;
                bi = j;
            } else if (d < b2)
                b2 = d;
        }
        best_idx[i] = bi;
        best_d[i] = b1;
        second_d[i] = b2;
    }
}

void orbo_knn2_lists(const uint8_t *q, int nq, const uint8_t *db, const int32_t *off,
                     const int32_t *cand, int32_t *best_idx, int32_t *best_d, int32_t *second_d)
{
    for (int i = 0; i < nq; i++) {
        int b1 = 256, b2 = 256, bi = -1;
        for (int t = off[i]; t < off[i + 1]; t++) {
            int j = cand[t];
            int d = orbo_descriptor_distance(q + (size_t)i * 32, db + (size_t)j * 32);
            if (d < b1) {
                b2 = b1;
                b1 = d;
                bi = j;
            } else if (d < b2)
                b2 = d;
        }
        best_idx[i] = bi;
        best_d[i] = b1;
        second_d[i] = b2;
    }
}

/* ref: src/ORBmatcher.cc:1629-1670 */
void orbo_three_maxima(const int *hs, int L, int *ind1, int *ind2, int *ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    int i1 = -1, i2 = -1, i3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = hs[i];
        if (s > max1) {
            max3 = max2;
            max2 = max1;
            max1 = s;
            i3 = i2;
            i2 = i1;
            i1 = i;
        } else if (s > max2) {
            max3 = max2;
            max2 = s;
            i3 = i2;
            i2 = i;
        } else if (s > max3) {
            max3 = s;
            i3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) {
        i2 = -1;
        i3 = -1;
    } else if ((float)max3 < 0.1f * (float)max1) {
        i3 = -1;
    }
    *ind1 = i1;
    *ind2 = i2;
    *ind3 = i3;
}

/* ------------------------------------------------------------------------------------------
 * M1/M2: SearchByBoW.  ref: src/ORBmatcher.cc:159-288 (KeyFrame,Frame) and :522-655 (KF,KF).
 * Greedy, order dependent: side-1 features are visited node by node in FeatureVector order and
 * claim side-2 features; a claimed side-2 feature is skipped by later side-1 features.
 * Canonical FeatureVector order = ascending feature index inside a node (Appendix C.2); the
 * caller supplies idx arrays in the order it wants reproduced.
 * ---------------------------------------------------------------------------------------- */
#define HISTO_LENGTH 30

int orbo_search_by_bow(const uint8_t *desc1, int n1, const uint8_t *valid1, const float *angle1,
                       const int32_t *node1, const int32_t *off1, const int32_t *idx1, int ng1,
                       const uint8_t *desc2, int n2, const uint8_t *valid2, const float *angle2,
                       const int32_t *node2, const int32_t *off2, const int32_t *idx2, int ng2,
                       int th, int th_mode, float nnratio, int check_ori, int32_t *match12,
                       int32_t *match21)
{
    for (int i = 0; i < n1; i++) match12[i] = -1;
    for (int i = 0; i < n2; i++) match21[i] = -1;
    int nmatches = 0;
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) {
        hist[i] = (int *)malloc(sizeof(int) * (size_t)(n1 + 1));
        hn[i] = 0;
    }
    const float factor = 1.0f / HISTO_LENGTH;
    int g1 = 0, g2 = 0;
    while (g1 < ng1 && g2 < ng2) {
        if (node1[g1] == node2[g2]) {
            for (int a = off1[g1]; a < off1[g1 + 1]; a++) {
                const int i1 = idx1[a];
                if (!valid1[i1]) continue;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int b = off2[g2]; b < off2[g2 + 1]; b++) {
                    const int i2 = idx2[b];
                    if (match21[i2] >= 0) continue;
                    if (valid2 && !valid2[i2]) continue;
                    const int dist = orbo_descriptor_distance(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
                    if (dist < bestDist1) {
                        bestDist2 = bestDist1;
                        bestDist1 = dist;
                        bestIdx2 = i2;
                    } else if (dist < bestDist2)
                        bestDist2 = dist;
                }
                const int pass = th_mode ? (bestDist1 < th) : (bestDist1 <= th);
                if (pass && (float)bestDist1 < nnratio * (float)bestDist2) {
                    match12[i1] = bestIdx2;
                    match21[bestIdx2] = i1;
                    if (check_ori) {
                        float rot = angle1[i1] - angle2[bestIdx2];
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        if (bin >= 0 && bin < HISTO_LENGTH) hist[bin][hn[bin]++] = i1;
                    }
                    nmatches++;
                }
            }
            g1++;
            g2++;
        } else if (node1[g1] < node2[g2]) {
            while (g1 < ng1 && node1[g1] < node2[g2]) g1++; /* lower_bound */
        } else {
            while (g2 < ng2 && node2[g2] < node1[g1]) g2++;
        }
    }
    if (check_ori) {
        int i1, i2, i3;
        orbo_three_maxima(hn, HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int j = 0; j < hn[i]; j++) {
                int a = hist[i][j];
                if (match12[a] >= 0) match21[match12[a]] = -1;
                match12[a] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(hist[i]);
    return nmatches;
}

/* ------------------------------------------------------------------------------------------
 * Next row: ORB vocabulary tree -- load, per-feature transform, BowVector.
 * ref: Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1680-1721 (load), :1443-1485 (transform),
 * :1167-1258 (feature-set transform), BowVector.cpp:34-86, FORB.cpp:82-103 (distance = Hamming).
 * ---------------------------------------------------------------------------------------- */
struct orbo_vocab {
    int k, L, scoring, weighting;
    int nnodes;           /* including the root (id 0) */
    int nwords;
    int32_t *parent;      /* [nnodes] */
    uint8_t *desc;        /* [nnodes][32] */
    float *weight;        /* [nnodes] */
    uint8_t *leaf;        /* [nnodes] */
    int32_t *word;        /* [nnodes] word id of a leaf, -1 otherwise */
    int32_t *child_off;   /* [nnodes + 1] */
    int32_t *child;       /* children in file order */
};

orbo_vocab *orbo_vocab_load(const void *blob, size_t nbytes)
{
    const uint8_t *b = (const uint8_t *)blob;
    if (!b || nbytes < 24) return NULL;
    uint32_t nb_nodes, size_node;
    int32_t hdr[4];
    memcpy(&nb_nodes, b, 4);
    memcpy(&size_node, b + 4, 4);
    memcpy(hdr, b + 8, 16);
    if (size_node != 41 || nb_nodes < 1) return NULL;
    const size_t n = (nbytes - 24) / 41;
    if (n != (size_t)nb_nodes - 1) return NULL;
    orbo_vocab *v = (orbo_vocab *)calloc(1, sizeof(*v));
    v->k = hdr[0];
    v->L = hdr[1];
    v->scoring = hdr[2];
    v->weighting = hdr[3];
    v->nnodes = (int)nb_nodes;
    v->parent = (int32_t *)calloc(nb_nodes, 4);
    v->desc = (uint8_t *)calloc(nb_nodes, 32);
    v->weight = (float *)calloc(nb_nodes, 4);
    v->leaf = (uint8_t *)calloc(nb_nodes, 1);
    v->word = (int32_t *)malloc(4 * (size_t)nb_nodes);
    v->child_off = (int32_t *)calloc((size_t)nb_nodes + 1, 4);
    v->child = (int32_t *)calloc(nb_nodes, 4);
    int *cnt = (int *)calloc(nb_nodes, sizeof(int));
    for (uint32_t id = 1; id < nb_nodes; id++) {
        const uint8_t *r = b + 24 + (size_t)(id - 1) * 41;
        memcpy(&v->parent[id], r, 4);
        memcpy(v->desc + (size_t)id * 32, r + 4, 32);
        memcpy(&v->weight[id], r + 36, 4);
        v->leaf[id] = r[40] ? 1 : 0;
        if (v->parent[id] < 0 || v->parent[id] >= (int32_t)nb_nodes) {
            free(cnt);
            orbo_vocab_free(v);
            return NULL;
        }
        cnt[v->parent[id]]++;
    }
    for (uint32_t i = 0; i < nb_nodes; i++) v->child_off[i + 1] = v->child_off[i] + cnt[i];
    memset(cnt, 0, sizeof(int) * nb_nodes);
    int nw = 0;
    for (uint32_t id = 0; id < nb_nodes; id++) v->word[id] = -1;
    for (uint32_t id = 1; id < nb_nodes; id++) {
        const int p = v->parent[id];
        v->child[v->child_off[p] + cnt[p]++] = (int32_t)id;   /* push_back in file order, :1703 */
        if (v->leaf[id]) v->word[id] = nw++;                  /* :1707-1712 */
    }
    v->nwords = nw;
    free(cnt);
    return v;
}

void orbo_vocab_free(orbo_vocab *v)
{
    if (!v) return;
    free(v->parent);
    free(v->desc);
    free(v->weight);
    free(v->leaf);
    free(v->word);
    free(v->child_off);
    free(v->child);
    free(v);
}

int orbo_vocab_info(const orbo_vocab *v, int *k, int *L, int *scoring, int *weighting, int *nnodes, int *nwords)
{
    if (!v) return -1;
    if (k) *k = v->k;
    if (L) *L = v->L;
    if (scoring) *scoring = v->scoring;
    if (weighting) *weighting = v->weighting;
    if (nnodes) *nnodes = v->nnodes;
    if (nwords) *nwords = v->nwords;
    return 0;
}

void orbo_vocab_transform(const orbo_vocab *v, const uint8_t *desc, int n, int levelsup, int32_t *word_id,
                          float *weight, int32_t *node_id)
{
    const int nid_level = v->L - levelsup;
    for (int i = 0; i < n; i++) {
        const uint8_t *f = desc + (size_t)i * 32;
        int nid = 0;             /* "root" when nid_level <= 0 (:1454); also when the level is never reached */
        int final_id = 0, current_level = 0;
        do {                     /* :1459-1480 */
            ++current_level;
            const int c0 = v->child_off[final_id], c1 = v->child_off[final_id + 1];
            if (c0 == c1) break; /* malformed tree: inner node without children */
            final_id = v->child[c0];
            int best_d = orbo_descriptor_distance(f, v->desc + (size_t)final_id * 32);
            for (int c = c0 + 1; c < c1; c++) {
                const int id = v->child[c];
                const int d = orbo_descriptor_distance(f, v->desc + (size_t)id * 32);
                if (d < best_d) {
                    best_d = d;
                    final_id = id;
                }
            }
            if (current_level == nid_level) nid = final_id;
        } while (!v->leaf[final_id]);
        word_id[i] = v->word[final_id];
        weight[i] = v->weight[final_id];
        node_id[i] = nid;
    }
}

typedef struct {
    int32_t w;
    double val;
} bow_ent;

static int bow_cmp(const void *a, const void *b)
{
    const bow_ent *x = (const bow_ent *)a, *y = (const bow_ent *)b;
    return x->w < y->w ? -1 : (x->w > y->w ? 1 : 0);
}

int orbo_vocab_bow(const orbo_vocab *v, const int32_t *word_id, const float *weight, int n, int32_t *out_word,
                   double *out_value)
{
    /* weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; scoring: 0 L1_NORM 1 L2_NORM 2 CHI_SQUARE 3 KL
     * 4 BHATTACHARYYA 5 DOT_PRODUCT (BowVector.h enums) */
    const int accumulate = (v->weighting == 0 || v->weighting == 1);
    const int must = v->scoring != 5;
    const int l2 = v->scoring == 1;
    bow_ent *e = (bow_ent *)malloc(sizeof(bow_ent) * (size_t)(n + 1));
    int m = 0;
    /* stable accumulation in ascending feature order: sort (word, index) then sum runs in index order */
    int *order = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    int cnt = 0;
    for (int i = 0; i < n; i++)
        if (weight[i] > 0 && word_id[i] >= 0) order[cnt++] = i; /* "not stopped", :1334 */
    /* insertion into a std::map keyed by word: equivalent to sorting by word, keeping index order */
    for (int a = 1; a < cnt; a++) {
        int x = order[a], b = a - 1;
        while (b >= 0 && word_id[order[b]] > word_id[x]) {
            order[b + 1] = order[b];
            b--;
        }
        order[b + 1] = x;
    }
    for (int a = 0; a < cnt; a++) {
        const int i = order[a];
        if (m > 0 && e[m - 1].w == word_id[i]) {
            if (accumulate) e[m - 1].val += (double)weight[i]; /* addWeight; addIfNotExist keeps the first */
        } else {
            e[m].w = word_id[i];
            e[m].val = (double)weight[i];
            m++;
        }
    }
    free(order);
    if (accumulate && m > 0 && !must) { /* :1226-1232 */
        const double nd = (double)m;
        for (int a = 0; a < m; a++) e[a].val /= nd;
    }
    if (must) { /* BowVector::normalize */
        double norm = 0.0;
        if (!l2)
            for (int a = 0; a < m; a++) norm += fabs(e[a].val);
        else {
            for (int a = 0; a < m; a++) norm += e[a].val * e[a].val;
            norm = sqrt(norm);
        }
        if (norm > 0.0)
            for (int a = 0; a < m; a++) e[a].val /= norm;
    }
    qsort(e, (size_t)m, sizeof(bow_ent), bow_cmp); /* already sorted; keeps the contract explicit */
    for (int a = 0; a < m; a++) {
        out_word[a] = e[a].w;
        out_value[a] = e[a].val;
    }
    free(e);
    return m;
}

/* ------------------------------------------------------------------------------------------
 * Next row: Frame::ComputeStereoMatches.  ref: src/Frame.cc:810-984.
 * Row-band table of the right keypoints (:819-837), per left keypoint the minimum descriptor distance
 * over the band with octave +-1 and u in [uL - maxD, uL] (:848-893), 11-shift SAD on 11x11 patches of
 * the pyramid level of the left keypoint (:896-936; cv::norm NORM_L1 of centre-subtracted float patches:
 * every term is an integer, so the sum is exact), parabola sub-pixel fit (:938-946), disparity -> depth
 * (:948-966), then the median-based outlier cut (:970-983).
 * Out-of-range row indices (undefined behaviour in the reference) are skipped; an empty match set
 * (the reference would index vDistIdx[0] of an empty vector) returns without the cut.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int dist, idx;
} sm_pair;

static int sm_cmp(const void *a, const void *b)
{
    const sm_pair *x = (const sm_pair *)a, *y = (const sm_pair *)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}

int orbo_stereo_matches(const orbo_keypoint *kL, const uint8_t *dL, int nL, const orbo_keypoint *kR,
                        const uint8_t *dR, int nR, const uint8_t *const *pyrL, const uint8_t *const *pyrR,
                        const int *lw, const int *lh, const float *mvScaleFactors, const float *mvInvScaleFactors,
                        float mb, float mbf, float *mvuRight, float *mvDepth)
{
    const int TH_HIGH = 100, TH_LOW = 50;
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = lh[0];
    for (int i = 0; i < nL; i++) {
        mvuRight[i] = -1.0f;
        mvDepth[i] = -1.0f;
    }
    /* row table: counts, then CSR */
    int *cnt = (int *)calloc((size_t)nRows + 1, sizeof(int));
    for (int iR = 0; iR < nR; iR++) {
        const float kpY = kR[iR].y;
        const float r = 2.0f * mvScaleFactors[kR[iR].octave];
        const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) cnt[yi + 1]++;
    }
    for (int i = 0; i < nRows; i++) cnt[i + 1] += cnt[i];
    int *rows = (int *)malloc(sizeof(int) * (size_t)(cnt[nRows] + 1));
    int *fill = (int *)calloc((size_t)nRows, sizeof(int));
    for (int iR = 0; iR < nR; iR++) {
        const float kpY = kR[iR].y;
        const float r = 2.0f * mvScaleFactors[kR[iR].octave];
        const int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) rows[cnt[yi] + fill[yi]++] = iR;
    }
    free(fill);
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    sm_pair *vDistIdx = (sm_pair *)malloc(sizeof(sm_pair) * (size_t)(nL + 1));
    int nd = 0;
    for (int iL = 0; iL < nL; iL++) {
        const int levelL = kL[iL].octave;
        const float vL = kL[iL].y, uL = kL[iL].x;
        const int rowi = (int)vL;
        if (rowi < 0 || rowi >= nRows) continue;
        if (cnt[rowi + 1] == cnt[rowi]) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH, bestIdxR = 0;
        for (int c = cnt[rowi]; c < cnt[rowi + 1]; c++) {
            const int iR = rows[c];
            if (kR[iR].octave < levelL - 1 || kR[iR].octave > levelL + 1) continue;
            const float uR = kR[iR].x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orbo_descriptor_distance(dL + (size_t)iL * 32, dR + (size_t)iR * 32);
                if (dist < bestDist) {
                    bestDist = dist;
                    bestIdxR = iR;
                }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = kR[bestIdxR].x;
            const float scaleFactor = mvInvScaleFactors[levelL];
            const float scaleduL = roundf(kL[iL].x * scaleFactor);
            const float scaledvL = roundf(kL[iL].y * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const uint8_t *imL = pyrL[levelL], *imR = pyrR[levelL];
            const int W = lw[levelL];
            const int cy = (int)scaledvL, cxL = (int)scaleduL, cxR = (int)scaleduR0;
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= (float)W) continue;
            int sadBest = 2147483647, bestincR = 0;
            float vDists[11];
            const int cLv = imL[(size_t)cy * W + cxL];
            for (int incR = -L; incR <= +L; incR++) {
                const int cRv = imR[(size_t)cy * W + cxR + incR];
                long sum = 0;
                for (int dy = -w; dy <= w; dy++)
                    for (int dx = -w; dx <= w; dx++) {
                        const int a = imL[(size_t)(cy + dy) * W + cxL + dx] - cLv;
                        const int b = imR[(size_t)(cy + dy) * W + cxR + incR + dx] - cRv;
                        sum += labs((long)(a - b));
                    }
                const float dist = (float)sum;
                if (dist < (float)sadBest) {
                    sadBest = (int)dist;
                    bestincR = incR;
                }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = mvScaleFactors[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = 0.01f;
                    bestuR = (float)((double)uL - 0.01); /* `uL-0.01` is evaluated in double, :957 */
                }
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = bestuR;
                vDistIdx[nd].dist = sadBest;
                vDistIdx[nd].idx = iL;
                nd++;
            }
        }
    }
    if (nd > 0) {
        qsort(vDistIdx, (size_t)nd, sizeof(sm_pair), sm_cmp);
        const float median = (float)vDistIdx[nd / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = nd - 1; i >= 0; i--) {
            if ((float)vDistIdx[i].dist < thDist) break;
            mvuRight[vDistIdx[i].idx] = -1;
            mvDepth[vDistIdx[i].idx] = -1;
        }
    }
    free(vDistIdx);
    free(rows);
    free(cnt);
    return nd;
}

/* ------------------------------------------------------------------------------------------
 * Next row: the frame grid and the guided (projection) search built on it.
 * ref: src/Frame.cc:574-589 (AssignFeaturesToGrid), :726-736 (PosInGrid), :671-724
 * (GetFeaturesInArea), include/Frame.h:41-42 (64 x 48), src/ORBmatcher.cc:45-129
 * (SearchByProjection(Frame&, vector<MapPoint*>&, th)), :1341-1498 (SearchByProjection(Current,
 * Last, th, bMono)).  The projection itself (pose * point) stays with the caller; a query carries
 * the projected position, the search radius and the octave range the caller derived from it.
 * ---------------------------------------------------------------------------------------- */
#define GRID_COLS 64
#define GRID_ROWS 48

/* cell id = ix * GRID_ROWS + iy: the order GetFeaturesInArea visits cells in (ix outer, iy inner) */
void orbo_grid_build(const orbo_keypoint *kps, int n, float minX, float minY, float invW, float invH,
                     int32_t *cell_off, int32_t *cell_idx)
{
    int *cell = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    for (int c = 0; c <= GRID_COLS * GRID_ROWS; c++) cell_off[c] = 0;
    for (int i = 0; i < n; i++) {
        /* PosInGrid, :728-735: float arithmetic, round half away from zero */
        const int px = (int)roundf((kps[i].x - minX) * invW);
        const int py = (int)roundf((kps[i].y - minY) * invH);
        cell[i] = (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) ? -1 : px * GRID_ROWS + py;
        if (cell[i] >= 0) cell_off[cell[i] + 1]++;
    }
    for (int c = 0; c < GRID_COLS * GRID_ROWS; c++) cell_off[c + 1] += cell_off[c];
    int *cur = (int *)malloc(sizeof(int) * GRID_COLS * GRID_ROWS);
    for (int c = 0; c < GRID_COLS * GRID_ROWS; c++) cur[c] = cell_off[c];
    for (int i = 0; i < n; i++)                      /* push_back in ascending i, :581-588 */
        if (cell[i] >= 0) cell_idx[cur[cell[i]]++] = i;
    free(cur);
    free(cell);
}

int orbo_features_in_area(const orbo_keypoint *kps, const int32_t *cell_off, const int32_t *cell_idx, float minX,
                          float minY, float invW, float invH, float x, float y, float r, int minLevel, int maxLevel,
                          int32_t *out, int cap)
{
    int n = 0;
    int nMinCellX = (int)floorf((x - minX - r) * invW);          /* :676-691 */
    if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= GRID_COLS) return 0;
    int nMaxCellX = (int)ceilf((x - minX + r) * invW);
    if (nMaxCellX > GRID_COLS - 1) nMaxCellX = GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = (int)floorf((y - minY - r) * invH);
    if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= GRID_ROWS) return 0;
    int nMaxCellY = (int)ceilf((y - minY + r) * invH);
    if (nMaxCellY > GRID_ROWS - 1) nMaxCellY = GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);  /* :693 */
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * GRID_ROWS + iy;
            for (int j = cell_off[c]; j < cell_off[c + 1]; j++) {
                const orbo_keypoint *k = &kps[cell_idx[j]];
                if (bCheckLevels) {
                    if (k->octave < minLevel) continue;
                    if (maxLevel >= 0 && k->octave > maxLevel) continue;
                }
                const float distx = k->x - x, disty = k->y - y;
                if (fabsf(distx) < r && fabsf(disty) < r) {
                    if (n < cap) out[n] = cell_idx[j];
                    n++;
                }
            }
        }
    return n;
}

/* Both SearchByProjection variants as one loop over queries in order.
 *   use_ratio = 1: :45-129  (second best + ratio when best and second lie on the same level)
 *   use_ratio = 0: :1341-1498 (best only; rotation histogram when check_ori)
 * occupied[i] != 0: the frame feature already holds a MapPoint with Observations() > 0 (:88-90, :1413-1415).
 * match[i]: index of the last query assigned to feature i, -1 = never assigned, -2 = assigned and then
 * removed by the rotation check.
 * A match of query q sets match[best] = q and, when the query's point has observations
 * (ORBO_Q_OBSERVED), makes the feature occupied for the queries after it.  u_right may be NULL (mono). */
int orbo_search_by_projection(const orbo_keypoint *kps, const uint8_t *desc, int n, const float *u_right,
                              const uint8_t *occupied_in, float minX, float minY, float invW, float invH,
                              const orbo_proj_query *q, const uint8_t *qdesc, int nq, int use_ratio, float nnratio,
                              int check_ori, int th_high, int32_t *match)
{
    int32_t *cell_off = (int32_t *)malloc(sizeof(int32_t) * (GRID_COLS * GRID_ROWS + 1));
    int32_t *cell_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    uint8_t *occ = (uint8_t *)calloc((size_t)n + 1, 1);
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) {
        hist[i] = (int *)malloc(sizeof(int) * (size_t)(nq + 1));
        hn[i] = 0;
    }
    const float factor = 1.0f / HISTO_LENGTH;
    orbo_grid_build(kps, n, minX, minY, invW, invH, cell_off, cell_idx);
    for (int i = 0; i < n; i++) {
        match[i] = -1;
        if (occupied_in) occ[i] = occupied_in[i] != 0;
    }
    int nmatches = 0;
    for (int iq = 0; iq < nq; iq++) {
        if (!(q[iq].flags & ORBO_Q_ACTIVE)) continue;
        const int nc = orbo_features_in_area(kps, cell_off, cell_idx, minX, minY, invW, invH, q[iq].u, q[iq].v,
                                             q[iq].radius, q[iq].min_level, q[iq].max_level, cand, n);
        if (nc == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            if (occ[idx]) continue;
            if (u_right && u_right[idx] > 0) {
                const float er = fabsf(q[iq].proj_xr - u_right[idx]);
                if (er > q[iq].radius) continue;
            }
            const int dist = orbo_descriptor_distance(qdesc + (size_t)iq * 32, desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestLevel2 = bestLevel;
                bestLevel = kps[idx].octave;
                bestIdx = idx;
            } else if (dist < bestDist2) {
                bestLevel2 = kps[idx].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= th_high) {
            if (use_ratio && bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
            match[bestIdx] = iq;
            if (q[iq].flags & ORBO_Q_OBSERVED) occ[bestIdx] = 1;
            nmatches++;
            if (!use_ratio && check_ori) {
                float rot = q[iq].angle - kps[bestIdx].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < HISTO_LENGTH) hist[bin][hn[bin]++] = bestIdx;
            }
        }
    }
    if (!use_ratio && check_ori) {
        int i1, i2, i3;
        orbo_three_maxima(hn, HISTO_LENGTH, &i1, &i2, &i3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == i1 || i == i2 || i == i3) continue;
            for (int j = 0; j < hn[i]; j++) {
                match[hist[i][j]] = -2; /* the reference stores NULL (:1489); -1 = never touched */
                nmatches--;
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(hist[i]);
    free(occ);
    free(cand);
    free(cell_idx);
    free(cell_off);
    return nmatches;
}

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:283-349) for P points at once: point p's observed descriptors
 * are rows off[p] .. off[p+1] of desc (the order of its observation map); all pair distances, per row the median
 * vDists[0.5*(N-1)] of the sorted row (self-distance 0 included), the first row of least median wins.
 * best[p] = row index within the point's list (-1 for an empty list), best_median[p] its median. */
void orbo_distinctive_descriptors(const uint8_t *desc, const int32_t *off, int P, int32_t *best, int32_t *best_median)
{
    for (int p = 0; p < P; p++) {
        const int N = off[p + 1] - off[p];
        const uint8_t *D = desc + (size_t)off[p] * 32;
        best[p] = -1;
        best_median[p] = INT_MAX;
        if (N <= 0) continue;
        int *dist = (int *)malloc(sizeof(int) * (size_t)N * (size_t)N);
        int *row = (int *)malloc(sizeof(int) * (size_t)N);
        for (int i = 0; i < N; i++) {
            dist[(size_t)i * N + i] = 0;
            for (int j = i + 1; j < N; j++) {
                const int d = orbo_descriptor_distance(D + (size_t)i * 32, D + (size_t)j * 32);
                dist[(size_t)i * N + j] = d;
                dist[(size_t)j * N + i] = d;
            }
        }
        int BestMedian = INT_MAX, BestIdx = 0;
        for (int i = 0; i < N; i++) {
            for (int j = 0; j < N; j++) { /* insertion sort of row i */
                int v = dist[(size_t)i * N + j], k = j;
                while (k > 0 && row[k - 1] > v) {
                    row[k] = row[k - 1];
                    k--;
                }
                row[k] = v;
            }
            const int median = row[(size_t)(0.5 * (N - 1))];
            if (median < BestMedian) {
                BestMedian = median;
                BestIdx = i;
            }
        }
        best[p] = BestIdx;
        best_median[p] = BestMedian;
        free(dist);
        free(row);
    }
}

/* The per-point inner loop shared by ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:887-950:
 * chi-square gate on the reprojection error when inv_level_sigma2 != NULL), Fuse(KeyFrame*, Scw, ...) (:1044-1075) and
 * SearchBySim3 (:1190-1224, :1270-1304; the gate is off there): KeyFrame::GetFeaturesInArea(u, v, radius)
 * (src/KeyFrame.cc:1138-1177), levels outside [min_level, max_level] = [predicted - 1, predicted] skipped, the first
 * feature of smallest distance wins (strict '<').  Every query is independent of the others.
 * best_idx[q] = -1 / best_dist[q] = 256 when the query is inactive or nothing is closer than 256 (the reference starts
 * from 256 in Fuse and INT_MAX in SearchBySim3; both then require <= TH_LOW / TH_HIGH, so the outcome is the same). */
void orbo_window_best(const orbo_keypoint *kps, const uint8_t *desc, int n, const float *u_right,
                      const float *inv_level_sigma2, float minX, float minY, float invW, float invH,
                      const orbo_proj_query *q, const uint8_t *qdesc, int nq, int32_t *best_idx, int32_t *best_dist)
{
    int32_t *cell_off = (int32_t *)malloc(sizeof(int32_t) * (GRID_COLS * GRID_ROWS + 1));
    int32_t *cell_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + 1));
    orbo_grid_build(kps, n, minX, minY, invW, invH, cell_off, cell_idx);
    for (int iq = 0; iq < nq; iq++) {
        best_idx[iq] = -1;
        best_dist[iq] = 256;
        if (!(q[iq].flags & ORBO_Q_ACTIVE)) continue;
        const float u = q[iq].u, v = q[iq].v, ur = q[iq].proj_xr;
        /* the KeyFrame grid query has no level filter; the loop below applies it */
        const int nc = orbo_features_in_area(kps, cell_off, cell_idx, minX, minY, invW, invH, u, v, q[iq].radius, -1, -1,
                                             cand, n);
        int bestDist = 256, bestIdx = -1;
        for (int c = 0; c < nc; c++) {
            const int idx = cand[c];
            const int kpLevel = kps[idx].octave;
            if (kpLevel < q[iq].min_level || kpLevel > q[iq].max_level) continue;
            if (inv_level_sigma2) {
                const float kpx = kps[idx].x, kpy = kps[idx].y;
                const float ex = u - kpx, ey = v - kpy;
                if (u_right && u_right[idx] >= 0) { /* :911-923 */
                    const float er = ur - u_right[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
                } else { /* :925-934 */
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = orbo_descriptor_distance(qdesc + (size_t)iq * 32, desc + (size_t)idx * 32);
            if (dist < bestDist) {
                bestDist = dist;
                bestIdx = idx;
            }
        }
        best_idx[iq] = bestIdx;
        best_dist[iq] = bestDist;
    }
    free(cell_off);
    free(cell_idx);
    free(cand);
}

/* ORBmatcher::SearchForInitialization (ref: src/ORBmatcher.cc:405-520), the monocular initialiser's matcher
 * (src/Tracking.cc MonocularInitialization): level-0 features of frame 1 in index order, window search around
 * prev_matched[i1] among the level-0 features of frame 2 (:424), a feature of frame 2 already matched at a
 * distance <= dist is skipped (:443-444), best / second with strict '<' (:446-455), bestDist <= TH_LOW and
 * bestDist < (float)bestDist2 * nnratio (:458-460), an earlier owner of the feature loses it (:462-466), rotation
 * histogram over every accepted i1 -- displaced ones stay in it (:471-481) -- and removal of all bins but the three
 * maxima (:487-509); finally prev_matched[i1] becomes the matched keypoint's position (:512-515).
 * kps1/kps2 are the undistorted keypoints (mvKeysUn).  matches12[n1] = index into frame 2 or -1. */
int orbo_search_for_initialization(const orbo_keypoint *kps1, const uint8_t *desc1, int n1, const orbo_keypoint *kps2,
                                   const uint8_t *desc2, int n2, float minX, float minY, float invW, float invH,
                                   float *prev_matched /* n1 x 2, in/out */, int window_size, float nnratio,
                                   int check_ori, int th_low, int32_t *matches12)
{
    int32_t *cell_off = (int32_t *)malloc(sizeof(int32_t) * (GRID_COLS * GRID_ROWS + 1));
    int32_t *cell_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    int *matched_dist = (int *)malloc(sizeof(int) * (size_t)(n2 + 1));
    int32_t *matches21 = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n2 + 1));
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) {
        hist[i] = (int *)malloc(sizeof(int) * (size_t)(n1 + 1));
        hn[i] = 0;
    }
    const float factor = 1.0f / HISTO_LENGTH;
    orbo_grid_build(kps2, n2, minX, minY, invW, invH, cell_off, cell_idx);
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    for (int i = 0; i < n2; i++) {
        matched_dist[i] = INT_MAX;
        matches21[i] = -1;
    }
    int nmatches = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = kps1[i1].octave;
        if (level1 > 0) continue;
        const int nc = orbo_features_in_area(kps2, cell_off, cell_idx, minX, minY, invW, invH, prev_matched[2 * i1],
                                             prev_matched[2 * i1 + 1], (float)window_size, level1, level1, cand, n2);
        if (nc == 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int c = 0; c < nc; c++) {
            const int i2 = cand[c];
            const int dist = orbo_descriptor_distance(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
            if (matched_dist[i2] <= dist) continue;
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx2 = i2;
            } else if (dist < bestDist2) {
                bestDist2 = dist;
            }
        }
        if (bestDist <= th_low && (float)bestDist < (float)bestDist2 * nnratio) {
            if (matches21[bestIdx2] >= 0) {
                matches12[matches21[bestIdx2]] = -1;
                nmatches--;
            }
            matches12[i1] = bestIdx2;
            matches21[bestIdx2] = i1;
            matched_dist[bestIdx2] = bestDist;
            nmatches++;
            if (check_ori) {
                float rot = kps1[i1].angle - kps2[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                if (bin >= 0 && bin < HISTO_LENGTH) hist[bin][hn[bin]++] = i1;
            }
        }
    }
    if (check_ori) {
        int a, b, c;
        orbo_three_maxima(hn, HISTO_LENGTH, &a, &b, &c);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == a || i == b || i == c) continue;
            for (int j = 0; j < hn[i]; j++) {
                const int idx1 = hist[i][j];
                if (matches12[idx1] >= 0) {
                    matches12[idx1] = -1;
                    nmatches--;
                }
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (matches12[i1] >= 0) {
            prev_matched[2 * i1] = kps2[matches12[i1]].x;
            prev_matched[2 * i1 + 1] = kps2[matches12[i1]].y;
        }
    for (int i = 0; i < HISTO_LENGTH; i++) free(hist[i]);
    free(matches21);
    free(matched_dist);
    free(cand);
    free(cell_idx);
    free(cell_off);
    return nmatches;
}

/* ORBmatcher::SearchForTriangulation (ref: src/ORBmatcher.cc:657-827; the matcher of LocalMapping::CreateNewMapPoints)
 * with CheckDistEpipolarLine (:140-157).  Features of key frame 1 that hold no MapPoint search, inside their vocabulary
 * node, the features of key frame 2 that hold none: Hamming distance <= TH_LOW and <= the best so far (:719-720, a later
 * candidate at the same distance replaces the earlier one), not closer than 10 px * scale to the epipole when neither
 * feature is stereo (:724-730), distance to the epipolar line x1' F12 below 3.84 sigma2 (:732-736).  This fork never
 * sets vbMatched2, so a feature of key frame 2 may be the match of several features of key frame 1.  Rotation histogram
 * as in the other routines (:745-755, :775-794).  skip1/skip2[i] != 0: the feature holds a MapPoint; u_right1/2 may be
 * NULL (monocular: nothing is stereo).  matches12[n1] = index into key frame 2 or -1; returns nmatches. */
int orbo_search_for_triangulation(const orbo_keypoint *kps1, const uint8_t *desc1, int n1, const uint8_t *skip1,
                                  const float *u_right1, const int32_t *node1, const int32_t *off1, const int32_t *idx1,
                                  int ng1, const orbo_keypoint *kps2, const uint8_t *desc2, int n2, const uint8_t *skip2,
                                  const float *u_right2, const int32_t *node2, const int32_t *off2, const int32_t *idx2,
                                  int ng2, const float *F12, float ex, float ey, const float *scale_factors2,
                                  const float *level_sigma2_2, int only_stereo, int check_ori, int th_low,
                                  int32_t *matches12)
{
    int *hist[HISTO_LENGTH], hn[HISTO_LENGTH];
    for (int i = 0; i < HISTO_LENGTH; i++) {
        hist[i] = (int *)malloc(sizeof(int) * (size_t)(n1 + 1));
        hn[i] = 0;
    }
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    int nmatches = 0;
    int g1 = 0, g2 = 0;
    while (g1 < ng1 && g2 < ng2) {
        if (node1[g1] == node2[g2]) {
            for (int a = off1[g1]; a < off1[g1 + 1]; a++) {
                const int i1 = idx1[a];
                if (skip1[i1]) continue;
                const int bStereo1 = u_right1 && u_right1[i1] >= 0;
                if (only_stereo && !bStereo1) continue;
                const orbo_keypoint *kp1 = &kps1[i1];
                int bestDist = th_low, bestIdx2 = -1;
                for (int b = off2[g2]; b < off2[g2 + 1]; b++) {
                    const int i2 = idx2[b];
                    if (skip2[i2]) continue; /* vbMatched2 stays false in this fork */
                    const int bStereo2 = u_right2 && u_right2[i2] >= 0;
                    if (only_stereo && !bStereo2) continue;
                    const int dist = orbo_descriptor_distance(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
                    if (dist > th_low || dist > bestDist) continue;
                    const orbo_keypoint *kp2 = &kps2[i2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - kp2->x, distey = ey - kp2->y;
                        if (distex * distex + distey * distey < 100 * scale_factors2[kp2->octave]) continue;
                    }
                    /* CheckDistEpipolarLine */
                    const float la = kp1->x * F12[0] + kp1->y * F12[3] + F12[6];
                    const float lb = kp1->x * F12[1] + kp1->y * F12[4] + F12[7];
                    const float lc = kp1->x * F12[2] + kp1->y * F12[5] + F12[8];
                    const float num = la * kp2->x + lb * kp2->y + lc;
                    const float den = la * la + lb * lb;
                    if (den == 0) continue;
                    const float dsqr = num * num / den;
                    if (dsqr < 3.84 * level_sigma2_2[kp2->octave]) {
                        bestIdx2 = i2;
                        bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {
                    matches12[i1] = bestIdx2;
                    nmatches++;
                    if (check_ori) {
                        float rot = kp1->angle - kps2[bestIdx2].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        if (bin >= 0 && bin < HISTO_LENGTH) hist[bin][hn[bin]++] = i1;
                    }
                }
            }
            g1++;
            g2++;
        } else if (node1[g1] < node2[g2])
            g1++; /* lower_bound on a sorted map = advance */
        else
            g2++;
    }
    if (check_ori) {
        int a, b, c;
        orbo_three_maxima(hn, HISTO_LENGTH, &a, &b, &c);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == a || i == b || i == c) continue;
            for (int j = 0; j < hn[i]; j++) {
                matches12[hist[i][j]] = -1;
                nmatches--;
            }
        }
    }
    for (int i = 0; i < HISTO_LENGTH; i++) free(hist[i]);
    return nmatches;
}

/* ------------------------------------------------------------------------------------------
 * Next row: the steps either side of extraction.
 *  - Frame::UndistortKeyPoints (ref: src/Frame.cc:748-778) = cv::undistortPoints(mat, mat, mK, mDistCoef,
 *    cv::Mat(), mK): OpenCV 2.4 cvUndistortPoints (modules/imgproc/src/undistort.cpp) -- double arithmetic,
 *    five fixed-point iterations of the inverse distortion, then x' = P * x.
 *  - stereo rectification (ref: Examples/Stereo/stereo_euroc.cc:96-98, :136-137): cv::initUndistortRectifyMap
 *    (same file, double, CV_32FC1 maps) once, cv::remap(..., INTER_LINEAR) per image: OpenCV 2.4 imgwarp.cpp --
 *    maps to fixed point with 5 fractional bits (cvRound(m * 32)), bilinear weights in 1/32768, BORDER_CONSTANT 0.
 * [OpenCV-2.4 recollection; parity unpinned like the rest of the OpenCV boundary]
 * ---------------------------------------------------------------------------------------- */
void orbo_undistort_points(const float *xy_in, int n, const float *K, const float *D, int nD, const float *P,
                           float *xy_out)
{
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < nD && i < 8; i++) k[i] = (double)D[i];
    const double fx = (double)K[0], fy = (double)K[4], cx = (double)K[2], cy = (double)K[5];
    const double ifx = 1. / fx, ify = 1. / fy;
    double RR[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    if (P)
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) RR[r][c] = (double)P[r * 3 + c]; /* P * I */
    const int iters = (D && nD > 0) ? 5 : 1;
    for (int i = 0; i < n; i++) {
        double x = (double)xy_in[2 * i], y = (double)xy_in[2 * i + 1];
        const double x0 = x = (x - cx) * ifx;
        const double y0 = y = (y - cy) * ify;
        for (int j = 0; j < iters; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x);
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        const double xx = RR[0][0] * x + RR[0][1] * y + RR[0][2];
        const double yy = RR[1][0] * x + RR[1][1] * y + RR[1][2];
        const double ww = 1. / (RR[2][0] * x + RR[2][1] * y + RR[2][2]);
        xy_out[2 * i] = (float)(xx * ww);
        xy_out[2 * i + 1] = (float)(yy * ww);
    }
}

void orbo_init_undistort_rectify_map(const double *K, const double *D, int nD, const double *R, const double *P,
                                     int w, int h, float *mapx, float *mapy)
{
    /* iR = (P(:, 0:3) * R)^-1 : 3x3 product accumulated k = 0..2, inverse by cofactors (cv::invert, n == 3) */
    double M[9], ir[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += P[r * 3 + k] * R[k * 3 + c];
            M[r * 3 + c] = s;
        }
#define Sd(y, x) M[(y)*3 + (x)]
    double d = Sd(0, 0) * (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) - Sd(0, 1) * (Sd(1, 0) * Sd(2, 2) - Sd(1, 2) * Sd(2, 0)) +
               Sd(0, 2) * (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0));
    d = 1. / d;
    ir[0] = (Sd(1, 1) * Sd(2, 2) - Sd(1, 2) * Sd(2, 1)) * d;
    ir[1] = (Sd(0, 2) * Sd(2, 1) - Sd(0, 1) * Sd(2, 2)) * d;
    ir[2] = (Sd(0, 1) * Sd(1, 2) - Sd(0, 2) * Sd(1, 1)) * d;
    ir[3] = (Sd(1, 2) * Sd(2, 0) - Sd(1, 0) * Sd(2, 2)) * d;
    ir[4] = (Sd(0, 0) * Sd(2, 2) - Sd(0, 2) * Sd(2, 0)) * d;
    ir[5] = (Sd(0, 2) * Sd(1, 0) - Sd(0, 0) * Sd(1, 2)) * d;
    ir[6] = (Sd(1, 0) * Sd(2, 1) - Sd(1, 1) * Sd(2, 0)) * d;
    ir[7] = (Sd(0, 1) * Sd(2, 0) - Sd(0, 0) * Sd(2, 1)) * d;
    ir[8] = (Sd(0, 0) * Sd(1, 1) - Sd(0, 1) * Sd(1, 0)) * d;
#undef Sd
    const double u0 = K[2], v0 = K[5], fx = K[0], fy = K[4];
    const double k1 = nD > 0 ? D[0] : 0, k2 = nD > 1 ? D[1] : 0, p1 = nD > 2 ? D[2] : 0, p2 = nD > 3 ? D[3] : 0;
    const double k3 = nD >= 5 ? D[4] : 0, k4 = nD >= 8 ? D[5] : 0, k5 = nD >= 8 ? D[6] : 0, k6 = nD >= 8 ? D[7] : 0;
    for (int i = 0; i < h; i++) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < w; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double ww = 1. / _w, x = _x * ww, y = _y * ww;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            const double u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0;
            const double v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0;
            mapx[(size_t)i * w + j] = (float)u;
            mapy[(size_t)i * w + j] = (float)v;
        }
    }
}

static short sat_short(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

/* map conversion of cv::remap for CV_32FC1 maps and INTER_LINEAR: float product, round half to even */
void orbo_remap_prepare(const float *mapx, const float *mapy, int w, int h, int16_t *xy, uint16_t *frac)
{
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const int sx = orbo_cvround((double)(mapx[i] * 32.0f)), sy = orbo_cvround((double)(mapy[i] * 32.0f));
        xy[2 * i] = sat_short(sx >> 5);
        xy[2 * i + 1] = sat_short(sy >> 5);
        frac[i] = (uint16_t)((sy & 31) * 32 + (sx & 31));
    }
}

/* remapBilinear, 8UC1, BORDER_CONSTANT with value 0.  Weights a*b*32 with a, b in 1/32 units: exactly the
 * table OpenCV builds (saturate_cast<short>(v * 32768)); its only saturating entry (fx = fy = 0 -> 32767 plus a
 * one-unit correction on the opposite tap) yields the same 8-bit result as the exact weight. */
void orbo_remap_linear_u8(const uint8_t *src, int sw, int sh, int sstride, const int16_t *xy, const uint16_t *frac,
                          int dw, int dh, uint8_t *dst, int dstride)
{
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            const size_t i = (size_t)y * dw + x;
            const int sx = xy[2 * i], sy = xy[2 * i + 1];
            const int fx = frac[i] & 31, fy = frac[i] >> 5;
            const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32,
                      w11 = fx * fy * 32;
            int p00 = 0, p01 = 0, p10 = 0, p11 = 0;
            if (sx >= 0 && sx < sw && sy >= 0 && sy < sh) p00 = src[(size_t)sy * sstride + sx];
            if (sx + 1 >= 0 && sx + 1 < sw && sy >= 0 && sy < sh) p01 = src[(size_t)sy * sstride + sx + 1];
            if (sx >= 0 && sx < sw && sy + 1 >= 0 && sy + 1 < sh) p10 = src[(size_t)(sy + 1) * sstride + sx];
            if (sx + 1 >= 0 && sx + 1 < sw && sy + 1 >= 0 && sy + 1 < sh) p11 = src[(size_t)(sy + 1) * sstride + sx + 1];
            dst[(size_t)y * dstride + x] = (uint8_t)((p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15);
        }
}
