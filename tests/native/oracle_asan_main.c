/* Runs the oracle's main entry points on procedural inputs; built with -fsanitize=address,undefined by
 * tests/test_oracle_sanitizers.py (the reference has no sanitizer coverage). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orb_oracle.h"

static unsigned lcg(unsigned *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(void)
{
    const int W = 400, H = 300;
    unsigned seed = 12345;
    unsigned char *img = (unsigned char *)malloc((size_t)W * H);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int v = 110 + (((x / 23) + (y / 17)) & 1) * 70 + (int)(lcg(&seed) % 9);
            if (((x / 61) + (y / 47)) % 3 == 0) v = 100 + (int)(lcg(&seed) % 12);
            img[y * W + x] = (unsigned char)v;
        }
    orbo_extractor *e = orbo_create(500, 1.2f, 6, 20, 7);
    const int cap = 2200;
    orbo_keypoint *k = (orbo_keypoint *)malloc(sizeof(orbo_keypoint) * cap);
    unsigned char *d = (unsigned char *)malloc((size_t)cap * 32);
    int n = orbo_extract(e, img, W, H, W, k, d, cap);
    if (n <= 0) { fprintf(stderr, "extract failed %d\n", n); return 1; }
    int n2 = orbo_extract(e, img, W, H, W, k, d, cap);   /* second call reuses / frees the stage buffers */
    if (n2 != n) return 2;
    int32_t *bi = (int32_t *)malloc(4 * (size_t)n), *bd = (int32_t *)malloc(4 * (size_t)n), *sd = (int32_t *)malloc(4 * (size_t)n);
    orbo_knn2(d, n, d, n, bi, bd, sd);
    for (int i = 0; i < n; i++)
        if (bd[i] != 0) return 3;
    /* SearchByBoW with two nodes */
    int32_t node[2] = {3, 9}, off[3] = {0, n / 2, n};
    int32_t *idx = (int32_t *)malloc(4 * (size_t)n), *m12 = (int32_t *)malloc(4 * (size_t)n), *m21 = (int32_t *)malloc(4 * (size_t)n);
    unsigned char *valid = (unsigned char *)malloc((size_t)n);
    float *ang = (float *)malloc(4 * (size_t)n);
    for (int i = 0; i < n; i++) { idx[i] = i; valid[i] = 1; ang[i] = k[i].angle; }
    int nm = orbo_search_by_bow(d, n, valid, ang, node, off, idx, 2, d, n, NULL, ang, node, off, idx, 2, 50, 0, 0.9f, 1, m12, m21);
    if (nm <= 0) return 4;
    /* stereo on the extractor's own pyramid (left == right), grid + guided search,
     * undistortion and rectification */
    {
        const uint8_t *pyr[6];
        int lw[6], lh[6];
        const orbo_params *P = orbo_get_params(e);
        for (int l = 0; l < 6; l++) {
            int st;
            pyr[l] = orbo_pyramid_level(e, l, &lw[l], &lh[l], &st);
            if (st != lw[l]) return 6;
        }
        float *ur = (float *)malloc(4 * (size_t)n), *dz = (float *)malloc(4 * (size_t)n);
        int ns = orbo_stereo_matches(k, d, n, k, d, n, pyr, pyr, lw, lh, P->mvScaleFactor, P->mvInvScaleFactor, 0.1f, 30.f, ur, dz);
        if (ns <= 0) return 7;
        const float invW = 64.f / (float)W, invH = 48.f / (float)H;
        int32_t *coff = (int32_t *)malloc(4 * (64 * 48 + 1)), *cidx = (int32_t *)malloc(4 * (size_t)n), *pm = (int32_t *)malloc(4 * (size_t)n);
        orbo_grid_build(k, n, 0.f, 0.f, invW, invH, coff, cidx);
        orbo_proj_query *q = (orbo_proj_query *)calloc((size_t)n, sizeof(orbo_proj_query));
        for (int i = 0; i < n; i++) {
            q[i].u = k[i].x + 1.5f; q[i].v = k[i].y - 1.f; q[i].radius = 400.f * (float)(i % 3 == 0) + 12.f;
            q[i].min_level = k[i].octave - 1; q[i].max_level = k[i].octave + 1; q[i].angle = k[i].angle;
            q[i].flags = ORBO_Q_ACTIVE | (i % 4 ? ORBO_Q_OBSERVED : 0);
        }
        q[0].u = -1e6f; q[1].v = 1e6f;
        int np1 = orbo_search_by_projection(k, d, n, ur, NULL, 0.f, 0.f, invW, invH, q, d, n, 0, 0.9f, 1, 100, pm);
        int np2 = orbo_search_by_projection(k, d, n, NULL, valid, 0.f, 0.f, invW, invH, q, d, n, 1, 0.8f, 0, 100, pm);
        if (np1 <= 0 || np2 != 0) return 8;                 /* every feature occupied in the second run */
        {
            /* SearchForInitialization (huge and tiny windows, points outside the grid) and SearchForTriangulation */
            float *prev = (float *)malloc(8 * (size_t)n);
            for (int i = 0; i < n; i++) { prev[2 * i] = k[i].x + 2.f; prev[2 * i + 1] = k[i].y - 1.f; }
            prev[0] = -1e6f; prev[3] = 1e6f;
            int ni1 = orbo_search_for_initialization(k, d, n, k, d, n, 0.f, 0.f, invW, invH, prev, 100, 0.9f, 1, 50, pm);
            int ni2 = orbo_search_for_initialization(k, d, n, k, d, n, 0.f, 0.f, invW, invH, prev, 0, 0.9f, 0, 50, pm);
            if (ni1 <= 0 || ni2 != 0) return 9;
            const float F12[9] = {0.f, 0.f, 0.f, 0.f, 0.f, -1.f, 0.f, 1.f, 0.f};
            uint8_t *skip = (uint8_t *)calloc((size_t)n, 1);
            for (int i = 0; i < n; i += 5) skip[i] = 1;
            int nt = orbo_search_for_triangulation(k, d, n, skip, NULL, node, off, idx, 2, k, d, n, skip, ur, node, off, idx, 2, F12,
                                                   -50.f, 40.f, P->mvScaleFactor, P->mvLevelSigma2, 0, 1, 50, pm);
            if (nt <= 0) return 10;
            free(skip);
            free(prev);
            /* Fuse / SearchBySim3 window search (gate off, mono gate, stereo gate) and ComputeDistinctiveDescriptors */
            int32_t *bi = (int32_t *)malloc(4 * (size_t)n), *bd = (int32_t *)malloc(4 * (size_t)n);
            float inv_s2[8];
            for (int l = 0; l < 8; l++) inv_s2[l] = 1.f / P->mvLevelSigma2[l < P->nlevels ? l : 0];
            for (int i = 0; i < n; i++) q[i].max_level = k[i].octave;
            orbo_window_best(k, d, n, NULL, NULL, 0.f, 0.f, invW, invH, q, d, n, bi, bd);
            int found = 0;
            for (int i = 0; i < n; i++) found += bi[i] >= 0;
            orbo_window_best(k, d, n, NULL, inv_s2, 0.f, 0.f, invW, invH, q, d, n, bi, bd);
            orbo_window_best(k, d, n, ur, inv_s2, 0.f, 0.f, invW, invH, q, d, n, bi, bd);
            orbo_window_best(k, d, 0, NULL, NULL, 0.f, 0.f, invW, invH, q, d, n, bi, bd);
            if (found <= 0 || bi[0] != -1 || bd[0] != 256) return 12;
            int32_t loff[5] = {0, 0, 1, 7, n};
            orbo_distinctive_descriptors(d, loff, 4, bi, bd);
            if (bi[0] != -1 || bi[1] != 0 || bd[1] != 0 || bi[2] < 0 || bi[2] >= 6 || bi[3] < 0 || bi[3] >= n - 7) return 13;
            free(bi);
            free(bd);
        }
        double K[9] = {300, 0, 200, 0, 300, 150, 0, 0, 1}, Dd[5] = {-0.2, 0.05, 1e-4, -1e-4, 0.0};
        double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Pm[9] = {280, 0, 205, 0, 280, 148, 0, 0, 1};
        float *mx = (float *)malloc(4 * (size_t)W * H), *my = (float *)malloc(4 * (size_t)W * H);
        orbo_init_undistort_rectify_map(K, Dd, 5, R, Pm, W, H, mx, my);
        mx[5] = -40000.f; my[6] = 1e9f; mx[7] = (float)W - 0.5f;  /* off-image and edge taps */
        int16_t *xy = (int16_t *)malloc(4 * (size_t)W * H);
        uint16_t *fr = (uint16_t *)malloc(2 * (size_t)W * H);
        unsigned char *rect = (unsigned char *)malloc((size_t)W * H);
        orbo_remap_prepare(mx, my, W, H, xy, fr);
        orbo_remap_linear_u8(img, W, H, W, xy, fr, W, H, rect, W);
        float Kf[9] = {300, 0, 200, 0, 300, 150, 0, 0, 1}, Df[4] = {-0.2f, 0.05f, 1e-4f, -1e-4f};
        float *pin = (float *)malloc(8 * (size_t)n), *pout = (float *)malloc(8 * (size_t)n);
        for (int i = 0; i < n; i++) { pin[2 * i] = k[i].x; pin[2 * i + 1] = k[i].y; }
        orbo_undistort_points(pin, n, Kf, Df, 4, Kf, pout);
        orbo_undistort_points(pin, n, Kf, NULL, 0, NULL, pout);
        free(ur); free(dz); free(coff); free(cidx); free(pm); free(q); free(mx); free(my); free(xy); free(fr); free(rect);
        free(pin); free(pout);
    }
    /* tiny image: must be rejected, not crash */
    if (orbo_extract(e, img, 60, 60, W, k, d, cap) >= 0) return 5;
    orbo_destroy(e);
    free(img); free(k); free(d); free(bi); free(bd); free(sd); free(idx); free(m12); free(m21); free(valid); free(ang);
    printf("oracle sanitizer run ok: %d keypoints, %d matches\n", n, nm);
    return 0;
}
