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
    /* tiny image: must be rejected, not crash */
    if (orbo_extract(e, img, 60, 60, W, k, d, cap) >= 0) return 5;
    orbo_destroy(e);
    free(img); free(k); free(d); free(bi); free(bd); free(sd); free(idx); free(m12); free(m21); free(valid); free(ang);
    printf("oracle sanitizer run ok: %d keypoints, %d matches\n", n, nm);
    return 0;
}
