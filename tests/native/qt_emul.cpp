// Host emulation of the device quadtree formulation (csrc/quadtree_core.h) with the serial
// execution model, exported for the CPU test that compares it with the list-based oracle.
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "quadtree_core.h"

// first != 0: the caller's side of qt_distribute's firstCounted contract as k_quadtree's batch gather keeps it (one root only):
// labels = quadrant in the root << 30, ccnt[0..3] = the four counts, before the call
static int emul(const uint32_t *pts, int n, int regw, int regh, int N, uint32_t *out, int cap, int first)
{
    QtParams P;
    P.N = N;
    P.nIni = (int)roundf((float)regw / (float)regh);
    if (P.nIni < 1) return -2;
    P.hX = (float)regw / (float)P.nIni;
    P.regw = regw;
    P.regh = regh;
    int m = N + 4;
    if (4 * P.nIni + 4 > m) m = 4 * P.nIni + 4;
    P.maxNodes = m;
    P.maxIter = 64;
    void *mem = calloc(1, qt_shared_bytes(m));
    uint32_t *pnode = (uint32_t *)calloc((size_t)n + 1, 4);
    uint32_t *tmp = (uint32_t *)calloc((size_t)m, 4);
    QtShared sh;
    qt_carve(sh, mem, m);
    QtSerial x;
    const bool firstCounted = first && P.nIni == 1;
    if (firstCounted) {
        const int midx = qt_ceil_half((int)(short)(int)(P.hX * 1.f)), midy = qt_ceil_half((int)(short)regh);
        for (int q = 0; q < 4; q++) sh.ccnt[q] = 0;
        for (int i = 0; i < n; i++) {
            const int q = n > 1 ? (QT_X(pts[i]) < midx ? 0 : 1) + (QT_Y(pts[i]) < midy ? 0 : 2) : 0;
            pnode[i] = (uint32_t)q << 30;
            if (n > 1) sh.ccnt[q]++;
        }
    }
    int S = qt_distribute(x, P, n, pts, pnode, sh, tmp, firstCounted);
    int rc = S;
    if (S > cap)
        rc = -3;
    else
        memcpy(out, tmp, (size_t)S * 4);
    free(mem);
    free(pnode);
    free(tmp);
    return rc;
}

extern "C" int qt_emul_distribute(const uint32_t *pts, int n, int regw, int regh, int N, uint32_t *out, int cap)
{
    return emul(pts, n, regw, regh, N, out, cap, 0);
}

extern "C" int qt_emul_distribute_first_counted(const uint32_t *pts, int n, int regw, int regh, int N, uint32_t *out, int cap)
{
    return emul(pts, n, regw, regh, N, out, cap, 1);
}
