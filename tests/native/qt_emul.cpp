// Host emulation of the device quadtree formulation (csrc/quadtree_core.h) with the serial
// execution model, exported for the CPU test that compares it with the list-based oracle.
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "quadtree_core.h"

extern "C" int qt_emul_distribute(const uint32_t *pts, int n, int regw, int regh, int N, uint32_t *out,
                                  int cap)
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
    int S = qt_distribute(x, P, n, pts, pnode, sh, tmp);
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
