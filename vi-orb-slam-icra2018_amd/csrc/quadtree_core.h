// quadtree_core.h -- data-parallel formulation of ORBextractor::DistributeOctTree
// (reference: src/ORBextractor.cc:541-765, DivideNode :483-539).
//
// The reference walks a std::list and copies keypoint vectors around.  Here nothing moves:
//   * every candidate keeps its position in the canonical candidate order (cells row-major,
//     raster inside a cell) and only carries the list position of the node that owns it;
//   * a "pass" (one trip of the reference's while loop, :596-741) is: count the points of the
//     four children of every node that is being split (one atomic per point), then rebuild the
//     node list with prefix sums so that it has exactly the order std::list would have after
//     the push_front/erase sequence, then relabel the points -- and, in the same trip over the
//     points (r06), count them into the children of their NEW node for the next pass;
//   * the final phase (:675-739: expand the largest nodes first, stop as soon as the list has N
//     nodes) splits all candidates speculatively, ranks them by (size desc, list position asc)
//     and commits the prefix of that order up to the break point.  List position ascending is
//     the canonical replacement for the reference's heap-address tie-break (later-created node
//     = greater; later-created children sit nearer the list front) -- DESIGN.md "quadtree".
//   * the per-node winner (:746-762, max response, first in vKeys order wins ties) is one
//     atomicMax on (score, -candidate index), valid because DivideNode keeps vKeys in candidate
//     order.
//
// The template parameter X supplies the execution model: QtBlock (one HIP workgroup, LDS
// atomics, wave scans) in k_quadtree.hip, or QtSerial (one host thread) for the CPU test that
// checks this formulation against the list-based oracle.  Every loop between two x.sync()
// calls is a data-parallel loop, so the serial run is one legal schedule of the parallel one.
#ifndef ORBHIP_QUADTREE_CORE_H
#define ORBHIP_QUADTREE_CORE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define QT_HD __host__ __device__ __forceinline__
#define QT_UNROLL _Pragma("unroll")
#else
#define QT_HD inline
#define QT_UNROLL
#endif
#ifndef QT_K
#define QT_K 4   // points per thread and trip of the loops over the points
#endif

// Candidate packing: x[0..11] | y[12..23] | score[24..31]; x,y relative to (16,16) of the level.
#define QT_PACK(x, y, s) ((uint32_t)(x) | ((uint32_t)(y) << 12) | ((uint32_t)(s) << 24))
#define QT_X(p) ((int)((p)&0xFFFu))
#define QT_Y(p) ((int)(((p) >> 12) & 0xFFFu))
#define QT_S(p) ((int)((p) >> 24))

struct QtParams {
    int N;        // mnFeaturesPerLevel[level]
    int nIni;     // :545
    float hX;     // :547
    int regw;     // maxX - minX
    int regh;     // maxY - minY
    int maxNodes; // capacity of the node arrays (>= max(N + 3, 4 * nIni))
    int maxIter;  // 64 (the reference's loop ends long before); fewer only in the timing ablation of k_quadtree
};

// Working set of one (frame, level) problem.  All arrays have maxNodes entries unless noted.
struct QtShared {
    // current node list, in std::list order
    short *ulx, *uly, *brx, *bry;
    int *cnt;
    // next node list (double buffer)
    short *n_ulx, *n_uly, *n_brx, *n_bry;
    int *n_cnt;
    int *ccnt;     // [4 * maxNodes] child point counts of the pass
    int *ccnt2;    // [4 * maxNodes] ... of the next pass (filled while the points are relabelled; the two swap)
    int *npos;     // new list position of a surviving node (or -1 when it is erased)
    int *cpos;     // [4 * maxNodes] new list position of child q of node p (or -1)
    int *scan;     // scan workspace
    int *rank;     // careful phase: rank of candidate node p, -1 if not a candidate
    int *order;    // careful phase: node at rank r
    unsigned *best; // winner key per node
    int *scal;     // [8] block-wide scalars
};

enum { QT_S_SIZE = 0, QT_S_NEXP = 1, QT_S_TOTAL = 2, QT_S_JSTAR = 3, QT_S_NEWCH = 4 };

QT_HD int qt_ceil_half(int d)
{
    // ceil(static_cast<float>(d)/2), :485-486
    return (d + 1) >> 1; // d >= 0 always (node extents never go negative)
}

// Bytes of QtShared storage needed for maxNodes nodes (shorts first, all 4-byte aligned).
QT_HD size_t qt_shared_bytes(int maxNodes)
{
    size_t m = (size_t)((maxNodes + 1) & ~1);
    return m * 2 * 8      // 8 short arrays
           + m * 4 * 2    // cnt, n_cnt
           + m * 4 * 4 * 2 // ccnt, ccnt2
           + m * 4        // npos
           + m * 4 * 4    // cpos
           + (m + 8) * 4  // scan
           + m * 4 * 2    // rank, order
           + m * 4        // best
           + 8 * 4;       // scal
}

QT_HD void qt_carve(QtShared &sh, void *base, int maxNodes)
{
    size_t m = (size_t)((maxNodes + 1) & ~1);
    char *p = (char *)base;
    sh.ulx = (short *)p; p += m * 2;
    sh.uly = (short *)p; p += m * 2;
    sh.brx = (short *)p; p += m * 2;
    sh.bry = (short *)p; p += m * 2;
    sh.n_ulx = (short *)p; p += m * 2;
    sh.n_uly = (short *)p; p += m * 2;
    sh.n_brx = (short *)p; p += m * 2;
    sh.n_bry = (short *)p; p += m * 2;
    sh.cnt = (int *)p; p += m * 4;
    sh.n_cnt = (int *)p; p += m * 4;
    sh.ccnt = (int *)p; p += m * 16;
    sh.ccnt2 = (int *)p; p += m * 16;
    sh.npos = (int *)p; p += m * 4;
    sh.cpos = (int *)p; p += m * 16;
    sh.scan = (int *)p; p += (m + 8) * 4;
    sh.rank = (int *)p; p += m * 4;
    sh.order = (int *)p; p += m * 4;
    sh.best = (unsigned *)p; p += m * 4;
    sh.scal = (int *)p;
}

// Quadrant of a point inside node (ulx,uly,brx,bry): 0=n1(UL) 1=n2(UR) 2=n3(BL) 3=n4(BR), :511-528
QT_HD int qt_quadrant(int px, int py, int ulx, int uly, int brx, int bry)
{
    const int midx = ulx + qt_ceil_half(brx - ulx);
    const int midy = uly + qt_ceil_half(bry - uly);
    return (px < midx ? 0 : 1) + (py < midy ? 0 : 2);
}

QT_HD void qt_child_box(int q, int ulx, int uly, int brx, int bry, short &cx0, short &cy0, short &cx1,
                        short &cy1)
{
    const int midx = ulx + qt_ceil_half(brx - ulx);
    const int midy = uly + qt_ceil_half(bry - uly);
    cx0 = (short)((q & 1) ? midx : ulx);
    cx1 = (short)((q & 1) ? brx : midx);
    cy0 = (short)((q & 2) ? midy : uly);
    cy1 = (short)((q & 2) ? bry : midy);
}

// Runs the whole distribution for one (frame, level).
//   pts[n]   packed candidates in canonical order (read only)
//   pnode[n] scratch: list position of the owning node (bits 0..29) and quadrant (bits 30..31)
//   out[]    packed winners in list order (capacity maxNodes)
// Returns the number of winners (every thread gets the same value).
//   firstCounted  (one root only) the caller has labelled the points for the first pass while it laid them out: pnode[i] =
//            quadrant in the root << 30 (the root: (0, 0) .. ((int)hX, regh)), sh.ccnt[0..3] = the four counts
template <class X>
QT_HD int qt_distribute(X &x, const QtParams P, const int n, const uint32_t *__restrict__ pts,
                        uint32_t *__restrict__ pnode, QtShared &sh, uint32_t *__restrict__ out, const bool firstCounted = false)
{
    const int M = P.maxNodes;
    const bool oneRoot = P.nIni == 1;
    int S;
    // ---- roots (:549-587): nIni nodes side by side, points by x/hX, empty roots erased ----
    if (oneRoot) {
        // one root (every image less than 1.5 times as wide as high): all points are its points -- no trip over the points here,
        // the first pass labels them
        if (x.tid() == 0) {
            sh.ulx[0] = 0;
            sh.uly[0] = 0;
            sh.brx[0] = (short)(int)(P.hX * 1.f);
            sh.bry[0] = (short)P.regh;
            sh.cnt[0] = n;
        }
        if (!firstCounted)
            for (int i = x.tid(); i < 4; i += x.nth()) sh.ccnt[i] = 0;
        S = n > 0 ? 1 : 0;
        x.sync();
    } else {
        for (int i = x.tid(); i < P.nIni; i += x.nth()) sh.ccnt[i] = 0;
        x.sync();
        for (int i = x.tid(); i < n; i += x.nth()) {
            int r = (int)((float)QT_X(pts[i]) / P.hX);
            r = r < 0 ? 0 : (r >= P.nIni ? P.nIni - 1 : r);
            pnode[i] = (uint32_t)r;
            x.atomic_add(&sh.ccnt[r], 1);
        }
        x.sync();
        for (int i = x.tid(); i < P.nIni; i += x.nth()) sh.scan[i] = sh.ccnt[i] > 0 ? 1 : 0;
        x.sync();
        S = x.scan_exclusive(sh.scan, P.nIni); // sh.scan[i] = new position of root i
        for (int i = x.tid(); i < P.nIni; i += x.nth()) {
            if (sh.ccnt[i] > 0) {
                const int s = sh.scan[i];
                sh.ulx[s] = (short)(int)(P.hX * (float)i);
                sh.uly[s] = 0;
                sh.brx[s] = (short)(int)(P.hX * (float)(i + 1));
                sh.bry[s] = (short)P.regh;
                sh.cnt[s] = sh.ccnt[i];
                sh.npos[i] = s;
            } else
                sh.npos[i] = -1;
        }
        x.sync();
        for (int i = x.tid(); i < n; i += x.nth()) pnode[i] = (uint32_t)sh.npos[pnode[i]];
        for (int i = x.tid(); i < 4 * S; i += x.nth()) sh.ccnt[i] = 0;   // (the root counts were consumed before the last sync)
        x.sync();
    }

    bool careful = false; // inside the final phase (:675-739)
    bool winnersDone = false;
    for (int iter = 0; iter < P.maxIter; ++iter) {
        // ---- which nodes are split candidates: every node holding more than one point ----
        // (after a full pass, and after a completed careful pass, all such nodes are children
        //  created by the previous pass, i.e. exactly vSizeAndPointerToNode)
        // ---- children point counts (speculative for every candidate) ----
        // First pass only (ccnt[0 .. 4 S) was cleared before the loop): every later pass finds them counted by the
        // relabelling trip of the pass before it.
        // The trips over the points take QT_K points per thread at a time, every stage for all of them before the next one:
        // a trip is a chain of dependent round trips (labels and candidates from memory, then the node's table entries from
        // LDS, then the atomic), and neither the compiler (the atomics and the label stores keep it from moving loads) nor the
        // handful of waves a CU holds hides them.
        if (iter == 0 && !(firstCounted && oneRoot)) {
            for (int i0 = x.tid(); i0 < n; i0 += QT_K * x.nth()) {
                uint32_t pn[QT_K], pk[QT_K];
                int c[QT_K], bx0[QT_K], by0[QT_K], bx1[QT_K], by1[QT_K];
                QT_UNROLL
                for (int k = 0; k < QT_K; ++k) {
                    const int i = i0 + k * x.nth();
                    pn[k] = (!oneRoot && i < n) ? (pnode[i] & 0x3FFFFFFFu) : 0u;
                    pk[k] = i < n ? pts[i] : 0u;
                }
                QT_UNROLL
                for (int k = 0; k < QT_K; ++k) {
                    c[k] = sh.cnt[pn[k]];
                    bx0[k] = sh.ulx[pn[k]];
                    by0[k] = sh.uly[pn[k]];
                    bx1[k] = sh.brx[pn[k]];
                    by1[k] = sh.bry[pn[k]];
                }
                QT_UNROLL
                for (int k = 0; k < QT_K; ++k) {
                    const int i = i0 + k * x.nth();
                    if (i < n) {
                        uint32_t lab = pn[k];
                        if (c[k] > 1) {
                            const int q = qt_quadrant(QT_X(pk[k]), QT_Y(pk[k]), bx0[k], by0[k], bx1[k], by1[k]);
                            lab |= (uint32_t)q << 30;
                            x.atomic_add(&sh.ccnt[4 * pn[k] + q], 1);
                        }
                        pnode[i] = lab;
                    }
                }
            }
            x.sync();
        }

        int jstar = -1; // careful phase: last rank that is processed
        int C = 0;      // number of candidates
        int newS = 0;   // size of the next list
        if (!careful) {
            // full pass: every candidate is split, in list order
            // scan value packs (#non-empty children << 16) | (node survives unsplit)
            for (int p = x.tid(); p < S; p += x.nth()) {
                int v;
                if (sh.cnt[p] > 1) {
                    int nz = (sh.ccnt[4 * p] > 0) + (sh.ccnt[4 * p + 1] > 0) + (sh.ccnt[4 * p + 2] > 0) +
                             (sh.ccnt[4 * p + 3] > 0);
                    v = nz << 16;
                } else
                    v = 1;
                sh.scan[p] = v;
            }
            x.sync();
            const int tot = x.scan_exclusive(sh.scan, S);
            const int totalCh = tot >> 16;
            newS = totalCh + (tot & 0xFFFF);   // children created + nodes that stay
            for (int p = x.tid(); p < S; p += x.nth()) {
                const int pre = sh.scan[p];
                if (sh.cnt[p] > 1) {
                    int nz = (sh.ccnt[4 * p] > 0) + (sh.ccnt[4 * p + 1] > 0) + (sh.ccnt[4 * p + 2] > 0) +
                             (sh.ccnt[4 * p + 3] > 0);
                    // children of later-split nodes sit in front: suffix sum
                    int base = totalCh - (pre >> 16) - nz;
                    // within the node: n4 first ... n1 last
                    int k = 0;
                    for (int q = 3; q >= 0; --q) {
                        if (sh.ccnt[4 * p + q] > 0)
                            sh.cpos[4 * p + q] = base + k++;
                        else
                            sh.cpos[4 * p + q] = -1;
                    }
                    sh.npos[p] = -1;
                } else {
                    sh.npos[p] = totalCh + (pre & 0xFFFF);
                }
            }
            x.sync();
        } else {
            // careful pass: rank candidates by (cnt desc, list position asc)
            for (int p = x.tid(); p < S; p += x.nth()) sh.scan[p] = sh.cnt[p] > 1 ? 1 : 0;
            x.sync();
            C = x.scan_exclusive(sh.scan, S); // compact candidate index
            // sort keys of the candidates in list order (best[] is free until the winners): candidate d goes before candidate c
            // <=> more points, or as many and earlier in the list <=> key_d > key_c with key = count << 12 | (4095 - index).
            // (lists of more than 4096 nodes or levels of 2^19 candidates keep the two-part comparison)
            unsigned *ccand = sh.best;
            const bool keyed = S <= 4096 && n < (1 << 19);
            for (int p = x.tid(); p < S; p += x.nth())
                if (sh.cnt[p] > 1) {
                    const int ci = sh.scan[p];
                    sh.order[ci] = p; // temporarily: candidates in list order
                    ccand[ci] = keyed ? ((unsigned)sh.cnt[p] << 12) | (unsigned)(4095 - ci) : (unsigned)sh.cnt[p];
                }
            x.sync();
            // rank of candidate c = number of candidates that go before it: an all-pairs count, C * C comparisons.  One thread
            // per candidate made this a serial loop of C steps whatever the workgroup size (a third of the single-frame
            // quadtree time at C ~ 150); the comparisons of a candidate are now dealt to SPLIT threads that add up their parts.
            {
                // SPLIT = the largest power of two (<= 64) with C * SPLIT <= threads: the index arithmetic is shifts (three integer
                // divisions per thread were most of this step -- one workgroup runs ~5 cycles per instruction and wave), and the
                // SPLIT partial counts of a candidate sit in adjacent lanes of one wave: they are summed there (group_sum) and
                // one lane stores the rank -- a thousand atomics on C addresses took longer than the comparisons
                int sl = 0;
                while (C > 0 && sl < 6 && ((C << (sl + 1)) <= x.nth())) sl++;
                const int SPLIT = 1 << sl;
                for (int i0 = 0; i0 < C * SPLIT; i0 += x.nth()) {
                    const int idx = i0 + x.tid();
                    const bool live = idx < C * SPLIT;
                    const int c = live ? idx >> sl : 0, part = idx & (SPLIT - 1);
                    const int d0 = live ? (part * C) >> sl : 0, d1 = live ? ((part + 1) * C) >> sl : 0;
                    const unsigned mykey = ccand[c];
                    int r = 0;
                    // (one read and one comparison per pair from the compact key array, four in flight; through order[] and
                    // cnt[] it was a chain of two dependent reads and ten instructions)
                    if (keyed) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 4
#endif
                        for (int d = d0; d < d1; ++d) r += ccand[d] > mykey;
                    } else {
                        for (int d = d0; d < d1; ++d) {
                            const unsigned pc = ccand[d];
                            r += (pc > mykey) || (pc == mykey && d < c);
                        }
                    }
                    r = x.group_sum(r, sl);   // called by every thread of the workgroup
                    if (live && part == 0) sh.rank[sh.order[c]] = r;
                }
            }
            x.sync();
            // (the rank loops above read order[] of every candidate: rewrite it only after the sync)
            for (int p = x.tid(); p < S; p += x.nth())
                if (sh.cnt[p] > 1) sh.order[sh.rank[p]] = p; // node at rank r
            x.sync();
            // inclusive running size after processing rank r: S + sum_{i<=r} (nz_i - 1)
            for (int r = x.tid(); r < C; r += x.nth()) {
                const int p = sh.order[r];
                int nz = (sh.ccnt[4 * p] > 0) + (sh.ccnt[4 * p + 1] > 0) + (sh.ccnt[4 * p + 2] > 0) +
                         (sh.ccnt[4 * p + 3] > 0);
                sh.scan[r] = nz;
            }
            if (x.tid() == 0) sh.scal[QT_S_JSTAR] = C - 1;
            x.sync();
            x.scan_exclusive(sh.scan, C); // scan[r] = sum_{i<r} nz_i
            // first rank whose inclusive size reaches N (:732-733)
            for (int r = x.tid(); r < C; r += x.nth()) {
                const int p = sh.order[r];
                int nz = (sh.ccnt[4 * p] > 0) + (sh.ccnt[4 * p + 1] > 0) + (sh.ccnt[4 * p + 2] > 0) +
                         (sh.ccnt[4 * p + 3] > 0);
                // the running size never shrinks (nz >= 1), so exactly one rank crosses N first: it alone writes (an atomic
                // minimum over all ranks beyond it was ~150 same-address atomics in the final pass)
                const int sizeAfter = S + (sh.scan[r] + nz) - (r + 1);
                const int sizeBefore = S + sh.scan[r] - r;
                if (sizeAfter >= P.N && (r == 0 || sizeBefore < P.N)) sh.scal[QT_S_JSTAR] = r;
            }
            x.sync();
            jstar = sh.scal[QT_S_JSTAR];
            x.sync();
            // total children created = inclusive sum at jstar
            if (x.tid() == 0) {
                int tc = 0;
                if (C > 0) {
                    const int p = sh.order[jstar];
                    int nz = (sh.ccnt[4 * p] > 0) + (sh.ccnt[4 * p + 1] > 0) + (sh.ccnt[4 * p + 2] > 0) +
                             (sh.ccnt[4 * p + 3] > 0);
                    tc = sh.scan[jstar] + nz;
                }
                sh.scal[QT_S_NEWCH] = tc;
            }
            x.sync();
            const int totalCh = sh.scal[QT_S_NEWCH];
            // children positions: later-processed nodes in front
            for (int r = x.tid(); r < C; r += x.nth()) {
                const int p = sh.order[r];
                if (r <= jstar) {
                    int nz = (sh.ccnt[4 * p] > 0) + (sh.ccnt[4 * p + 1] > 0) + (sh.ccnt[4 * p + 2] > 0) +
                             (sh.ccnt[4 * p + 3] > 0);
                    int base = totalCh - sh.scan[r] - nz;
                    int k = 0;
                    for (int q = 3; q >= 0; --q) {
                        if (sh.ccnt[4 * p + q] > 0)
                            sh.cpos[4 * p + q] = base + k++;
                        else
                            sh.cpos[4 * p + q] = -1;
                    }
                }
            }
            x.sync();
            // surviving old nodes keep their relative order behind the new children
            for (int p = x.tid(); p < S; p += x.nth()) {
                const bool processed = sh.cnt[p] > 1 && sh.rank[p] <= jstar;
                sh.npos[p] = processed ? -1 : 0; // mark
            }
            x.sync();
            for (int p = x.tid(); p < S; p += x.nth()) sh.scan[p] = sh.npos[p] < 0 ? 0 : 1;
            x.sync();
            const int stay = x.scan_exclusive(sh.scan, S);
            newS = totalCh + stay;
            for (int p = x.tid(); p < S; p += x.nth())
                if (sh.npos[p] >= 0) sh.npos[p] = totalCh + sh.scan[p];
            x.sync();
        }

        // The loop ends behind this pass when the list is long enough or did not grow (the loop control below): its trip over the
        // points then picks the winners of the new nodes instead of preparing a pass that never runs.
        const bool last = newS >= P.N || newS == S;
        // ---- build the next list ----
        // (the size of the next list is known from the scans above; the number of expandable children is summed per wave
        // before it touches the shared counter -- one atomic per node on two counters was the longest step of a pass)
        if (x.tid() == 0) sh.scal[QT_S_NEXP] = 0;
        x.sync();
        for (int p0 = 0; p0 < S; p0 += x.nth()) {
            const int p = p0 + x.tid();
            int nexp = 0;
            if (p >= S) {
            } else if (sh.npos[p] >= 0) {
                const int s = sh.npos[p];
                sh.n_ulx[s] = sh.ulx[p];
                sh.n_uly[s] = sh.uly[p];
                sh.n_brx[s] = sh.brx[p];
                sh.n_bry[s] = sh.bry[p];
                sh.n_cnt[s] = sh.cnt[p];
            } else {
                for (int q = 0; q < 4; ++q) {
                    const int s = sh.cpos[4 * p + q];
                    if (s < 0) continue;
                    qt_child_box(q, sh.ulx[p], sh.uly[p], sh.brx[p], sh.bry[p], sh.n_ulx[s], sh.n_uly[s],
                                 sh.n_brx[s], sh.n_bry[s]);
                    sh.n_cnt[s] = sh.ccnt[4 * p + q];
                    nexp += sh.ccnt[4 * p + q] > 1;
                }
            }
            x.reduce_add(&sh.scal[QT_S_NEXP], nexp);   // called by every thread of the workgroup
        }
        if (last) {
            for (int i = x.tid(); i < newS; i += x.nth()) sh.best[i] = 0u;   // (the careful pass's keys in best[] are spent)
        } else {
            for (int i = x.tid(); i < 4 * newS; i += x.nth()) sh.ccnt2[i] = 0;   // the next pass's child counts
        }
        x.sync();
        // ---- relabel the points, and count them into the children of their new node (the next pass's counts; one pass too
        // many at the end, whose quadrant bits the winners mask off) ----
        for (int i0 = x.tid(); i0 < n; i0 += QT_K * x.nth()) {
            uint32_t v[QT_K], pk[QT_K];
            int np[QT_K], cp[QT_K], c[QT_K], bx0[QT_K], by0[QT_K], bx1[QT_K], by1[QT_K];
            QT_UNROLL
            for (int k = 0; k < QT_K; ++k) {
                const int i = i0 + k * x.nth();
                v[k] = i < n ? pnode[i] : 0u;
                pk[k] = i < n ? pts[i] : 0u;
            }
            QT_UNROLL
            for (int k = 0; k < QT_K; ++k) {
                const uint32_t pn = v[k] & 0x3FFFFFFFu;
                np[k] = sh.npos[pn];
                cp[k] = sh.cpos[4 * pn + (v[k] >> 30)];   // (stale where the node was not split: not taken then)
            }
            if (last) {
                // winners (:743-764): max response per node, the first in candidate order among equals
                QT_UNROLL
                for (int k = 0; k < QT_K; ++k) {
                    const int i = i0 + k * x.nth();
                    if (i < n)
                        x.atomic_max(&sh.best[np[k] >= 0 ? np[k] : cp[k]], ((uint32_t)QT_S(pk[k]) << 24) | (0xFFFFFFu - (uint32_t)i));
                }
                continue;
            }
            QT_UNROLL
            for (int k = 0; k < QT_K; ++k) {
                const int i = i0 + k * x.nth();
                const int s = i < n ? (np[k] >= 0 ? np[k] : cp[k]) : 0;
                np[k] = s;
                c[k] = sh.n_cnt[s];
                bx0[k] = sh.n_ulx[s];
                by0[k] = sh.n_uly[s];
                bx1[k] = sh.n_brx[s];
                by1[k] = sh.n_bry[s];
            }
            QT_UNROLL
            for (int k = 0; k < QT_K; ++k) {
                const int i = i0 + k * x.nth();
                if (i < n) {
                    uint32_t lab = (uint32_t)np[k];
                    if (c[k] > 1) {
                        const int q = qt_quadrant(QT_X(pk[k]), QT_Y(pk[k]), bx0[k], by0[k], bx1[k], by1[k]);
                        lab |= (uint32_t)q << 30;
                        x.atomic_add(&sh.ccnt2[4 * np[k] + q], 1);
                    }
                    pnode[i] = lab;
                }
            }
        }
        const int nToExpand = sh.scal[QT_S_NEXP];
        x.sync();
        // swap list buffers
        {
            short *t;
            int *ti;
            t = sh.ulx; sh.ulx = sh.n_ulx; sh.n_ulx = t;
            t = sh.uly; sh.uly = sh.n_uly; sh.n_uly = t;
            t = sh.brx; sh.brx = sh.n_brx; sh.n_brx = t;
            t = sh.bry; sh.bry = sh.n_bry; sh.n_bry = t;
            ti = sh.cnt; sh.cnt = sh.n_cnt; sh.n_cnt = ti;
            ti = sh.ccnt; sh.ccnt = sh.ccnt2; sh.ccnt2 = ti;
        }
        const int prevS = S;
        S = newS;
        winnersDone = last;
        // ---- loop control (:665-673, :736-737) ----
        if (S >= P.N || S == prevS) break;
        if (!careful && S + nToExpand * 3 > P.N) careful = true;
        (void)M;
    }

    // ---- winners (:743-764) ---- (picked by the last pass; this trip only when the loop ran out of passes: the timing ablation)
    if (!winnersDone) {
        for (int s = x.tid(); s < S; s += x.nth()) sh.best[s] = 0u;
        x.sync();
        for (int i = x.tid(); i < n; i += x.nth())
            x.atomic_max(&sh.best[pnode[i] & 0x3FFFFFFFu], ((uint32_t)QT_S(pts[i]) << 24) | (0xFFFFFFu - (uint32_t)i));
        x.sync();
    }
    for (int s = x.tid(); s < S; s += x.nth()) out[s] = pts[0xFFFFFFu - (sh.best[s] & 0xFFFFFFu)];
    x.sync();
    return S;
}

// Serial execution model for the host-side check of the formulation.
struct QtSerial {
    QT_HD int tid() const { return 0; }
    QT_HD int nth() const { return 1; }
    QT_HD void sync() const {}
    QT_HD int atomic_add(int *p, int v) const
    {
        int o = *p;
        *p = o + v;
        return o;
    }
    QT_HD void atomic_min(int *p, int v) const
    {
        if (v < *p) *p = v;
    }
    QT_HD void reduce_add(int *p, int v) const { *p += v; }
    QT_HD int group_sum(int v, int) const { return v; }   // groups of 2^sl adjacent threads: one thread, groups of one
    QT_HD void atomic_max(unsigned *p, unsigned v) const
    {
        if (v > *p) *p = v;
    }
    QT_HD int scan_exclusive(int *a, int n) const
    {
        int s = 0;
        for (int i = 0; i < n; ++i) {
            int v = a[i];
            a[i] = s;
            s += v;
        }
        return s;
    }
};

#endif // ORBHIP_QUADTREE_CORE_H
