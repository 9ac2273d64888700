// host_restate.h -- ORBmatcher's searches restated on the host for the native drop-in tests (test infrastructure, like oracle/):
// plain loops over the Frame / KeyFrame / MapPoint accessors with KeyFrame::GetFeaturesInArea, Frame::GetFeaturesInArea and
// ORBmatcher::DescriptorDistance, each following the cited lines of the reference's src/ORBmatcher.cc.  The drop-in (HIP) result of
// the same call must be equal.  Used by test_fuse_dropin.cpp (two identical worlds) and test_threads_dropin.cpp (three threads).
// Compile with -ffp-contract=off: the projections are float arithmetic in the reference's order.
#ifndef ORBHIP_TESTS_HOST_RESTATE_H
#define ORBHIP_TESTS_HOST_RESTATE_H

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <utility>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"

using namespace ORB_SLAM2;
using std::vector;

static unsigned long long g_rng = 88172645463325252ull;
static double urand()   // xorshift64*, [0, 1)
{
    g_rng ^= g_rng >> 12; g_rng ^= g_rng << 25; g_rng ^= g_rng >> 27;
    return (double)((g_rng * 2685821657736338717ull) >> 11) / 9007199254740992.0;
}

static cv::Mat pose(float ax, float ay, float tx, float ty, float tz)
{
    cv::Mat T = cv::Mat::zeros(4, 4, CV_32F);
    const float cx = cosf(ax), sx = sinf(ax), cy = cosf(ay), sy = sinf(ay);
    const float R[9] = {cy, 0, sy, sx * sy, cx, -sx * cy, -cx * sy, sx, cx * cy};   // Rx(ax) * Ry(ay)
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) T.at<float>(r, c) = R[r * 3 + c];
    T.at<float>(0, 3) = tx; T.at<float>(1, 3) = ty; T.at<float>(2, 3) = tz; T.at<float>(3, 3) = 1.f;
    return T;
}

static void mul3(const cv::Mat &R, const float x[3], const float *t, float out[3], bool transpose = false, double alpha = 1.0)
{
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)(transpose ? R.at<float>(k, r) : R.at<float>(r, k)) * (double)x[k];
        out[r] = (float)(alpha * s + (t ? (double)t[r] : 0.0));
    }
}

// ---------------- the routines restated on the host ----------------
static bool window(KeyFrame *pKF, MapPoint *pMP, const float p3Dc[3], const float *PO, float dist3D, float th, float &u, float &v,
                   float &ur, int &level, float &radius)
{
    if (p3Dc[2] < 0.0f) return false;
    const float invz = 1.0 / p3Dc[2];
    const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
    u = pKF->fx * x + pKF->cx;
    v = pKF->fy * y + pKF->cy;
    if (!pKF->IsInImage(u, v)) return false;
    ur = u - pKF->mbf * invz;
    if (dist3D < pMP->GetMinDistanceInvariance() || dist3D > pMP->GetMaxDistanceInvariance()) return false;
    if (PO) {
        cv::Mat Pn = pMP->GetNormal();
        double dot = 0;
        for (int k = 0; k < 3; k++) dot += (double)PO[k] * (double)Pn.at<float>(k, 0);
        if (dot < 0.5 * dist3D) return false;
    }
    level = pMP->PredictScale(dist3D, pKF);
    radius = th * pKF->mvScaleFactors[level];
    return true;
}

// ref: src/KeyFrame.cc:1138-1177 -- KeyFrame::GetFeaturesInArea walked on the host over the key frame's mGrid (the drop-in's own
// asks the device): cells touched by the window, |dx| < r and |dy| < r, no level filter
static vector<size_t> refFeaturesInArea(const KeyFrame &K, float x, float y, float r)
{
    vector<size_t> out;
    const int cx0 = std::max(0, (int)floor((x - K.mnMinX - r) * K.mfGridElementWidthInv));
    const int cx1 = std::min(K.mnGridCols - 1, (int)ceil((x - K.mnMinX + r) * K.mfGridElementWidthInv));
    const int cy0 = std::max(0, (int)floor((y - K.mnMinY - r) * K.mfGridElementHeightInv));
    const int cy1 = std::min(K.mnGridRows - 1, (int)ceil((y - K.mnMinY + r) * K.mfGridElementHeightInv));
    if (cx0 >= K.mnGridCols || cx1 < 0 || cy0 >= K.mnGridRows || cy1 < 0) return out;
    for (int ix = cx0; ix <= cx1; ix++)
        for (int iy = cy0; iy <= cy1; iy++) {
            const vector<size_t> &cell = K.mGrid[ix][iy];
            for (size_t j = 0; j < cell.size(); j++) {
                const cv::KeyPoint &kp = K.mvKeysUn[cell[j]];
                if (fabs(kp.pt.x - x) < r && fabs(kp.pt.y - y) < r) out.push_back(cell[j]);
            }
        }
    return out;
}

static float norm3(const float a[3])
{
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)a[k] * (double)a[k];
    return std::sqrt(s);
}

// best feature of the window (gate: Fuse's chi-square test); vpClosed != NULL: features holding a match are skipped
static int bestInWindow(KeyFrame *pKF, MapPoint *pMP, float u, float v, float ur, int level, float radius, bool gate,
                        const vector<MapPoint *> *vpClosed, int &bestDist)
{
    const vector<size_t> vIndices = refFeaturesInArea(*pKF, u, v, radius);
    const cv::Mat dMP = pMP->GetDescriptor();
    bestDist = 256;
    int bestIdx = -1;
    for (size_t c = 0; c < vIndices.size(); c++) {
        const size_t idx = vIndices[c];
        if (vpClosed && (*vpClosed)[idx]) continue;
        const cv::KeyPoint &kp = pKF->mvKeysUn[idx];
        const int kpLevel = kp.octave;
        if (kpLevel < level - 1 || kpLevel > level) continue;
        if (gate) {
            const float ex = u - kp.pt.x, ey = v - kp.pt.y;
            if (pKF->mvuRight[idx] >= 0) {
                const float er = ur - pKF->mvuRight[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * pKF->mvInvLevelSigma2[kpLevel] > 7.8) continue;
            } else {
                const float e2 = ex * ex + ey * ey;
                if (e2 * pKF->mvInvLevelSigma2[kpLevel] > 5.99) continue;
            }
        }
        const int dist = ORBmatcher::DescriptorDistance(dMP, pKF->mDescriptors.row((int)idx));
        if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
    }
    return bestIdx;
}

static void worldPos(MapPoint *p, float xw[3])
{
    cv::Mat m = p->GetWorldPos();
    for (int k = 0; k < 3; k++) xw[k] = m.at<float>(k, 0);
}

static void kfPose(KeyFrame *K, cv::Mat &R, float t[3], float o[3])
{
    R = K->GetRotation();
    cv::Mat tm = K->GetTranslation(), om = K->GetCameraCenter();
    for (int k = 0; k < 3; k++) { t[k] = tm.at<float>(k, 0); o[k] = om.at<float>(k, 0); }
}

static void sim3Pose(const cv::Mat &Scw, cv::Mat &Rcw, float tcw[3], float Ow[3])
{
    double dot = 0;
    for (int k = 0; k < 3; k++) dot += (double)Scw.at<float>(0, k) * (double)Scw.at<float>(0, k);
    const float scw = sqrt(dot);
    Rcw = cv::Mat(3, 3, CV_32F);
    for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) Rcw.at<float>(r, k) = (float)((double)Scw.at<float>(r, k) * (1.0 / scw));
        tcw[r] = (float)((double)Scw.at<float>(r, 3) * (1.0 / scw));
    }
    mul3(Rcw, tcw, NULL, Ow, true, -1.0);
}

static int refFuse(KeyFrame *pKF, const vector<MapPoint *> &vpMapPoints, float th)
{
    cv::Mat Rcw; float tcw[3], Ow[3];
    kfPose(pKF, Rcw, tcw, Ow);
    int nFused = 0;
    for (size_t i = 0; i < vpMapPoints.size(); i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        float xw[3], pc[3], PO[3], u, v, ur, radius; int level, bestDist;
        worldPos(pMP, xw);
        mul3(Rcw, xw, tcw, pc);
        for (int k = 0; k < 3; k++) PO[k] = xw[k] - Ow[k];
        if (!window(pKF, pMP, pc, PO, norm3(PO), th, u, v, ur, level, radius)) continue;
        const int bestIdx = bestInWindow(pKF, pMP, u, v, ur, level, radius, true, NULL, bestDist);
        if (bestIdx >= 0 && bestDist <= ORBmatcher::TH_LOW) {
            MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) {
                    if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                    else pMPinKF->Replace(pMP);
                }
            } else {
                pMP->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(pMP, bestIdx);
            }
            nFused++;
        }
    }
    return nFused;
}

static int refFuseScw(KeyFrame *pKF, const cv::Mat &Scw, const vector<MapPoint *> &vpPoints, float th, vector<MapPoint *> &vpReplacePoint)
{
    cv::Mat Rcw; float tcw[3], Ow[3];
    sim3Pose(Scw, Rcw, tcw, Ow);
    const std::set<MapPoint *> spAlreadyFound = pKF->GetMapPoints();
    int nFused = 0;
    for (size_t i = 0; i < vpPoints.size(); i++) {
        MapPoint *pMP = vpPoints[i];
        if (!pMP || pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        float xw[3], pc[3], PO[3], u, v, ur, radius; int level, bestDist;
        worldPos(pMP, xw);
        mul3(Rcw, xw, tcw, pc);
        for (int k = 0; k < 3; k++) PO[k] = xw[k] - Ow[k];
        if (!window(pKF, pMP, pc, PO, norm3(PO), th, u, v, ur, level, radius)) continue;
        const int bestIdx = bestInWindow(pKF, pMP, u, v, ur, level, radius, false, NULL, bestDist);
        if (bestIdx >= 0 && bestDist <= ORBmatcher::TH_LOW) {
            MapPoint *pMPinKF = pKF->GetMapPoint(bestIdx);
            if (pMPinKF) {
                if (!pMPinKF->isBad()) vpReplacePoint[i] = pMPinKF;
            } else {
                pMP->AddObservation(pKF, bestIdx);
                pKF->AddMapPoint(pMP, bestIdx);
            }
            nFused++;
        }
    }
    return nFused;
}

static int refProjScw(KeyFrame *pKF, const cv::Mat &Scw, const vector<MapPoint *> &vpPoints, vector<MapPoint *> &vpMatched, int th)
{
    cv::Mat Rcw; float tcw[3], Ow[3];
    sim3Pose(Scw, Rcw, tcw, Ow);
    std::set<MapPoint *> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPoint *>(NULL));
    int nmatches = 0;
    for (size_t i = 0; i < vpPoints.size(); i++) {
        MapPoint *pMP = vpPoints[i];
        if (!pMP || pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        float xw[3], pc[3], PO[3], u, v, ur, radius; int level, bestDist;
        worldPos(pMP, xw);
        mul3(Rcw, xw, tcw, pc);
        for (int k = 0; k < 3; k++) PO[k] = xw[k] - Ow[k];
        if (!window(pKF, pMP, pc, PO, norm3(PO), (float)th, u, v, ur, level, radius)) continue;
        const int bestIdx = bestInWindow(pKF, pMP, u, v, ur, level, radius, false, &vpMatched, bestDist);
        if (bestIdx >= 0 && bestDist <= ORBmatcher::TH_LOW) { vpMatched[bestIdx] = pMP; nmatches++; }
    }
    return nmatches;
}

static int refSim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint *> &vpMatches12, float s12, const cv::Mat &R12, const cv::Mat &t12, float th)
{
    cv::Mat R1w, R2w; float t1w[3], t2w[3], o[3];
    kfPose(pKF1, R1w, t1w, o);
    kfPose(pKF2, R2w, t2w, o);
    cv::Mat sR12(3, 3, CV_32F), sR21(3, 3, CV_32F);
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) {
            sR12.at<float>(r, k) = (float)((double)s12 * (double)R12.at<float>(r, k));
            sR21.at<float>(r, k) = (float)((1.0 / s12) * (double)R12.at<float>(k, r));
        }
    float t12v[3] = {t12.at<float>(0, 0), t12.at<float>(1, 0), t12.at<float>(2, 0)}, t21[3];
    mul3(sR21, t12v, NULL, t21, false, -1.0);
    const vector<MapPoint *> vp1 = pKF1->GetMapPointMatches(), vp2 = pKF2->GetMapPointMatches();
    const int N1 = (int)vp1.size(), N2 = (int)vp2.size();
    vector<bool> done1(N1, false), done2(N2, false);
    for (int i = 0; i < N1; i++)
        if (vpMatches12[i]) {
            done1[i] = true;
            const int idx2 = vpMatches12[i]->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) done2[idx2] = true;
        }
    vector<int> m1(N1, -1), m2(N2, -1);
    for (int dir = 0; dir < 2; dir++) {
        const vector<MapPoint *> &vp = dir ? vp2 : vp1;
        for (int i = 0; i < (int)vp.size(); i++) {
            MapPoint *pMP = vp[i];
            if (!pMP || (dir ? done2[i] : done1[i]) || pMP->isBad()) continue;
            float xw[3], pa[3], pb[3], u, v, ur, radius; int level, bestDist;
            worldPos(pMP, xw);
            if (!dir) { mul3(R1w, xw, t1w, pa); mul3(sR21, pa, t21, pb); }
            else { mul3(R2w, xw, t2w, pa); mul3(sR12, pa, t12v, pb); }
            KeyFrame *dst = dir ? pKF1 : pKF2;
            if (!window(dst, pMP, pb, NULL, norm3(pb), th, u, v, ur, level, radius)) continue;
            const int bestIdx = bestInWindow(dst, pMP, u, v, ur, level, radius, false, NULL, bestDist);
            if (bestIdx >= 0 && bestDist <= ORBmatcher::TH_HIGH) (dir ? m2 : m1)[i] = bestIdx;
        }
    }
    int nFound = 0;
    for (int i1 = 0; i1 < N1; i1++)
        if (m1[i1] >= 0 && m2[m1[i1]] == i1) { vpMatches12[i1] = vp2[m1[i1]]; nFound++; }
    return nFound;
}


// ---------------- rotation consistency (ref: src/ORBmatcher.cc:1629-1673, ComputeThreeMaxima; histogram bins as written at
// :236-243: factor = 1 / HISTO_LENGTH, bin = round(rot * factor)) ----------------
static int rotBin(float a1, float a2)
{
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)round(rot * (1.0f / ORBmatcher::HISTO_LENGTH));
    if (bin == ORBmatcher::HISTO_LENGTH) bin = 0;
    return bin;
}

// the three largest bins; the second / third are dropped when they hold fewer than a tenth of the largest
static void refThreeMaxima(const vector<int> *histo, int L, int keep[3])
{
    int best[3] = {0, 0, 0};
    keep[0] = keep[1] = keep[2] = -1;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > best[0]) {
            best[2] = best[1]; keep[2] = keep[1];
            best[1] = best[0]; keep[1] = keep[0];
            best[0] = s; keep[0] = i;
        } else if (s > best[1]) {
            best[2] = best[1]; keep[2] = keep[1];
            best[1] = s; keep[1] = i;
        } else if (s > best[2]) {
            best[2] = s; keep[2] = i;
        }
    }
    if (best[1] < 0.1f * (float)best[0]) keep[1] = keep[2] = -1;
    else if (best[2] < 0.1f * (float)best[0]) keep[2] = -1;
}

// entries of the bins that are not among the three maxima (what every search un-matches at its end)
template <class Undo>
static int rotPrune(const vector<int> *histo, Undo undo)
{
    int keep[3], dropped = 0;
    refThreeMaxima(histo, ORBmatcher::HISTO_LENGTH, keep);
    for (int i = 0; i < ORBmatcher::HISTO_LENGTH; i++) {
        if (i == keep[0] || i == keep[1] || i == keep[2]) continue;
        for (size_t j = 0; j < histo[i].size(); j++) { undo(histo[i][j]); dropped++; }
    }
    return dropped;
}

// the walk over two FeatureVectors that visits the nodes both hold, in ascending node id (ref: :176-264, the while / lower_bound loop)
template <class Visit>
static void sharedNodes(const DBoW2::FeatureVector &fa, const DBoW2::FeatureVector &fb, Visit visit)
{
    DBoW2::FeatureVector::const_iterator a = fa.begin(), b = fb.begin();
    while (a != fa.end() && b != fb.end()) {
        if (a->first == b->first) { visit(a->second, b->second); ++a; ++b; }
        else if (a->first < b->first) a = fa.lower_bound(b->first);
        else b = fb.lower_bound(a->first);
    }
}

// ref: src/ORBmatcher.cc:159-288 -- SearchByBoW(KeyFrame, Frame): key-frame features with a good map point against the frame's
// unmatched features of the same node; <= TH_LOW, ratio test, rotation histogram
static int refSearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint *> &vpMapPointMatches, float nnratio, bool checkOri)
{
    const vector<MapPoint *> vpKF = pKF->GetMapPointMatches();
    vpMapPointMatches.assign(F.N, static_cast<MapPoint *>(NULL));
    vector<int> hist[30];
    int nmatches = 0;
    sharedNodes(pKF->mFeatVec, F.mFeatVec, [&](const vector<unsigned int> &iKF, const vector<unsigned int> &iF) {
        for (size_t a = 0; a < iKF.size(); a++) {
            const unsigned int i1 = iKF[a];
            MapPoint *pMP = vpKF[i1];
            if (!pMP || pMP->isBad()) continue;
            int d1 = 256, d2 = 256, arg = -1;
            for (size_t b = 0; b < iF.size(); b++) {
                const unsigned int i2 = iF[b];
                if (vpMapPointMatches[i2]) continue;
                const int d = ORBmatcher::DescriptorDistance(pKF->mDescriptors.row((int)i1), F.mDescriptors.row((int)i2));
                if (d < d1) { d2 = d1; d1 = d; arg = (int)i2; }
                else if (d < d2) d2 = d;
            }
            if (d1 <= ORBmatcher::TH_LOW && (float)d1 < nnratio * (float)d2) {
                vpMapPointMatches[arg] = pMP;
                if (checkOri) hist[rotBin(pKF->mvKeysUn[i1].angle, F.mvKeys[arg].angle)].push_back(arg);
                nmatches++;
            }
        }
    });
    if (checkOri) nmatches -= rotPrune(hist, [&](int i2) { vpMapPointMatches[i2] = static_cast<MapPoint *>(NULL); });
    return nmatches;
}

// ref: src/ORBmatcher.cc:522-655 -- SearchByBoW(KeyFrame, KeyFrame): both sides need a good map point, side 2 is claimed,
// strictly below TH_LOW
static int refSearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint *> &vpMatches12, float nnratio, bool checkOri)
{
    const vector<MapPoint *> vp1 = pKF1->GetMapPointMatches(), vp2 = pKF2->GetMapPointMatches();
    vpMatches12.assign(vp1.size(), static_cast<MapPoint *>(NULL));
    vector<bool> taken2(vp2.size(), false);
    vector<int> hist[30];
    int nmatches = 0;
    sharedNodes(pKF1->mFeatVec, pKF2->mFeatVec, [&](const vector<unsigned int> &l1, const vector<unsigned int> &l2) {
        for (size_t a = 0; a < l1.size(); a++) {
            const size_t i1 = l1[a];
            MapPoint *p1 = vp1[i1];
            if (!p1 || p1->isBad()) continue;
            int d1 = 256, d2 = 256, arg = -1;
            for (size_t b = 0; b < l2.size(); b++) {
                const size_t i2 = l2[b];
                MapPoint *p2 = vp2[i2];
                if (taken2[i2] || !p2 || p2->isBad()) continue;
                const int d = ORBmatcher::DescriptorDistance(pKF1->mDescriptors.row((int)i1), pKF2->mDescriptors.row((int)i2));
                if (d < d1) { d2 = d1; d1 = d; arg = (int)i2; }
                else if (d < d2) d2 = d;
            }
            if (d1 < ORBmatcher::TH_LOW && (float)d1 < nnratio * (float)d2) {
                vpMatches12[i1] = vp2[arg];
                taken2[arg] = true;
                if (checkOri) hist[rotBin(pKF1->mvKeysUn[i1].angle, pKF2->mvKeysUn[arg].angle)].push_back((int)i1);
                nmatches++;
            }
        }
    });
    if (checkOri) nmatches -= rotPrune(hist, [&](int i1) { vpMatches12[i1] = static_cast<MapPoint *>(NULL); });
    return nmatches;
}

// ref: src/ORBmatcher.cc:131-157 (CheckDistEpipolarLine) and :657-827 -- SearchForTriangulation: features WITHOUT a map point on
// both sides, same node, distance <= TH_LOW and not above the best so far, not too close to the epipole (monocular pairs),
// within 3.84 sigma^2 of the epipolar line.  (The fork never sets vbMatched2: a feature of the second key frame can be
// the match of several of the first.)
static int refSearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, const cv::Mat &F12, vector<std::pair<size_t, size_t> > &pairs,
                                     bool onlyStereo, bool checkOri)
{
    cv::Mat R2w; float t2w[3], o2[3], C2[3];
    kfPose(pKF2, R2w, t2w, o2);
    cv::Mat Cwm = pKF1->GetCameraCenter();
    const float cw[3] = {Cwm.at<float>(0, 0), Cwm.at<float>(1, 0), Cwm.at<float>(2, 0)};
    mul3(R2w, cw, t2w, C2);
    const float invz = 1.0f / C2[2];
    const float ex = pKF2->fx * C2[0] * invz + pKF2->cx;
    const float ey = pKF2->fy * C2[1] * invz + pKF2->cy;
    vector<int> m12(pKF1->N, -1);
    vector<int> hist[30];
    int nmatches = 0;
    sharedNodes(pKF1->mFeatVec, pKF2->mFeatVec, [&](const vector<unsigned int> &l1, const vector<unsigned int> &l2) {
        for (size_t a = 0; a < l1.size(); a++) {
            const size_t i1 = l1[a];
            if (pKF1->GetMapPoint(i1)) continue;
            const bool stereo1 = pKF1->mvuRight[i1] >= 0;
            if (onlyStereo && !stereo1) continue;
            const cv::KeyPoint &kp1 = pKF1->mvKeysUn[i1];
            int best = ORBmatcher::TH_LOW, arg = -1;
            for (size_t b = 0; b < l2.size(); b++) {
                const size_t i2 = l2[b];
                if (pKF2->GetMapPoint(i2)) continue;
                const bool stereo2 = pKF2->mvuRight[i2] >= 0;
                if (onlyStereo && !stereo2) continue;
                const int d = ORBmatcher::DescriptorDistance(pKF1->mDescriptors.row((int)i1), pKF2->mDescriptors.row((int)i2));
                if (d > ORBmatcher::TH_LOW || d > best) continue;
                const cv::KeyPoint &kp2 = pKF2->mvKeysUn[i2];
                if (!stereo1 && !stereo2) {
                    const float dx = ex - kp2.pt.x, dy = ey - kp2.pt.y;
                    if (dx * dx + dy * dy < 100 * pKF2->mvScaleFactors[kp2.octave]) continue;
                }
                const float la = kp1.pt.x * F12.at<float>(0, 0) + kp1.pt.y * F12.at<float>(1, 0) + F12.at<float>(2, 0);
                const float lb = kp1.pt.x * F12.at<float>(0, 1) + kp1.pt.y * F12.at<float>(1, 1) + F12.at<float>(2, 1);
                const float lc = kp1.pt.x * F12.at<float>(0, 2) + kp1.pt.y * F12.at<float>(1, 2) + F12.at<float>(2, 2);
                const float num = la * kp2.pt.x + lb * kp2.pt.y + lc;
                const float den = la * la + lb * lb;
                if (den == 0) continue;
                const float dsqr = num * num / den;
                if (dsqr < 3.84 * pKF2->mvLevelSigma2[kp2.octave]) { arg = (int)i2; best = d; }
            }
            if (arg >= 0) {
                m12[i1] = arg;
                nmatches++;
                if (checkOri) hist[rotBin(kp1.angle, pKF2->mvKeysUn[arg].angle)].push_back((int)i1);
            }
        }
    });
    if (checkOri) nmatches -= rotPrune(hist, [&](int i1) { m12[i1] = -1; });
    pairs.clear();
    for (size_t i = 0; i < m12.size(); i++)
        if (m12[i] >= 0) pairs.push_back(std::make_pair(i, (size_t)m12[i]));
    return nmatches;
}

// ref: src/Frame.cc:671-724 -- Frame::GetFeaturesInArea walked on the host over the frame's mGrid (the drop-in's own
// GetFeaturesInArea asks the device): cells touched by the window, optional octave range, |dx| < r and |dy| < r
static vector<size_t> refFeaturesInArea(const Frame &F, float x, float y, float r, int minLevel = -1, int maxLevel = -1)
{
    vector<size_t> out;
    const int cx0 = std::max(0, (int)floor((x - Frame::mnMinX - r) * Frame::mfGridElementWidthInv));
    const int cx1 = std::min((int)FRAME_GRID_COLS - 1, (int)ceil((x - Frame::mnMinX + r) * Frame::mfGridElementWidthInv));
    const int cy0 = std::max(0, (int)floor((y - Frame::mnMinY - r) * Frame::mfGridElementHeightInv));
    const int cy1 = std::min((int)FRAME_GRID_ROWS - 1, (int)ceil((y - Frame::mnMinY + r) * Frame::mfGridElementHeightInv));
    if (cx0 >= FRAME_GRID_COLS || cx1 < 0 || cy0 >= FRAME_GRID_ROWS || cy1 < 0) return out;
    const bool levels = minLevel > 0 || maxLevel >= 0;
    for (int ix = cx0; ix <= cx1; ix++)
        for (int iy = cy0; iy <= cy1; iy++) {
            const vector<size_t> &cell = F.mGrid[ix][iy];
            for (size_t j = 0; j < cell.size(); j++) {
                const cv::KeyPoint &kp = F.mvKeysUn[cell[j]];
                if (levels && (kp.octave < minLevel || (maxLevel >= 0 && kp.octave > maxLevel))) continue;
                if (fabs(kp.pt.x - x) < r && fabs(kp.pt.y - y) < r) out.push_back(cell[j]);
            }
        }
    return out;
}

// ref: src/ORBmatcher.cc:1341-1498 -- SearchByProjection(CurrentFrame, LastFrame, th, bMono): the last frame's inlier map points
// projected with the current pose, window by the last octave, best distance <= TH_HIGH, rotation histogram
static int refSearchByProjection(Frame &Cur, const Frame &Last, float th, bool bMono, bool checkOri)
{
    cv::Mat Rcw(3, 3, CV_32F), Rlw(3, 3, CV_32F);
    float tcw[3], tlw[3], twc[3], tlc[3];
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) { Rcw.at<float>(r, c) = Cur.mTcw.at<float>(r, c); Rlw.at<float>(r, c) = Last.mTcw.at<float>(r, c); }
        tcw[r] = Cur.mTcw.at<float>(r, 3);
        tlw[r] = Last.mTcw.at<float>(r, 3);
    }
    mul3(Rcw, tcw, NULL, twc, true, -1.0);
    mul3(Rlw, twc, tlw, tlc);
    const bool forward = tlc[2] > Cur.mb && !bMono, backward = -tlc[2] > Cur.mb && !bMono;
    vector<int> hist[30];
    int nmatches = 0;
    for (int i = 0; i < Last.N; i++) {
        MapPoint *pMP = Last.mvpMapPoints[i];
        if (!pMP || Last.mvbOutlier[i]) continue;
        float xw[3], pc[3];
        worldPos(pMP, xw);
        mul3(Rcw, xw, tcw, pc);
        const float xc = pc[0], yc = pc[1];
        const float invzc = 1.0 / pc[2];
        if (invzc < 0) continue;
        const float u = Cur.fx * xc * invzc + Cur.cx, v = Cur.fy * yc * invzc + Cur.cy;
        if (u < Cur.mnMinX || u > Cur.mnMaxX || v < Cur.mnMinY || v > Cur.mnMaxY) continue;
        const int oct = Last.mvKeys[i].octave;
        const float radius = th * Cur.mvScaleFactors[oct];
        const vector<size_t> cand = forward ? refFeaturesInArea(Cur, u, v, radius, oct)
                                  : backward ? refFeaturesInArea(Cur, u, v, radius, 0, oct)
                                             : refFeaturesInArea(Cur, u, v, radius, oct - 1, oct + 1);
        if (cand.empty()) continue;
        const cv::Mat dMP = pMP->GetDescriptor();
        int best = 256, arg = -1;
        for (size_t c = 0; c < cand.size(); c++) {
            const size_t i2 = cand[c];
            if (Cur.mvpMapPoints[i2] && Cur.mvpMapPoints[i2]->Observations() > 0) continue;
            if (!Cur.mvuRight.empty() && Cur.mvuRight[i2] > 0) {
                const float ur = u - Cur.mbf * invzc;
                if (fabs(ur - Cur.mvuRight[i2]) > radius) continue;
            }
            const int d = ORBmatcher::DescriptorDistance(dMP, Cur.mDescriptors.row((int)i2));
            if (d < best) { best = d; arg = (int)i2; }
        }
        if (best <= ORBmatcher::TH_HIGH) {
            Cur.mvpMapPoints[arg] = pMP;
            nmatches++;
            if (checkOri) hist[rotBin(Last.mvKeysUn[i].angle, Cur.mvKeysUn[arg].angle)].push_back(arg);
        }
    }
    if (checkOri) nmatches -= rotPrune(hist, [&](int i2) { Cur.mvpMapPoints[i2] = static_cast<MapPoint *>(NULL); });
    return nmatches;
}

#endif
