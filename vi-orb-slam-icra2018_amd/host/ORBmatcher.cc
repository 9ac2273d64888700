// ORBmatcher.cc -- host side of the drop-in ORB_SLAM2::ORBmatcher (include/orbhip/ORBmatcher.h).
// SearchByBoW keeps the reference's interface and result conventions (src/ORBmatcher.cc:159-288,
// :522-655): the FeatureVectors are flattened to CSR in std::map order, "has a good MapPoint"
// becomes a byte mask, and the node-constrained brute force + ratio test + rotation histogram run
// in liborbhip.so (orbhip_search_by_bow).
#include "ORBmatcher.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <string>

#include "ORBextractor.h"
#include "hiperror.h"
#include "orbhip.h"

using namespace std;

namespace ORB_SLAM2
{

const int ORBmatcher::TH_HIGH = 100;     // ref: src/ORBmatcher.cc:37-39
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

static int g_match_device = 0;
void ORBmatcher::SetDevice(int device) { g_match_device = device; }
static std::atomic<int> g_set_limit(0);                  // 0: the library's default
void ORBmatcher::SetResidentSetLimit(int n) { g_set_limit.store(n < 4 ? 4 : n); }

namespace {
// one small device context per host thread (matchers are used concurrently from Tracking,
// LocalMapping and LoopClosing; a context is not re-entrant)
struct ThreadCtx {
    orbhip_ctx *ctx = nullptr;
    int limit = 0;                                       // the value of g_set_limit this thread's table runs with
    ~ThreadCtx() { if (ctx) orbhip_destroy(ctx); }
    orbhip_ctx *get()
    {
        if (!ctx) {
            ctx = orbhip_create(g_match_device, 50, 1.2f, 1, 20, 7, 128, 128, 1);
            if (!ctx) hipdetail::Fail("ORBmatcher (device context)", orbhip_last_error(nullptr));   // every C call rejects a null context
        }
        return ctx;
    }
};
thread_local ThreadCtx tls;

struct Csr {
    vector<int32_t> node, off, idx;
};
Csr flatten(const DBoW2::FeatureVector &fv)
{
    Csr c;
    c.off.push_back(0);
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
        c.node.push_back((int32_t)it->first);
        for (size_t k = 0; k < it->second.size(); k++) c.idx.push_back((int32_t)it->second[k]);
        c.off.push_back((int32_t)c.idx.size());
    }
    return c;
}
vector<uint8_t> contiguous(const cv::Mat &d)
{
    vector<uint8_t> v((size_t)d.rows * 32);
    for (int i = 0; i < d.rows; i++) memcpy(&v[(size_t)i * 32], d.ptr(i), 32);
    return v;
}

// Resident feature sets (include/orbhip.h, orbhip_set_*): descriptors, undistorted keypoints, FeatureVector and feature grid of
// a key frame or frame stay on the device under its id once a matcher of this thread has met it (a set nobody uses any
// more ages out of the 96 the context keeps).  An id is not an identity: Tracking::Reset restarts KeyFrame::nNextId and
// Frame::nNextId (ref: src/Tracking.cc:2758-2759), and a key frame can be met before KeyFrame::ComputeBoW has filled its
// FeatureVector (ref: src/KeyFrame.cc:392-400).  So a resident set is a hit only if its feature count, its FeatureVector
// size and its fingerprint (first keypoint, first and last descriptor) are those of the object in hand; otherwise it is
// put again.  A Frame that its extractor has just built (ORBextractor::SetFrameBuild) enters from the device block of that
// build -- only its FeatureVector travels.  ORBmatcher::DropResidentSets() (for Tracking::Reset) empties the
// calling thread's table.  ORBHIP_NO_SETS=1 restores the upload-per-call entry points (A/B runs).
const uint64_t KF_KEY = 1ull << 62, FRAME_KEY = 1ull << 61;
bool use_sets()
{
    static const bool off = getenv("ORBHIP_NO_SETS") && atoi(getenv("ORBHIP_NO_SETS")) != 0;
    return !off;
}
template <class T>
bool ensure_set(uint64_t key, const T &t, const vector<cv::KeyPoint> &keysUn, float minX, float minY, float invW, float invH,
                orbhip_ctx *builder)
{
    const int n = t.mDescriptors.rows;
    if (n <= 0 || (int)keysUn.size() != n) return false;
    orbhip_ctx *c = tls.get();
    const int lim = g_set_limit.load();
    if (lim != tls.limit && c) {
        orbhip_set_limit(c, lim);
        tls.limit = lim;
    }
    const uint64_t fp = orbhip_set_fingerprint_rows(reinterpret_cast<const orbhip_keypoint *>(keysUn.data()), t.mDescriptors.ptr(0),
                                                    t.mDescriptors.ptr(n - 1), n);
    int n0 = 0, ng0 = 0;
    uint64_t fp0 = 0;
    if (orbhip_set_info(c, key, &n0, &ng0, &fp0) && n0 == n && fp0 == fp && ng0 == (int)t.mFeatVec.size()) return true;
    const Csr fv = flatten(t.mFeatVec);
    // the frame its extractor built last is still on the device: block to block, the FeatureVector alone travels
    if (builder && orbhip_frame_fingerprint(builder) == fp &&
        orbhip_set_put_from_frame(c, key, builder, fv.node.data(), fv.off.data(), fv.idx.data(), (int)fv.node.size()) == ORBHIP_OK)
        return true;
    const vector<uint8_t> d = contiguous(t.mDescriptors);
    return orbhip_set_put(c, key, reinterpret_cast<const orbhip_keypoint *>(keysUn.data()), d.data(), n, fv.node.data(),
                          fv.off.data(), fv.idx.data(), (int)fv.node.size(), minX, minY, invW, invH) == ORBHIP_OK;
}
bool ensure_set(KeyFrame *pKF)
{
    return ensure_set(KF_KEY | (uint64_t)(pKF->mnId + 1), *pKF, pKF->mvKeysUn, pKF->mnMinX, pKF->mnMinY, pKF->mfGridElementWidthInv,
                      pKF->mfGridElementHeightInv, NULL);
}
bool ensure_set(Frame &F)
{
    return ensure_set(FRAME_KEY | (uint64_t)(F.mnId + 1), F, F.mvKeysUn, Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv,
                      Frame::mfGridElementHeightInv, F.mpORBextractorLeft ? F.mpORBextractorLeft->Context() : NULL);
}
}  // namespace

ORBmatcher::ORBmatcher(float nnratio, bool checkOri): mfNNratio(nnratio), mbCheckOrientation(checkOri)
{
}

ORBmatcher::~ORBmatcher() {}

void ORBmatcher::DropResidentSets()
{
    if (tls.ctx) orbhip_set_drop(tls.ctx, 0);
}

int ORBmatcher::SearchByBoW(KeyFrame* pKF,Frame &F, vector<MapPoint*> &vpMapPointMatches)
{
    const vector<MapPoint*> kfPoints(pKF->GetMapPointMatches());
    vpMapPointMatches.assign(F.N, (MapPoint *)0);

    const int n1 = pKF->mDescriptors.rows, n2 = F.mDescriptors.rows;
    if (n1 == 0 || n2 == 0) return 0;
    vector<uint8_t> valid1(n1, 0);
    vector<float> a1(n1, 0.f), a2(n2, 0.f);
    for (int i = 0; i < n1; i++) {
        MapPoint *pMP = i < (int)kfPoints.size() ? kfPoints[i] : NULL;
        valid1[i] = (pMP && !pMP->isBad()) ? 1 : 0;          // ref: :193-199
        a1[i] = pKF->mvKeysUn[i].angle;                        // ref: :234
    }
    for (int i = 0; i < n2; i++) a2[i] = F.mvKeys[i].angle;   // ref: :238
    vector<int32_t> m12(n1), m21(n2);
    int found = 0;
    // (the rotation check reads pKF->mvKeysUn[i].angle and F.mvKeys[i].angle: undistortion leaves the angle alone, so the
    // resident sets -- built from the undistorted keypoints -- hold the same values)
    if (use_sets() && (int)F.mvKeysUn.size() == n2 && ensure_set(pKF) && ensure_set(F)) {
        const int rc = orbhip_search_by_bow_sets(tls.get(), KF_KEY | (uint64_t)(pKF->mnId + 1), valid1.data(),
                                                 FRAME_KEY | (uint64_t)(F.mnId + 1), NULL, TH_LOW, 0, mfNNratio,
                                                 mbCheckOrientation ? 1 : 0, m12.data(), m21.data(), &found);
        if (rc != ORBHIP_OK) return hipdetail::Fail("ORBmatcher::SearchByBoW", orbhip_last_error(tls.get())), 0;
        for (int i2 = 0; i2 < n2 && i2 < F.N; i2++)
            if (m21[i2] >= 0) vpMapPointMatches[i2] = kfPoints[m21[i2]];   // ref: :232
        return found;
    }
    const Csr c1 = flatten(pKF->mFeatVec), c2 = flatten(F.mFeatVec);
    const vector<uint8_t> d1 = contiguous(pKF->mDescriptors), d2 = contiguous(F.mDescriptors);
    const int rc = orbhip_search_by_bow(tls.get(), d1.data(), n1, valid1.data(), a1.data(), c1.node.data(),
                                        c1.off.data(), c1.idx.data(), (int)c1.node.size(), d2.data(), n2, NULL,
                                        a2.data(), c2.node.data(), c2.off.data(), c2.idx.data(), (int)c2.node.size(),
                                        TH_LOW, 0, mfNNratio, mbCheckOrientation ? 1 : 0, m12.data(), m21.data(),
                                        &found);
    if (rc != ORBHIP_OK) return hipdetail::Fail("ORBmatcher::SearchByBoW", orbhip_last_error(tls.get())), 0;
    for (int i2 = 0; i2 < n2 && i2 < F.N; i2++)
        if (m21[i2] >= 0) vpMapPointMatches[i2] = kfPoints[m21[i2]];   // ref: :232
    return found;
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint *> &vpMatches12)
{
    const vector<MapPoint*> points1(pKF1->GetMapPointMatches()), points2(pKF2->GetMapPointMatches());
    vpMatches12.assign(points1.size(), (MapPoint *)0);

    const int n1 = pKF1->mDescriptors.rows, n2 = pKF2->mDescriptors.rows;
    if (n1 == 0 || n2 == 0) return 0;
    vector<uint8_t> valid1(n1, 0), valid2(n2, 0);
    vector<float> a1(n1, 0.f), a2(n2, 0.f);
    for (int i = 0; i < n1; i++) {
        MapPoint *p = i < (int)points1.size() ? points1[i] : NULL;
        valid1[i] = (p && !p->isBad()) ? 1 : 0;               // ref: :556-560
        a1[i] = pKF1->mvKeysUn[i].angle;
    }
    for (int i = 0; i < n2; i++) {
        MapPoint *p = i < (int)points2.size() ? points2[i] : NULL;
        valid2[i] = (p && !p->isBad()) ? 1 : 0;               // ref: :572-578
        a2[i] = pKF2->mvKeysUn[i].angle;
    }
    vector<int32_t> m12(n1), m21(n2);
    int found = 0;
    if (use_sets() && pKF1 != pKF2 && ensure_set(pKF1) && ensure_set(pKF2)) {
        const int rc = orbhip_search_by_bow_sets(tls.get(), KF_KEY | (uint64_t)(pKF1->mnId + 1), valid1.data(),
                                                 KF_KEY | (uint64_t)(pKF2->mnId + 1), valid2.data(), TH_LOW, 1, mfNNratio,
                                                 mbCheckOrientation ? 1 : 0, m12.data(), m21.data(), &found);
        if (rc != ORBHIP_OK) return hipdetail::Fail("ORBmatcher::SearchByBoW", orbhip_last_error(tls.get())), 0;
        for (int i1 = 0; i1 < n1 && i1 < (int)vpMatches12.size(); i1++)
            if (m12[i1] >= 0) vpMatches12[i1] = points2[m12[i1]];           // ref: :602
        return found;
    }
    const Csr c1 = flatten(pKF1->mFeatVec), c2 = flatten(pKF2->mFeatVec);
    const vector<uint8_t> d1 = contiguous(pKF1->mDescriptors), d2 = contiguous(pKF2->mDescriptors);
    const int rc = orbhip_search_by_bow(tls.get(), d1.data(), n1, valid1.data(), a1.data(), c1.node.data(),
                                        c1.off.data(), c1.idx.data(), (int)c1.node.size(), d2.data(), n2,
                                        valid2.data(), a2.data(), c2.node.data(), c2.off.data(), c2.idx.data(),
                                        (int)c2.node.size(), TH_LOW, 1, mfNNratio, mbCheckOrientation ? 1 : 0,
                                        m12.data(), m21.data(), &found);
    if (rc != ORBHIP_OK) return hipdetail::Fail("ORBmatcher::SearchByBoW", orbhip_last_error(tls.get())), 0;
    for (int i1 = 0; i1 < n1 && i1 < (int)vpMatches12.size(); i1++)
        if (m12[i1] >= 0) vpMatches12[i1] = points2[m12[i1]];           // ref: :602
    return found;
}

float ORBmatcher::RadiusByViewingCos(const float &viewCos)
{
    if(viewCos>0.998)                                          // ref: src/ORBmatcher.cc:131-137
        return 2.5;
    else
        return 4.0;
}

namespace {
// Runs the window search of one frame on the device and writes the result into F.mvpMapPoints the way the
// reference's loops do: the last point assigned to a feature stays, a feature whose match the rotation check
// removed becomes NULL, every other feature is left alone.
int run_projection_search(Frame &F, const vector<orbhip_proj_query> &q, const vector<uint8_t> &qdesc,
                          const vector<MapPoint *> &source, bool use_ratio, float nnratio, bool check_ori, int th_high,
                          bool anyPointCloses = false, bool useRight = true)
{
    const int n = F.N, nq = (int)q.size();
    if (n == 0 || nq == 0) return 0;
    vector<uint8_t> occupied(n, 0);
    for (int i = 0; i < n; i++)
        if (F.mvpMapPoints[i] && (anyPointCloses || F.mvpMapPoints[i]->Observations() > 0))
            occupied[i] = 1;   // ref: :88-90, :1413-1415 (observed points only); :1565-1566 (any point)
    const vector<uint8_t> d = contiguous(F.mDescriptors);
    vector<int32_t> match(n);
    int found = 0;
    const int rc = orbhip_search_by_projection(
        tls.get(), reinterpret_cast<const orbhip_keypoint *>(F.mvKeysUn.data()), d.data(), n,
        (useRight && (int)F.mvuRight.size() == n) ? F.mvuRight.data() : NULL, occupied.data(), Frame::mnMinX, Frame::mnMinY,
        Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv, q.data(), qdesc.data(), nq, use_ratio ? 1 : 0, nnratio,
        check_ori ? 1 : 0, th_high, match.data(), &found);
    if (rc != ORBHIP_OK) return hipdetail::Fail("ORBmatcher::SearchByProjection", orbhip_last_error(tls.get())), 0;
    for (int i = 0; i < n; i++) {
        if (match[i] >= 0)
            F.mvpMapPoints[i] = source[match[i]];
        else if (match[i] == -2)
            F.mvpMapPoints[i] = static_cast<MapPoint *>(NULL);
    }
    return found;
}

// d = R * x + t for 3x3 / 3x1 float matrices.  OpenCV evaluates the MatExpr Rcw*x3Dw+tcw as one gemm whose
// float kernel accumulates in double and rounds once (modules/core/src/matmul.cpp, GEMMSingleMul<float,double>).
void affine3(const cv::Mat &R, const float x[3], const float t[3], float out[3], bool transpose = false, double alpha = 1.0)
{
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)(transpose ? R.at<float>(k, r) : R.at<float>(r, k)) * (double)x[k];
        out[r] = (float)(alpha * s + (t ? (double)t[r] : 0.0));
    }
}
}  // namespace

namespace {
// A frame's camera as the guided searches of Tracking use it: world -> camera -> pixel, and the bounds of the undistorted
// image.  The float operations are those of the reference's loops in their order (the native guided test pins them): the
// camera coordinates come out of one gemm (affine3), the reciprocal depth is a double division rounded to float, a pixel
// coordinate is ((f * coordinate) * reciprocal depth) + principal point in float.
struct FrameCamera {
    cv::Mat R;                 // rotation block of mTcw
    float t[3];                // translation column of mTcw
    const Frame *F;

    explicit FrameCamera(const Frame &frame): R(frame.mTcw.rowRange(0,3).colRange(0,3)), F(&frame)
    {
        for (int r = 0; r < 3; r++) t[r] = frame.mTcw.at<float>(r, 3);
    }
    void centre(float out[3]) const { affine3(R, t, NULL, out, true, -1.0); }   // -R' t
    // false where the point lies outside the undistorted image; *invz is set either way (its sign is the caller's test)
    bool project(const float xw[3], float *u, float *v, float *invz) const
    {
        float pc[3];
        affine3(R, xw, t, pc);
        const float iz = 1.0/pc[2];
        *invz = iz;
        *u = F->fx*pc[0]*iz+F->cx;
        *v = F->fy*pc[1]*iz+F->cy;
        return !(*u<Frame::mnMinX || *u>Frame::mnMaxX || *v<Frame::mnMinY || *v>Frame::mnMaxY);
    }
};

inline void world_point(MapPoint *pMP, float xw[3])
{
    const cv::Mat p = pMP->GetWorldPos();
    for (int k = 0; k < 3; k++) xw[k] = p.at<float>(k, 0);
}

inline void query_descriptor(MapPoint *pMP, vector<uint8_t> &qdesc, int slot)
{
    const cv::Mat d = pMP->GetDescriptor();
    memcpy(&qdesc[(size_t)slot * 32], d.ptr(0), 32);
}
}  // namespace

int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)
{
    // Tracking::SearchLocalPoints (ref: src/ORBmatcher.cc:41-128): the points Frame::isInFrustum marked, each with the
    // projection and the octave it stored in the point.
    const bool widen = th!=1.0;
    const int nq = (int)vpMapPoints.size();
    vector<orbhip_proj_query> q(nq);
    vector<uint8_t> qdesc((size_t)nq * 32, 0);
    memset(q.data(), 0, sizeof(orbhip_proj_query) * (size_t)nq);
    for (int k = 0; k < nq; k++) {
        MapPoint *pt = vpMapPoints[k];
        if (!pt->mbTrackInView || pt->isBad()) continue;       // ref: :55-59
        const int level = pt->mnTrackScaleLevel;
        // window radius: narrow when the point is seen almost head-on, wider otherwise (ref: :130-138), times th on request
        float radius = RadiusByViewingCos(pt->mTrackViewCos);
        if (widen) radius*=th;
        orbhip_proj_query &e = q[k];
        e.u = pt->mTrackProjX;                                 // ref: :68-69
        e.v = pt->mTrackProjY;
        e.radius = radius*F.mvScaleFactors[level];
        e.min_level = level-1;
        e.max_level = level;
        e.proj_xr = pt->mTrackProjXR;                          // ref: :94
        e.flags = ORBHIP_Q_ACTIVE | (pt->Observations()>0 ? ORBHIP_Q_OBSERVED : 0);
        query_descriptor(pt, qdesc, k);
    }
    return run_projection_search(F, q, qdesc, vpMapPoints, true, mfNNratio, false, TH_HIGH);
}

// Tracking::TrackWithMotionModel (ref: src/ORBmatcher.cc:1340-1498): every point the last frame holds is projected with the
// current frame's predicted pose and searched in a window around the projection, on the octaves the camera's motion along its
// axis allows.
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
{
    const FrameCamera cam(CurrentFrame), last(LastFrame);
    // the current camera's centre in the last camera's frame: its z says whether the camera moved forward or backward by more
    // than the stereo baseline (ref: :1351-1365; never for monocular input)
    float centreW[3], centreL[3];
    cam.centre(centreW);
    affine3(last.R, centreW, last.t, centreL);
    enum { SAME, FORWARD, BACKWARD } motion = SAME;
    if (!bMono && centreL[2]>CurrentFrame.mb) motion = FORWARD;
    else if (!bMono && -centreL[2]>CurrentFrame.mb) motion = BACKWARD;

    const int nq = LastFrame.N;
    vector<orbhip_proj_query> q(nq);
    vector<uint8_t> qdesc((size_t)nq * 32, 0);
    memset(q.data(), 0, sizeof(orbhip_proj_query) * (size_t)nq);
    for (int i = 0; i < nq; i++) {
        MapPoint *pMP = LastFrame.mvpMapPoints[i];
        if (!pMP || LastFrame.mvbOutlier[i]) continue;
        float xw[3], u, v, invz;
        world_point(pMP, xw);
        const bool inside = cam.project(xw, &u, &v, &invz);    // ref: :1376-1398
        if (invz<0 || !inside) continue;
        const int octave = LastFrame.mvKeys[i].octave;
        orbhip_proj_query &e = q[i];
        e.u = u;
        e.v = v;
        e.radius = th*CurrentFrame.mvScaleFactors[octave];     // ref: :1403-1416
        e.min_level = motion == FORWARD ? octave : motion == BACKWARD ? 0 : octave-1;
        e.max_level = motion == FORWARD ? -1 : motion == BACKWARD ? octave : octave+1;
        e.proj_xr = u - CurrentFrame.mbf*invz;                 // ref: :1435
        e.angle = LastFrame.mvKeysUn[i].angle;                 // ref: :1461
        e.flags = ORBHIP_Q_ACTIVE | (pMP->Observations()>0 ? ORBHIP_Q_OBSERVED : 0);
        query_descriptor(pMP, qdesc, i);
    }
    return run_projection_search(CurrentFrame, q, qdesc, LastFrame.mvpMapPoints, false, mfNNratio, mbCheckOrientation,
                                 TH_HIGH);
}

// Tracking::Relocalization (ref: src/ORBmatcher.cc:1500-1627): the key frame's points that PnP has not matched yet, projected
// with the pose PnP found.  Best only, no right-coordinate test, every feature that holds a point is closed (:1565-1566),
// threshold ORBdist, rotation histogram.
int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th , const int ORBdist)
{
    const FrameCamera cam(CurrentFrame);
    float Ow[3];
    cam.centre(Ow);

    const vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
    const int nq = (int)vpMPs.size();
    vector<orbhip_proj_query> q(nq);
    vector<uint8_t> qdesc((size_t)nq * 32, 0);
    memset(q.data(), 0, sizeof(orbhip_proj_query) * (size_t)nq);
    for (int i = 0; i < nq; i++) {
        MapPoint *pMP = vpMPs[i];
        if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        float xw[3], u, v, invz;
        world_point(pMP, xw);
        if (!cam.project(xw, &u, &v, &invz)) continue;         // ref: :1524-1539 (no depth-sign test here)
        // the octave the point should appear on at this distance (ref: :1541-1553); cv::norm sums the squares in double
        double sq = 0;
        for (int k = 0; k < 3; k++) {
            const float po = xw[k]-Ow[k];
            sq += (double)po*(double)po;
        }
        const float dist3D = std::sqrt(sq);
        if (dist3D<pMP->GetMinDistanceInvariance() || dist3D>pMP->GetMaxDistanceInvariance()) continue;
        const int level = pMP->PredictScale(dist3D,&CurrentFrame);
        orbhip_proj_query &e = q[i];
        e.u = u;
        e.v = v;
        e.radius = th*CurrentFrame.mvScaleFactors[level];      // ref: :1555-1558
        e.min_level = level-1;
        e.max_level = level+1;
        e.angle = pKF->mvKeysUn[i].angle;                      // ref: :1587
        e.flags = ORBHIP_Q_ACTIVE | ORBHIP_Q_OBSERVED;
        query_descriptor(pMP, qdesc, i);
    }
    return run_projection_search(CurrentFrame, q, qdesc, vpMPs, false, mfNNratio, mbCheckOrientation, ORBdist, true, false);
}

namespace {
// ---- the KeyFrame-side searches of LocalMapping / LoopClosing (ref: src/ORBmatcher.cc:290-403, 825-1326) ----
// The wrappers project the points on the host exactly as the reference's loops do (cv::gemm arithmetic: double
// accumulation, one rounding), hand the windows to the device and apply the map updates in the reference's order.

// out = float(alpha * R) (or R transposed): what OpenCV materialises for s*R, R/s and (1/s)*R.t()
void scale3(const cv::Mat &R, double alpha, bool transpose, cv::Mat &out)
{
    out = cv::Mat(3, 3, CV_32F);
    for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++)
            out.at<float>(r, k) = (float)(alpha * (double)(transpose ? R.at<float>(k, r) : R.at<float>(r, k)));
}

// Scw -> Rcw, tcw, Ow (ref: :299-303, :989-993)
void decompose_sim3(const cv::Mat &Scw, cv::Mat &Rcw, float tcw[3], float Ow[3])
{
    double dot = 0;
    for (int k = 0; k < 3; k++) dot += (double)Scw.at<float>(0, k) * (double)Scw.at<float>(0, k);
    const float scw = sqrt(dot);
    const cv::Mat sRcw = Scw.rowRange(0,3).colRange(0,3);
    scale3(sRcw, 1.0 / scw, false, Rcw);
    for (int r = 0; r < 3; r++) tcw[r] = (float)((double)Scw.at<float>(r, 3) * (1.0 / scw));
    affine3(Rcw, tcw, NULL, Ow, true, -1.0);                   // Ow = -Rcw.t()*tcw
}

inline float norm3(const float a[3])                           // cv::norm of a float vector sums the squares in double
{
    double sq = 0;
    for (int k = 0; k < 3; k++) sq += (double)a[k]*(double)a[k];
    return std::sqrt(sq);
}

// The window of one point in pKF from its camera coordinates (ref: :851-893, :330-369, :1163-1192).  PO != NULL adds
// the viewing-angle test against the point's normal.  Returns false where the reference's loop says continue.
bool kf_window(KeyFrame *pKF, MapPoint *pMP, const float p3Dc[3], const float *PO, float dist3D, float th, bool bf,
               orbhip_proj_query &e, uint8_t *qdesc)
{
    if(p3Dc[2]<0.0f)
        return false;
    const float invz = 1.0/p3Dc[2];
    const float x = p3Dc[0]*invz;
    const float y = p3Dc[1]*invz;
    const float u = pKF->fx*x+pKF->cx;
    const float v = pKF->fy*y+pKF->cy;
    if(!pKF->IsInImage(u,v))
        return false;
    if (dist3D<pMP->GetMinDistanceInvariance() || dist3D>pMP->GetMaxDistanceInvariance())
        return false;
    if(PO)
    {
        const cv::Mat Pn = pMP->GetNormal();
        double dot = 0;
        for (int k = 0; k < 3; k++) dot += (double)PO[k]*(double)Pn.at<float>(k, 0);
        if(dot<0.5*dist3D)
            return false;
    }
    const int level = pMP->PredictScale(dist3D, pKF);
    e.u = u;
    e.v = v;
    e.radius = th*pKF->mvScaleFactors[level];
    e.proj_xr = bf ? u-pKF->mbf*invz : 0.f;
    e.min_level = level-1;
    e.max_level = level;
    e.flags = ORBHIP_Q_ACTIVE | ORBHIP_Q_OBSERVED;
    const cv::Mat dsc(pMP->GetDescriptor());
    memcpy(qdesc, dsc.ptr(0), 32);
    return true;
}

void run_window_best(KeyFrame *pKF, const vector<orbhip_proj_query> &q, const vector<uint8_t> &qdesc, bool gate,
                     vector<int32_t> &bestIdx, vector<int32_t> &bestDist)
{
    const int n = (int)pKF->mvKeysUn.size(), nq = (int)q.size();
    bestIdx.assign(nq, -1);
    bestDist.assign(nq, 256);
    if (n == 0 || nq == 0) return;
    if (use_sets() && ensure_set(pKF)) {
        const int rc = orbhip_window_best_set(
            tls.get(), KF_KEY | (uint64_t)(pKF->mnId + 1), (gate && (int)pKF->mvuRight.size() == n) ? pKF->mvuRight.data() : NULL,
            gate ? pKF->mvInvLevelSigma2.data() : NULL, gate ? (int)pKF->mvInvLevelSigma2.size() : 0, q.data(), qdesc.data(), nq,
            bestIdx.data(), bestDist.data());
        if (rc != ORBHIP_OK) {
            hipdetail::Fail("ORBmatcher::Fuse / SearchBySim3 (KeyFrame window search)", orbhip_last_error(tls.get()));
            bestIdx.assign(nq, -1);
            bestDist.assign(nq, 256);
        }
        return;
    }
    const vector<uint8_t> d = contiguous(pKF->mDescriptors);
    const int rc = orbhip_window_best(
        tls.get(), reinterpret_cast<const orbhip_keypoint *>(pKF->mvKeysUn.data()), d.data(), n,
        (gate && (int)pKF->mvuRight.size() == n) ? pKF->mvuRight.data() : NULL, gate ? pKF->mvInvLevelSigma2.data() : NULL,
        gate ? (int)pKF->mvInvLevelSigma2.size() : 0, pKF->mnMinX, pKF->mnMinY, pKF->mfGridElementWidthInv,
        pKF->mfGridElementHeightInv, q.data(), qdesc.data(), nq, bestIdx.data(), bestDist.data());
    if (rc != ORBHIP_OK) {
        hipdetail::Fail("ORBmatcher::Fuse / SearchBySim3 (KeyFrame window search)", orbhip_last_error(tls.get()));
        bestIdx.assign(nq, -1);      // no point finds a feature: the callers fuse / match nothing
        bestDist.assign(nq, 256);
    }
}
}  // namespace

int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th)
{
    // ref: src/ORBmatcher.cc:290-403.  Best only on levels [predicted-1, predicted], a feature that holds a match is
    // closed (:372-373), threshold TH_LOW: the frame-side search with every point closing its feature.
    cv::Mat Rcw;
    float tcw[3], Ow[3];
    decompose_sim3(Scw, Rcw, tcw, Ow);

    set<MapPoint*> known;
    for (size_t i = 0; i < vpMatched.size(); i++)
        if (vpMatched[i]) known.insert(vpMatched[i]);

    const int nq = (int)vpPoints.size(), n = (int)pKF->mvKeysUn.size();
    vector<orbhip_proj_query> q(nq);
    vector<uint8_t> qdesc((size_t)nq * 32, 0);
    memset(q.data(), 0, sizeof(orbhip_proj_query) * (size_t)nq);
    for (int k = 0; k < nq; k++) {
        MapPoint *pMP = vpPoints[k];
        if (!pMP || pMP->isBad() || known.count(pMP)) continue;
        float xw[3], p3Dc[3], PO[3];
        world_point(pMP, xw);
        affine3(Rcw, xw, tcw, p3Dc);
        for (int k = 0; k < 3; k++) PO[k] = xw[k]-Ow[k];
        kf_window(pKF, pMP, p3Dc, PO, norm3(PO), (float)th, false, q[k], &qdesc[(size_t)k * 32]);
    }
    if (n == 0 || nq == 0) return 0;
    vector<uint8_t> occupied(n, 0);
    for (int i = 0; i < n && i < (int)vpMatched.size(); i++) occupied[i] = vpMatched[i] ? 1 : 0;
    const vector<uint8_t> d = contiguous(pKF->mDescriptors);
    vector<int32_t> match(n);
    int found = 0;
    const int rc = orbhip_search_by_projection(
        tls.get(), reinterpret_cast<const orbhip_keypoint *>(pKF->mvKeysUn.data()), d.data(), n, NULL, occupied.data(),
        pKF->mnMinX, pKF->mnMinY, pKF->mfGridElementWidthInv, pKF->mfGridElementHeightInv, q.data(), qdesc.data(), nq, 0,
        mfNNratio, 0, TH_LOW, match.data(), &found);
    if (rc != ORBHIP_OK) return hipdetail::Fail("ORBmatcher::SearchByProjection", orbhip_last_error(tls.get())), 0;
    for (int i = 0; i < n; i++)
        if (match[i] >= 0) vpMatched[i] = vpPoints[match[i]];
    return found;
}

int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint *> &vpMapPoints, const float th)
{
    // ref: src/ORBmatcher.cc:825-975.  The window of a point does not depend on the points before it; whether it is
    // skipped (isBad / IsInKeyFrame, :847-848) and what its feature holds (:953) do, so those are read in the loop
    // that applies the results, in the reference's order.
    const cv::Mat Rcw = pKF->GetRotation();
    const cv::Mat tcwM = pKF->GetTranslation();
    const cv::Mat OwM = pKF->GetCameraCenter();
    float tcw[3], Ow[3];
    for (int r = 0; r < 3; r++) { tcw[r] = tcwM.at<float>(r, 0); Ow[r] = OwM.at<float>(r, 0); }

    const int npts = (int)vpMapPoints.size();
    vector<orbhip_proj_query> q(npts);
    vector<uint8_t> qdesc((size_t)npts * 32, 0);
    memset(q.data(), 0, sizeof(orbhip_proj_query) * (size_t)npts);
    for (int i = 0; i < npts; i++) {
        MapPoint *pMP = vpMapPoints[i];
        if (!pMP || pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        float xw[3], p3Dc[3], PO[3];
        world_point(pMP, xw);
        affine3(Rcw, xw, tcw, p3Dc);
        for (int k = 0; k < 3; k++) PO[k] = xw[k]-Ow[k];
        kf_window(pKF, pMP, p3Dc, PO, norm3(PO), th, true, q[i], &qdesc[(size_t)i * 32]);
    }
    vector<int32_t> bestIdx, bestDist;
    run_window_best(pKF, q, qdesc, true, bestIdx, bestDist);

    // The device search is done; the map edits follow on the host, point by point in the caller's order, because an edit can
    // change what a later point of the same call sees: a point replaced a moment ago is bad now, a feature claimed a moment
    // ago holds a point now.  Both are therefore looked at here, not before the search (ref: :847-848 and :951-972).
    int fused = 0;
    for (int i = 0; i < npts; i++) {
        MapPoint *cand = vpMapPoints[i];
        const bool usable = cand && !cand->isBad() && !cand->IsInKeyFrame(pKF);
        if (!usable || bestIdx[i] < 0 || bestDist[i] > TH_LOW) continue;
        const int feat = bestIdx[i];
        if (MapPoint *held = pKF->GetMapPoint(feat)) {
            // the feature has a point already: the two are one landmark, the one seen from fewer key frames gives way
            if (!held->isBad()) {
                if (held->Observations() > cand->Observations()) cand->Replace(held);
                else held->Replace(cand);
            }
        } else {
            // a free feature: the candidate gains an observation and the key frame a point
            cand->AddObservation(pKF, feat);
            pKF->AddMapPoint(cand, feat);
        }
        fused++;      // (counted also when the held point was bad and nothing changed, as in the reference)
    }
    return fused;
}

int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint *> &vpPoints, float th, vector<MapPoint *> &vpReplacePoint)
{
    // ref: src/ORBmatcher.cc:977-1100
    cv::Mat Rcw;
    float tcw[3], Ow[3];
    decompose_sim3(Scw, Rcw, tcw, Ow);

    // points the key frame observes already are left out of the search (:995-996)
    const set<MapPoint*> held(pKF->GetMapPoints());

    const int ncand = (int)vpPoints.size();
    vector<orbhip_proj_query> q(ncand);
    vector<uint8_t> qdesc((size_t)ncand * 32, 0);
    memset(q.data(), 0, sizeof(orbhip_proj_query) * (size_t)ncand);
    for (int k = 0; k < ncand; k++) {
        MapPoint *pMP = vpPoints[k];
        if (!pMP || pMP->isBad() || held.count(pMP)) continue;
        float xw[3], p3Dc[3], PO[3];
        world_point(pMP, xw);
        affine3(Rcw, xw, tcw, p3Dc);
        for (int k = 0; k < 3; k++) PO[k] = xw[k]-Ow[k];
        kf_window(pKF, pMP, p3Dc, PO, norm3(PO), th, false, q[k], &qdesc[(size_t)k * 32]);
    }
    vector<int32_t> bestIdx, bestDist;
    run_window_best(pKF, q, qdesc, false, bestIdx, bestDist);

    // Loop closing does not replace points here (the caller does, under the map mutex): a feature that holds a good point is
    // reported through vpReplacePoint, a free one takes the candidate at once (ref: :1079-1096).
    int fused = 0;
    for (int i = 0; i < ncand; i++) {
        const int feat = bestIdx[i];
        if (feat < 0 || bestDist[i] > TH_LOW) continue;
        MapPoint *held = pKF->GetMapPoint(feat);
        if (!held) {
            vpPoints[i]->AddObservation(pKF, feat);
            pKF->AddMapPoint(vpPoints[i], feat);
        } else if (!held->isBad()) {
            vpReplacePoint[i] = held;
        }
        fused++;
    }
    return fused;
}

int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12,
                             const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th)
{
    // ref: src/ORBmatcher.cc:1102-1326.  Poses of the two key frames (world -> camera) and the similarity between the cameras in
    // both directions; the two projection searches run on the device, the mutual-agreement pass on the host.
    const cv::Mat R1w(pKF1->GetRotation()), t1wM(pKF1->GetTranslation());
    const cv::Mat R2w(pKF2->GetRotation()), t2wM(pKF2->GetTranslation());

    cv::Mat sR12, sR21;
    scale3(R12, (double)s12, false, sR12);                     // s12*R12
    scale3(R12, 1.0/s12, true, sR21);                          // (1.0/s12)*R12.t()
    float t1w[3], t2w[3], t12v[3], t21[3];
    for (int r = 0; r < 3; r++) { t1w[r] = t1wM.at<float>(r, 0); t2w[r] = t2wM.at<float>(r, 0); t12v[r] = t12.at<float>(r, 0); }
    affine3(sR21, t12v, NULL, t21, false, -1.0);               // t21 = -sR21*t12

    const vector<MapPoint*> points1(pKF1->GetMapPointMatches()), points2(pKF2->GetMapPointMatches());
    const int N1 = (int)points1.size(), N2 = (int)points2.size();

    // what the caller has matched already stays out of both searches (:1136-1151)
    vector<bool> taken1(N1), taken2(N2);
    for (int i = 0; i < N1; i++) {
        MapPoint *known = vpMatches12[i];
        if (!known) continue;
        taken1[i] = true;
        const int at2 = known->GetIndexInKeyFrame(pKF2);
        if (at2 >= 0 && at2 < N2) taken2[at2] = true;
    }

    // direction 0: the points of key frame 1 into key frame 2; direction 1: the other way round
    vector<int32_t> vnMatch1, vnMatch2, dist1, dist2;
    for (int dir = 0; dir < 2; dir++)
    {
        const vector<MapPoint*> &src = dir == 0 ? points1 : points2;
        const vector<bool> &taken = dir == 0 ? taken1 : taken2;
        KeyFrame *pKFdst = dir == 0 ? pKF2 : pKF1;
        const int N = (int)src.size();
        vector<orbhip_proj_query> q(N);
        vector<uint8_t> qdesc((size_t)N * 32, 0);
        memset(q.data(), 0, sizeof(orbhip_proj_query) * (size_t)N);
        for (int i = 0; i < N; i++) {
            MapPoint *pMP = src[i];
            if (!pMP || taken[i] || pMP->isBad()) continue;
            float xw[3], pa[3], pb[3];
            world_point(pMP, xw);
            if (dir == 0) {
                affine3(R1w, xw, t1w, pa);                     // p3Dc1 = R1w*p3Dw + t1w
                affine3(sR21, pa, t21, pb);                    // p3Dc2 = sR21*p3Dc1 + t21
            } else {
                affine3(R2w, xw, t2w, pa);                     // p3Dc2 = R2w*p3Dw + t2w
                affine3(sR12, pa, t12v, pb);                   // p3Dc1 = sR12*p3Dc2 + t12
            }
            kf_window(pKFdst, pMP, pb, NULL, norm3(pb), th, false, q[i], &qdesc[(size_t)i * 32]);
        }
        run_window_best(pKFdst, q, qdesc, false, dir == 0 ? vnMatch1 : vnMatch2, dir == 0 ? dist1 : dist2);
    }
    for(int i1=0; i1<N1; i1++) if(dist1[i1]>TH_HIGH) vnMatch1[i1] = -1;
    for(int i2=0; i2<N2; i2++) if(dist2[i2]>TH_HIGH) vnMatch2[i2] = -1;

    // a match counts when the two directions name each other (:1306-1323)
    int found = 0;
    for (int i1 = 0; i1 < N1; i1++) {
        const int i2 = vnMatch1[i1];
        if (i2 >= 0 && vnMatch2[i2] == i1) {
            vpMatches12[i1] = points2[i2];
            found++;
        }
    }
    return found;
}

int ComputeDistinctiveDescriptors(const vector<MapPoint*> &vpMapPoints)
{
    // ref: src/MapPoint.cc:283-349, gathered for every point first
    vector<uint8_t> desc;
    vector<int32_t> off(1, 0);
    vector<MapPoint*> pts;
    typedef map<KeyFrame*,size_t> Seen;
    for (size_t i = 0; i < vpMapPoints.size(); i++) {
        MapPoint *pt = vpMapPoints[i];
        if (!pt || pt->isBad()) continue;
        const Seen seen(pt->GetObservations());
        const size_t before = desc.size();
        for (Seen::const_iterator it = seen.begin(); it != seen.end(); ++it) {
            if (it->first->isBad()) continue;                  // ref: :299-300
            const uint8_t *row = it->first->mDescriptors.ptr((int)it->second);
            desc.insert(desc.end(), row, row+32);
        }
        if (desc.size() == before) continue;                   // no observation in a good key frame (:290, :305)
        off.push_back((int32_t)(desc.size()/32));
        pts.push_back(pt);
    }
    const int P = (int)pts.size();
    if (P == 0) return 0;
    vector<int32_t> best(P);
    const int rc = orbhip_distinctive_descriptors(tls.get(), desc.data(), off.data(), P, best.data(), NULL);
    if (rc != ORBHIP_OK) return hipdetail::Fail("ComputeDistinctiveDescriptors", orbhip_last_error(tls.get())), 0;   // descriptors stay as they were
    for (int p = 0; p < P; p++)
    {
        cv::Mat d(1, 32, CV_8U);
        memcpy(d.ptr(0), &desc[(size_t)(off[p]+best[p])*32], 32);
        pts[p]->SetDescriptor(d);
    }
    return P;
}

int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12,
                                       vector<pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo)
{
    // ref: src/ORBmatcher.cc:657-827.  Epipole in the second image (:664-671) on the host; the node-grouped search with
    // the epipolar tests and the rotation histogram in one call.
    const cv::Mat Cw(pKF1->GetCameraCenter()), R2w(pKF2->GetRotation()), t2w(pKF2->GetTranslation());
    const float cw[3] = {Cw.at<float>(0, 0), Cw.at<float>(1, 0), Cw.at<float>(2, 0)};
    const float t2[3] = {t2w.at<float>(0, 0), t2w.at<float>(1, 0), t2w.at<float>(2, 0)};
    float C2[3];
    affine3(R2w, cw, t2, C2);                                  // C2 = R2w*Cw+t2w
    const float invz = 1.0f/C2[2];
    const float ex =pKF2->fx*C2[0]*invz+pKF2->cx;
    const float ey =pKF2->fy*C2[1]*invz+pKF2->cy;

    const int n1 = pKF1->N, n2 = pKF2->N;
    vMatchedPairs.clear();
    if(n1==0 || n2==0)
        return 0;
    vector<uint8_t> skip1(n1), skip2(n2);
    for(int i=0; i<n1; i++) skip1[i] = pKF1->GetMapPoint(i) ? 1 : 0;     // ref: :702-705
    for(int i=0; i<n2; i++) skip2[i] = pKF2->GetMapPoint(i) ? 1 : 0;     // ref: :727-731
    const Csr f1 = flatten(pKF1->mFeatVec), f2 = flatten(pKF2->mFeatVec);
    float F[9];
    for(int r=0; r<3; r++)
        for(int c=0; c<3; c++)
            F[3*r+c] = F12.at<float>(r,c);
    const vector<uint8_t> d1 = contiguous(pKF1->mDescriptors), d2 = contiguous(pKF2->mDescriptors);
    vector<int> vMatches12(n1,-1);
    int found=0;
    const int rc = orbhip_search_for_triangulation(
        tls.get(), reinterpret_cast<const orbhip_keypoint *>(pKF1->mvKeysUn.data()), d1.data(), n1, skip1.data(),
        (int)pKF1->mvuRight.size()==n1 ? pKF1->mvuRight.data() : NULL, f1.node.data(), f1.off.data(), f1.idx.data(), (int)f1.node.size(),
        reinterpret_cast<const orbhip_keypoint *>(pKF2->mvKeysUn.data()), d2.data(), n2, skip2.data(),
        (int)pKF2->mvuRight.size()==n2 ? pKF2->mvuRight.data() : NULL, f2.node.data(), f2.off.data(), f2.idx.data(), (int)f2.node.size(),
        F, ex, ey, pKF2->mvScaleFactors.data(), pKF2->mvLevelSigma2.data(), (int)pKF2->mvScaleFactors.size(),
        bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0, vMatches12.data(), &found);
    if(rc != ORBHIP_OK)
        return hipdetail::Fail("ORBmatcher::SearchForTriangulation", orbhip_last_error(tls.get())), 0;

    vMatchedPairs.reserve(found);
    for (int i = 0; i < n1; i++)
        if (vMatches12[i] >= 0) vMatchedPairs.push_back(pair<size_t, size_t>(i, vMatches12[i]));
    return found;
}

int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize)
{
    // ref: src/ORBmatcher.cc:405-520.  cv::KeyPoint is the 28-byte record of orbhip_keypoint, cv::Point2f two floats.
    const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
    vnMatches12 = vector<int>(n1,-1);
    if(n1==0 || n2==0)
        return 0;
    if((int)vbPrevMatched.size()<n1)
        return hipdetail::Fail("ORBmatcher::SearchForInitialization", "vbPrevMatched is shorter than F1.mvKeysUn"), 0;
    int found=0;
    const int rc = orbhip_search_for_initialization(tls.get(), (const orbhip_keypoint *)F1.mvKeysUn.data(), F1.mDescriptors.ptr(0), n1,
                                                    (const orbhip_keypoint *)F2.mvKeysUn.data(), F2.mDescriptors.ptr(0), n2,
                                                    Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv,
                                                    Frame::mfGridElementHeightInv, (float *)vbPrevMatched.data(), windowSize,
                                                    mfNNratio, mbCheckOrientation ? 1 : 0, vnMatches12.data(), &found);
    if(rc != ORBHIP_OK)
    {
        vnMatches12.assign(n1,-1);
        return hipdetail::Fail("ORBmatcher::SearchForInitialization", orbhip_last_error(tls.get())), 0;
    }
    return found;
}

// ---- KeyFrame grid twin (ref: src/KeyFrame.cc:55-89, :1138-1177) ----
void KeyFrame::CopyGridFrom(const Frame &F)
{
    mnGridCols = FRAME_GRID_COLS;
    mnGridRows = FRAME_GRID_ROWS;
    mfGridElementWidthInv = Frame::mfGridElementWidthInv;
    mfGridElementHeightInv = Frame::mfGridElementHeightInv;
    mnMinX = Frame::mnMinX; mnMinY = Frame::mnMinY; mnMaxX = Frame::mnMaxX; mnMaxY = Frame::mnMaxY;
    mGrid.assign(mnGridCols, vector<vector<size_t> >(mnGridRows));
    for (int c = 0; c < mnGridCols; c++)
        for (int r = 0; r < mnGridRows; r++) mGrid[c][r] = F.mGrid[c][r];
}

vector<size_t> KeyFrame::GetFeaturesInArea(const float &x, const float &y, const float &r) const
{
    // the window query of liborbhip on mvKeysUn (the grid it builds from them is the one mGrid holds)
    vector<size_t> inside;
    const int n = (int)mvKeysUn.size();
    if (n == 0) return inside;
    orbhip_proj_query q = {x, y, r, 0.f, -1, -1, 0.f, ORBHIP_Q_ACTIVE};
    vector<int32_t> idx(n);
    int32_t off[2] = {0, 0};
    if(orbhip_features_in_area(tls.get(), reinterpret_cast<const orbhip_keypoint *>(mvKeysUn.data()), n, mnMinX, mnMinY,
                               mfGridElementWidthInv, mfGridElementHeightInv, &q, 1, off, idx.data(), n) != ORBHIP_OK)
        return hipdetail::Fail("KeyFrame::GetFeaturesInArea", orbhip_last_error(tls.get())), inside;
    inside.assign(idx.begin(), idx.begin() + off[1]);
    return inside;
}

void ORBmatcher::ComputeThreeMaxima(vector<int>* histo, const int L, int &ind1, int &ind2, int &ind3)
{
    // ref: src/ORBmatcher.cc:1629-1670
    int best[3] = {0, 0, 0};
    int where[3] = {-1, -1, -1};
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        int slot = s > best[0] ? 0 : (s > best[1] ? 1 : (s > best[2] ? 2 : 3));
        for (int k = 2; k > slot; k--) {
            best[k] = best[k - 1];
            where[k] = where[k - 1];
        }
        if (slot < 3) {
            best[slot] = s;
            where[slot] = i;
        }
    }
    if ((float)best[1] < 0.1f * (float)best[0]) {
        where[1] = -1;
        where[2] = -1;
    } else if ((float)best[2] < 0.1f * (float)best[0]) {
        where[2] = -1;
    }
    ind1 = where[0];
    ind2 = where[1];
    ind3 = where[2];
}

// Bit set count over the 256-bit XOR (ref: src/ORBmatcher.cc:1675-1691)
int ORBmatcher::DescriptorDistance(const cv::Mat &a, const cv::Mat &b)
{
    const uint32_t *pa = a.ptr<uint32_t>();
    const uint32_t *pb = b.ptr<uint32_t>();
    int dist = 0;
    for (int i = 0; i < 8; i++) dist += __builtin_popcount(pa[i] ^ pb[i]);
    return dist;
}

} //namespace ORB_SLAM
