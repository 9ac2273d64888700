// test_fuse_dropin.cpp -- the KeyFrame-side searches of LocalMapping / LoopClosing through the drop-in classes:
// ORBmatcher::Fuse (both forms), SearchByProjection(KeyFrame*, Scw, ...) and SearchBySim3 (ref: src/ORBmatcher.cc:290-403,
// 825-1326; callers src/LocalMapping.cc SearchInNeighbors, src/LoopClosing.cc ComputeSim3 / SearchAndFuse).
// Two identical worlds are built; one runs the drop-in (HIP), the other the loops below, which restate the routines on
// the host with KeyFrame::GetFeaturesInArea and ORBmatcher::DescriptorDistance; the resulting map states must be equal.
// Self-checking: exit code 0 and one "ok" line per routine.
#include "host_restate.h"

struct Shared {                    // one extraction, shared by every key frame of both worlds
    vector<cv::KeyPoint> keys;
    cv::Mat desc;
    vector<float> scale, sigma2, invSigma2;
    Frame F;                       // holds the grid
    int w, h;
};

struct World {
    KeyFrame kf[3];                // kf[0], kf[1]: the pair; kf[2]: a third observer
    vector<MapPoint> pts;
    vector<MapPoint *> cand;       // fuse candidates (points of kf[0] and of kf[2])
};

static void setupKF(KeyFrame &K, const Shared &S, const cv::Mat &Tcw, float stereoShare)
{
    K.mvKeys = S.keys; K.mvKeysUn = S.keys; K.mDescriptors = S.desc; K.N = (int)S.keys.size();
    K.fx = 517.3f; K.fy = 516.5f; K.cx = 318.6f; K.cy = 255.3f; K.mbf = 40.f;
    K.mvScaleFactors = S.scale; K.mvLevelSigma2 = S.sigma2; K.mvInvLevelSigma2 = S.invSigma2;
    K.mnScaleLevels = 8; K.mfScaleFactor = 1.2f; K.mfLogScaleFactor = logf(1.2f);
    K.Tcw = Tcw.clone();
    K.Ow = cv::Mat(3, 1, CV_32F);
    float t[3] = {Tcw.at<float>(0, 3), Tcw.at<float>(1, 3), Tcw.at<float>(2, 3)}, o[3];
    mul3(Tcw, t, NULL, o, true, -1.0);
    for (int r = 0; r < 3; r++) K.Ow.at<float>(r, 0) = o[r];
    K.mvuRight.assign(K.N, -1.f);
    for (int i = 0; i < K.N; i++)
        if (urand() < stereoShare) K.mvuRight[i] = S.keys[i].pt.x - (float)(2 + 30 * urand());
    K.mvpMapPoints.assign(K.N, static_cast<MapPoint *>(NULL));
    K.CopyGridFrom(S.F);
}

// a point seen by K at feature i: back-projected at a random depth, descriptor of the feature with a few bits flipped
static void makePoint(MapPoint &p, KeyFrame &K, int i, const Shared &S, float pixelNoise, int flips)
{
    const cv::KeyPoint &kp = K.mvKeysUn[i];
    const float z = (float)(2 + 8 * urand());
    const float xc[3] = {(kp.pt.x + pixelNoise * (float)(urand() - 0.5) - K.cx) / K.fx * z,
                         (kp.pt.y + pixelNoise * (float)(urand() - 0.5) - K.cy) / K.fy * z, z};
    float t[3] = {K.Tcw.at<float>(0, 3), K.Tcw.at<float>(1, 3), K.Tcw.at<float>(2, 3)}, d[3], xw[3];
    for (int k = 0; k < 3; k++) d[k] = xc[k] - t[k];
    mul3(K.Tcw, d, NULL, xw, true);                             // Xw = R^T (Xc - t)
    p.mWorldPos = cv::Mat(3, 1, CV_32F);
    p.mNormalVector = cv::Mat(3, 1, CV_32F);
    float po[3], n = 0;
    for (int k = 0; k < 3; k++) { p.mWorldPos.at<float>(k, 0) = xw[k]; po[k] = xw[k] - K.Ow.at<float>(k, 0); n += po[k] * po[k]; }
    n = sqrtf(n);
    const bool sideways = urand() < 0.04;                       // fails the viewing-angle test
    for (int k = 0; k < 3; k++) p.mNormalVector.at<float>(k, 0) = sideways ? (k == 0 ? 1.f : 0.f) : po[k] / n;
    p.mfMaxDistance = n * S.scale[kp.octave] * (float)(0.86 + 0.1 * urand());
    p.mfMinDistance = p.mfMaxDistance / S.scale[7];
    if (urand() < 0.04) p.mfMaxDistance *= 0.05f;               // out of the scale-invariance range
    p.mDescriptor = cv::Mat(1, 32, CV_8U);
    memcpy(p.mDescriptor.ptr(0), K.mDescriptors.ptr(i), 32);
    for (int f = 0; f < flips; f++) { const int b = (int)(256 * urand()); p.mDescriptor.ptr(0)[b >> 3] ^= 1 << (b & 7); }
    if (urand() < 0.03) p.SetBadFlag();
}

static void buildWorld(World &W, const Shared &S, unsigned long long seed)
{
    g_rng = seed;
    setupKF(W.kf[0], S, pose(0.f, 0.f, 0.f, 0.f, 0.f), 0.f);
    setupKF(W.kf[1], S, pose(0.0012f, -0.0015f, 0.012f, -0.007f, 0.02f), 0.5f);
    setupKF(W.kf[2], S, pose(-0.002f, 0.001f, -0.01f, 0.004f, -0.015f), 0.3f);
    const int N = W.kf[0].N;
    W.pts.assign((size_t)3 * N, MapPoint());
    int np = 0;
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < N; i++) {
            if (urand() > (k == 1 ? 0.45 : 0.7)) continue;
            MapPoint &p = W.pts[np++];
            makePoint(p, W.kf[k], i, S, k == 1 ? 0.f : 3.f, (int)(40 * urand() * urand()));
            p.AddObservation(&W.kf[k], i);
            W.kf[k].mvpMapPoints[i] = &p;
            if (k == 0 && urand() < 0.3) p.nObs += (int)(3 * urand());   // more observers elsewhere: decides who replaces whom
            if (k != 1 && !(k == 0 && urand() < 0.1)) W.cand.push_back(&p);
        }
    W.cand.push_back(static_cast<MapPoint *>(NULL));
    if (W.kf[1].mvpMapPoints[7]) W.cand.push_back(W.kf[1].mvpMapPoints[7]);   // already in the key frame -> skipped
}

// ---------------- state comparison ----------------
static long idx(const World &W, const MapPoint *p) { return p ? (long)(p - &W.pts[0]) : -1; }

static bool sameState(World &A, World &B, const char *what)
{
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < A.kf[k].N; i++)
            if (idx(A, A.kf[k].mvpMapPoints[i]) != idx(B, B.kf[k].mvpMapPoints[i])) {
                printf("%s: kf[%d].mvpMapPoints[%d] differs: %ld vs %ld\n", what, k, i, idx(A, A.kf[k].mvpMapPoints[i]), idx(B, B.kf[k].mvpMapPoints[i]));
                return false;
            }
    for (size_t i = 0; i < A.pts.size(); i++) {
        MapPoint &a = A.pts[i], &b = B.pts[i];
        bool same = a.isBad() == b.isBad() && a.Observations() == b.Observations() && idx(A, a.GetReplaced()) == idx(B, b.GetReplaced()) &&
                    a.mObservations.size() == b.mObservations.size();
        for (int k = 0; same && k < 3; k++) same = a.GetIndexInKeyFrame(&A.kf[k]) == b.GetIndexInKeyFrame(&B.kf[k]);
        if (!same) { printf("%s: point %zu differs\n", what, i); return false; }
    }
    return true;
}

static bool sameVec(const World &A, const vector<MapPoint *> &a, const World &B, const vector<MapPoint *> &b, const char *what)
{
    if (a.size() != b.size()) { printf("%s: sizes differ\n", what); return false; }
    for (size_t i = 0; i < a.size(); i++)
        if (idx(A, a[i]) != idx(B, b[i])) { printf("%s: entry %zu differs: %ld vs %ld\n", what, i, idx(A, a[i]), idx(B, b[i])); return false; }
    return true;
}

static int count(const vector<MapPoint *> &v) { int n = 0; for (size_t i = 0; i < v.size(); i++) n += v[i] != NULL; return n; }

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s w h nfeatures frame.raw [seed]\n", argv[0]); return 2; }
    const unsigned long long seed0 = argc > 5 ? strtoull(argv[5], NULL, 10) : 0ull;
    Shared S;
    S.w = atoi(argv[1]); S.h = atoi(argv[2]);
    const int nf = atoi(argv[3]);
    vector<unsigned char> raw((size_t)S.w * S.h);
    FILE *f = fopen(argv[4], "rb");
    if (!f || fread(raw.data(), 1, raw.size(), f) != raw.size()) { perror(argv[4]); return 2; }
    fclose(f);

    ORBextractor ex(nf, 1.2f, 8, 20, 7);
    ex.SetPyramidDownload(false);
    cv::Mat im(S.h, S.w, CV_8UC1, raw.data());
    ex(im, cv::Mat(), S.keys, S.desc);
    S.scale = ex.GetScaleFactors(); S.sigma2 = ex.GetScaleSigmaSquares(); S.invSigma2 = ex.GetInverseScaleSigmaSquares();
    Frame::mnMinX = 0; Frame::mnMaxX = (float)S.w; Frame::mnMinY = 0; Frame::mnMaxY = (float)S.h;
    Frame::mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / (Frame::mnMaxX - Frame::mnMinX);
    Frame::mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / (Frame::mnMaxY - Frame::mnMinY);
    S.F.mvKeys = S.keys; S.F.mvKeysUn = S.keys; S.F.N = (int)S.keys.size();
    S.F.mpORBextractorLeft = &ex;
    S.F.AssignFeaturesToGrid();
    printf("features %d\n", S.F.N);

    int fails = 0;
    for (int round = 0; round < 3; round++) {
        const unsigned long long seed = 1234567ull + 7919ull * round + 104729ull * seed0;
        // ---- Fuse(pKF, vpMapPoints, th): LocalMapping::SearchInNeighbors ----
        {
            World A, B;
            buildWorld(A, S, seed); buildWorld(B, S, seed);
            ORBmatcher matcher;
            const float th = round == 2 ? 6.f : 3.f;
            const int na = matcher.Fuse(&A.kf[1], A.cand, th), nb = refFuse(&B.kf[1], B.cand, th);
            int replaced = 0;
            for (size_t i = 0; i < A.pts.size(); i++) replaced += A.pts[i].GetReplaced() != NULL;
            const bool ok = na == nb && sameState(A, B, "Fuse") && na > 100 && replaced > 20;
            printf("Fuse round %d: %s fused %d (ref %d) of %zu candidates, %d replaced\n", round, ok ? "ok" : "FAILED", na, nb, A.cand.size(), replaced);
            fails += !ok;
        }
        const cv::Mat T1 = pose(0.0012f, -0.0015f, 0.012f, -0.007f, 0.02f);
        cv::Mat Scw = T1.clone();
        const float s = round == 0 ? 1.f : 1.f + 0.002f * round;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 4; c++) Scw.at<float>(r, c) = s * T1.at<float>(r, c) + (c == 3 ? 0.002f * round : 0.f);
        // ---- Fuse(pKF, Scw, vpPoints, th, vpReplacePoint): LoopClosing::SearchAndFuse ----
        {
            World A, B;
            buildWorld(A, S, seed); buildWorld(B, S, seed);
            vector<MapPoint *> pa, pb;
            for (size_t i = 0; i < A.cand.size(); i++)
                if (A.cand[i]) { pa.push_back(A.cand[i]); pb.push_back(B.cand[i]); }
            vector<MapPoint *> ra(pa.size(), static_cast<MapPoint *>(NULL)), rb(pb.size(), static_cast<MapPoint *>(NULL));
            ORBmatcher matcher(0.8f);
            const int na = matcher.Fuse(&A.kf[1], Scw, pa, 4.f, ra), nb = refFuseScw(&B.kf[1], Scw, pb, 4.f, rb);
            const bool ok = na == nb && sameState(A, B, "Fuse(Scw)") && sameVec(A, ra, B, rb, "vpReplacePoint") && na > 100 && count(ra) > 20;
            printf("Fuse(Scw) round %d: %s fused %d (ref %d), %d to replace\n", round, ok ? "ok" : "FAILED", na, nb, count(ra));
            fails += !ok;
        }
        // ---- SearchByProjection(pKF, Scw, vpPoints, vpMatched, th): LoopClosing::ComputeSim3 ----
        {
            World A, B;
            buildWorld(A, S, seed); buildWorld(B, S, seed);
            vector<MapPoint *> pa, pb;
            for (size_t i = 0; i < A.cand.size(); i++)
                if (A.cand[i]) { pa.push_back(A.cand[i]); pb.push_back(B.cand[i]); }
            vector<MapPoint *> ma(A.kf[1].N, static_cast<MapPoint *>(NULL)), mb(B.kf[1].N, static_cast<MapPoint *>(NULL));
            for (int i = 0; i < A.kf[1].N; i += 5) { ma[i] = pa[(size_t)i % pa.size()]; mb[i] = pb[(size_t)i % pb.size()]; }   // found earlier
            const int before = count(ma);
            ORBmatcher matcher(0.75f, true);
            const int na = matcher.SearchByProjection(&A.kf[1], Scw, pa, ma, 10), nb = refProjScw(&B.kf[1], Scw, pb, mb, 10);
            const bool ok = na == nb && sameVec(A, ma, B, mb, "vpMatched") && na > 100 && count(ma) == before + na;
            printf("SearchByProjection(Scw) round %d: %s matches %d (ref %d)\n", round, ok ? "ok" : "FAILED", na, nb);
            fails += !ok;
        }
        // ---- SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th): LoopClosing::ComputeSim3 ----
        {
            World A, B;
            buildWorld(A, S, seed); buildWorld(B, S, seed);
            // camera 1 = kf[0] (identity), camera 2 = kf[1]: S12 maps camera-2 coordinates to camera 1
            cv::Mat R12(3, 3, CV_32F), t12(3, 1, CV_32F);
            float t[3] = {T1.at<float>(0, 3), T1.at<float>(1, 3), T1.at<float>(2, 3)}, o[3];
            mul3(T1, t, NULL, o, true, -1.0);
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) R12.at<float>(r, c) = T1.at<float>(c, r);
                t12.at<float>(r, 0) = o[r] + 0.001f * round;
            }
            vector<MapPoint *> ma(A.kf[0].N, static_cast<MapPoint *>(NULL)), mb(B.kf[0].N, static_cast<MapPoint *>(NULL));
            for (int i = 3; i < A.kf[0].N; i += 11)
                if (A.kf[1].mvpMapPoints[i]) { ma[i] = A.kf[1].mvpMapPoints[i]; mb[i] = B.kf[1].mvpMapPoints[i]; }   // from SearchByBoW
            const int before = count(ma);
            ORBmatcher matcher(0.75f, true);
            const int na = matcher.SearchBySim3(&A.kf[0], &A.kf[1], ma, s, R12, t12, 7.5f), nb = refSim3(&B.kf[0], &B.kf[1], mb, s, R12, t12, 7.5f);
            const bool ok = na == nb && sameVec(A, ma, B, mb, "vpMatches12") && na > 50 && count(ma) == before + na;
            printf("SearchBySim3 round %d: %s found %d (ref %d), %d given\n", round, ok ? "ok" : "FAILED", na, nb, before);
            fails += !ok;
        }
    }
    // ---- ComputeDistinctiveDescriptors over the points of a fused world (LocalMapping::SearchInNeighbors' last loop) ----
    {
        World A, B;
        buildWorld(A, S, 424242ull + seed0); buildWorld(B, S, 424242ull + seed0);
        ORBmatcher matcher;
        matcher.Fuse(&A.kf[1], A.cand, 3.f);                     // gives many points two or three observers
        refFuse(&B.kf[1], B.cand, 3.f);
        A.kf[2].mbBad = B.kf[2].mbBad = true;                    // a bad observer's descriptor is not used
        // the observers share one image here; make their descriptor rows differ so that the choice matters
        for (int k = 0; k < 3; k++)
            for (int i = 0; i < A.kf[k].N; i++)
                for (int f = 0; f < 3 * k; f++) {
                    const int b = (i * 31 + f * 97 + k * 7) & 255;
                    A.kf[k].mDescriptors.ptr(i)[b >> 3] ^= 1 << (b & 7);
                    B.kf[k].mDescriptors.ptr(i)[b >> 3] ^= 1 << (b & 7);
                }
        vector<MapPoint *> va, vb;
        for (size_t i = 0; i < A.pts.size(); i++) { va.push_back(&A.pts[i]); vb.push_back(&B.pts[i]); }
        va.push_back(static_cast<MapPoint *>(NULL));
        const int na = ComputeDistinctiveDescriptors(va);
        int nb = 0, multi = 0, diff = 0;
        for (size_t i = 0; i < vb.size(); i++) {                 // src/MapPoint.cc:283-349 restated
            MapPoint *pMP = vb[i];
            if (pMP->isBad()) continue;
            const std::map<KeyFrame *, size_t> obs = pMP->GetObservations();
            vector<cv::Mat> vDescriptors;
            for (std::map<KeyFrame *, size_t>::const_iterator it = obs.begin(); it != obs.end(); it++)
                if (!it->first->isBad()) vDescriptors.push_back(it->first->mDescriptors.row((int)it->second));
            if (vDescriptors.empty()) continue;
            const size_t N = vDescriptors.size();
            int BestMedian = INT_MAX, BestIdx = 0;
            for (size_t r = 0; r < N; r++) {
                vector<int> vDists(N);
                for (size_t c = 0; c < N; c++) vDists[c] = r == c ? 0 : ORBmatcher::DescriptorDistance(vDescriptors[r], vDescriptors[c]);
                std::sort(vDists.begin(), vDists.end());
                const int median = vDists[0.5 * (N - 1)];
                if (median < BestMedian) { BestMedian = median; BestIdx = (int)r; }
            }
            pMP->SetDescriptor(vDescriptors[BestIdx]);
            nb++;
            multi += N > 1;
        }
        for (size_t i = 0; i < A.pts.size(); i++) {
            const cv::Mat da = A.pts[i].GetDescriptor(), db = B.pts[i].GetDescriptor();
            if ((da.data == NULL) != (db.data == NULL) || (da.data && memcmp(da.ptr(0), db.ptr(0), 32))) diff++;
        }
        const bool ok = na == nb && diff == 0 && multi > 100;
        printf("ComputeDistinctiveDescriptors: %s %d points (ref %d), %d with several observers, %d differ\n", ok ? "ok" : "FAILED", na, nb, multi, diff);
        fails += !ok;
    }
    printf(fails ? "FAILED (%d)\n" : "all ok\n", fails);
    return fails ? 1 : 0;
}
