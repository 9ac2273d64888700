// test_threads_dropin.cpp -- ORBmatcher as the reference uses it: from THREE threads at once over the SAME key frames.
// Tracking, LocalMapping and LoopClosing each construct matchers on their own stacks and run them concurrently on shared KeyFrames
// (ref: src/System.cc:365-375 starts the three threads; src/Tracking.cc TrackReferenceKeyFrame / TrackWithMotionModel,
// src/LocalMapping.cc CreateNewMapPoints / SearchInNeighbors, src/LoopClosing.cc ComputeSim3).  The drop-in answers with one device
// context and one table of resident key frames per thread (host/ORBmatcher.cc, thread_local).  Here:
//   thread T   SearchByBoW(KeyFrame, Frame) + SearchByProjection(CurrentFrame, LastFrame)         -- Tracking
//   thread M   SearchForTriangulation(KF, KF) + Fuse(KF, points) on a map of its own             -- LocalMapping
//   thread L   SearchByBoW(KeyFrame, KeyFrame) + SearchBySim3                                     -- LoopClosing
// run `rounds` rounds each over the same 30 key frames, with at most 8 resident sets per thread (ORBmatcher::SetResidentSetLimit:
// every round evicts), and T calls ORBmatcher::DropResidentSets() in the middle of the run and every 97th round.  Every result
// must equal the host restatement of the same call (host_restate.h), computed single-threaded before the threads start.
// (Fuse edits the map; slamlite's MapPoint / KeyFrame twins have no mutexes, so M fuses in a map no other thread reads --
// the matcher's own state is what is shared: the device, the library's statics, the HIP runtime.)
// Usage: test_threads_dropin w h nfeatures frames.raw nframes vocabulary.bin [rounds]      exit code 0 and "all ok".
//
// -DORBHIP_TSAN_MOCK (tests/native/Makefile: test_threads_tsan, built with -fsanitize=thread against tests/native/mock_orbhip.cc
// instead of liborbhip.so): the same three threads and the same schedule -- eight resident sets per thread,
// DropResidentSets() mid-run -- over synthetic features, no GPU.  The mock answers every search with a deterministic function
// of the data it was handed, so "expected" is the same call made single-threaded before the threads start; ThreadSanitizer
// watches the host side (thread_local contexts, the shared limit, the statics of host/*.cc) while they run.
// Usage there: test_threads_tsan [rounds]
#include <thread>

#include "ORBVocabulary.h"
#include "host_restate.h"

struct Extraction {
    vector<cv::KeyPoint> keys;
    cv::Mat desc;
    DBoW2::BowVector bow;
    DBoW2::FeatureVector fv;
    Frame grid;                    // holds the 64 x 48 grid of these keypoints
};
struct Common {
    int w, h;
    vector<float> scale, sigma2, invSigma2;
    vector<Extraction> ex;
};

static uint64_t mix(uint64_t h, uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h * 0x100000001B3ull; }

static void setupKF(KeyFrame &K, const Common &C, int e, const cv::Mat &Tcw, float stereoShare)
{
    const Extraction &X = C.ex[e];
    K.mvKeys = X.keys; K.mvKeysUn = X.keys; K.mDescriptors = X.desc; K.N = (int)X.keys.size();
    K.mFeatVec = X.fv;
    K.fx = 517.3f; K.fy = 516.5f; K.cx = 318.6f; K.cy = 255.3f; K.mbf = 40.f;
    K.mvScaleFactors = C.scale; K.mvLevelSigma2 = C.sigma2; K.mvInvLevelSigma2 = C.invSigma2;
    K.mnScaleLevels = 8; K.mfScaleFactor = 1.2f; K.mfLogScaleFactor = logf(1.2f);
    K.Tcw = Tcw.clone();
    K.Ow = cv::Mat(3, 1, CV_32F);
    float t[3] = {Tcw.at<float>(0, 3), Tcw.at<float>(1, 3), Tcw.at<float>(2, 3)}, o[3];
    mul3(Tcw, t, NULL, o, true, -1.0);
    for (int r = 0; r < 3; r++) K.Ow.at<float>(r, 0) = o[r];
    K.mvuRight.assign(K.N, -1.f);
    for (int i = 0; i < K.N; i++)
        if (urand() < stereoShare) K.mvuRight[i] = X.keys[i].pt.x - (float)(2 + 30 * urand());
    K.mvpMapPoints.assign(K.N, static_cast<MapPoint *>(NULL));
    K.CopyGridFrom(X.grid);
}

// a point seen by K at feature i: back-projected at a random depth, the feature's descriptor with a few bits flipped
static void makePoint(MapPoint &p, KeyFrame &K, int i, const Common &C, float pixelNoise, int flips)
{
    const cv::KeyPoint &kp = K.mvKeysUn[i];
    const float z = (float)(2 + 8 * urand());
    const float xc[3] = {(kp.pt.x + pixelNoise * (float)(urand() - 0.5) - K.cx) / K.fx * z,
                         (kp.pt.y + pixelNoise * (float)(urand() - 0.5) - K.cy) / K.fy * z, z};
    float t[3] = {K.Tcw.at<float>(0, 3), K.Tcw.at<float>(1, 3), K.Tcw.at<float>(2, 3)}, d[3], xw[3];
    for (int k = 0; k < 3; k++) d[k] = xc[k] - t[k];
    mul3(K.Tcw, d, NULL, xw, true);
    p.mWorldPos = cv::Mat(3, 1, CV_32F);
    p.mNormalVector = cv::Mat(3, 1, CV_32F);
    float po[3], n = 0;
    for (int k = 0; k < 3; k++) { p.mWorldPos.at<float>(k, 0) = xw[k]; po[k] = xw[k] - K.Ow.at<float>(k, 0); n += po[k] * po[k]; }
    n = sqrtf(n);
    for (int k = 0; k < 3; k++) p.mNormalVector.at<float>(k, 0) = po[k] / n;
    p.mfMaxDistance = n * C.scale[kp.octave] * (float)(0.86 + 0.1 * urand());
    p.mfMinDistance = p.mfMaxDistance / C.scale[7];
    p.mDescriptor = cv::Mat(1, 32, CV_8U);
    memcpy(p.mDescriptor.ptr(0), K.mDescriptors.ptr(i), 32);
    for (int f = 0; f < flips; f++) { const int b = (int)(256 * urand()); p.mDescriptor.ptr(0)[b >> 3] ^= 1 << (b & 7); }
    if (urand() < 0.03) p.SetBadFlag();
}

static cv::Mat poseOf(int k)
{
    return pose(0.0011f * (float)((k * 7) % 5 - 2), -0.0013f * (float)((k * 3) % 7 - 3), 0.004f * (float)(k % 5 - 2), -0.003f * (float)(k % 3 - 1),
                0.006f * (float)((k * 5) % 7 - 3));
}

// ---- the map every thread reads: 30 key frames over the extractions, each with points on ~55 % of its features ----
struct SharedMap {
    vector<KeyFrame> kf;
    vector<MapPoint> pts;
};
static void buildShared(SharedMap &S, const Common &C, int nkf)
{
    g_rng = 0x5EEDull;
    S.kf.resize(nkf);
    size_t total = 0;
    for (int k = 0; k < nkf; k++) {
        setupKF(S.kf[k], C, k % (int)C.ex.size(), poseOf(k), k % 4 == 1 ? 0.4f : 0.f);
        total += S.kf[k].N;
    }
    S.pts.assign(total, MapPoint());
    size_t np = 0;
    for (int k = 0; k < nkf; k++)
        for (int i = 0; i < S.kf[k].N; i++) {
            if (urand() > 0.55) continue;
            MapPoint &p = S.pts[np++];
            makePoint(p, S.kf[k], i, C, 2.f, (int)(30 * urand() * urand()));
            p.AddObservation(&S.kf[k], i);
            S.kf[k].mvpMapPoints[i] = &p;
        }
}
static uint64_t hashVec(const SharedMap &S, const vector<MapPoint *> &v)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < v.size(); i++) h = mix(h, v[i] ? (uint64_t)(v[i] - &S.pts[0]) + 1 : 0);
    return h;
}

// ---- M's own map for Fuse: six key frames of one extraction, points of each are fuse candidates for the next ----
struct FuseMap {
    vector<KeyFrame> kf;
    vector<MapPoint> pts;
    vector<vector<MapPoint *> > cand;
};
static void buildFuse(FuseMap &W, const Common &C, unsigned long long seed)
{
    g_rng = seed | 1ull;
    const int e = (int)(seed % C.ex.size());
    W.kf.assign(6, KeyFrame());
    W.cand.assign(6, vector<MapPoint *>());
    for (int k = 0; k < 6; k++) setupKF(W.kf[k], C, e, poseOf((int)(seed % 11) + k), k % 2 ? 0.5f : 0.f);
    const int N = W.kf[0].N;
    W.pts.assign((size_t)6 * N, MapPoint());
    size_t np = 0;
    for (int k = 0; k < 6; k++)
        for (int i = 0; i < N; i++) {
            if (urand() > 0.6) continue;
            MapPoint &p = W.pts[np++];
            makePoint(p, W.kf[k], i, C, k % 2 ? 0.f : 3.f, (int)(40 * urand() * urand()));
            p.AddObservation(&W.kf[k], i);
            W.kf[k].mvpMapPoints[i] = &p;
            if (urand() < 0.3) p.nObs += (int)(3 * urand());
            W.cand[(k + 1) % 6].push_back(&p);
        }
    for (int k = 0; k < 6; k++) W.cand[k].push_back(static_cast<MapPoint *>(NULL));
}
static uint64_t hashFuse(FuseMap &W)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t k = 0; k < W.kf.size(); k++)
        for (int i = 0; i < W.kf[k].N; i++) { MapPoint *p = W.kf[k].mvpMapPoints[i]; h = mix(h, p ? (uint64_t)(p - &W.pts[0]) + 1 : 0); }
    for (size_t i = 0; i < W.pts.size(); i++) {
        MapPoint &p = W.pts[i];
        h = mix(h, (uint64_t)p.isBad() | ((uint64_t)p.Observations() << 1) | ((uint64_t)p.mObservations.size() << 20) |
                       ((uint64_t)(p.GetReplaced() ? p.GetReplaced() - &W.pts[0] + 1 : 0) << 32));
    }
    return h;
}

// ---- the frames T tracks: current / last pairs over the shared key frames' points ----
static void setupFrame(Frame &F, const Common &C, int e, const cv::Mat &Tcw)
{
    const Extraction &X = C.ex[e];
    F = X.grid;                                    // keypoints + grid
    F.mDescriptors = X.desc; F.mFeatVec = X.fv; F.mBowVec = X.bow;
    F.mTcw = Tcw.clone();
    F.mvScaleFactors = C.scale; F.mnScaleLevels = 8; F.mfScaleFactor = 1.2f; F.mfLogScaleFactor = logf(1.2f);
    F.mvpMapPoints.assign(F.N, static_cast<MapPoint *>(NULL));
    F.mvbOutlier.assign(F.N, false);
    F.mvuRight.assign(F.N, -1.f);
    F.mbf = 40.f; F.mb = 0.08f;
    F.mpORBextractorLeft = NULL;                   // (not the frame its extractor built last: the set is put from the host)
}

struct Res { int n; uint64_t h; };
static bool same(const Res &a, const Res &b) { return a.n == b.n && a.h == b.h; }

struct Plan {
    int rounds, nkf;
    const Common *C;
    SharedMap *S;
};
static int kfB(int r, int nkf) { int b = (r * 7 + 3) % nkf; return b == r % nkf ? (b + 1) % nkf : b; }

// geometry between two key frames (any fixed matrices do: both sides of a comparison get the same ones)
static void relPose(KeyFrame &K1, KeyFrame &K2, cv::Mat &R12, cv::Mat &t12)
{
    // camera-2 coordinates -> camera-1 coordinates: x1 = R1 R2^T (x2 - t2) + t1
    R12 = cv::Mat(3, 3, CV_32F); t12 = cv::Mat(3, 1, CV_32F);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += (double)K1.Tcw.at<float>(r, k) * (double)K2.Tcw.at<float>(c, k);
            R12.at<float>(r, c) = (float)s;
        }
    for (int r = 0; r < 3; r++) {
        double s = K1.Tcw.at<float>(r, 3);
        for (int c = 0; c < 3; c++) s -= (double)R12.at<float>(r, c) * (double)K2.Tcw.at<float>(c, 3);
        t12.at<float>(r, 0) = (float)s;
    }
}
static cv::Mat fundamental(KeyFrame &K1, KeyFrame &K2)
{
    cv::Mat R12, t12;
    relPose(K1, K2, R12, t12);
    const double tx[9] = {0, -t12.at<float>(2, 0), t12.at<float>(1, 0), t12.at<float>(2, 0), 0, -t12.at<float>(0, 0), -t12.at<float>(1, 0), t12.at<float>(0, 0), 0};
    double E[9], KiT[9] = {1.0 / K1.fx, 0, 0, 0, 1.0 / K1.fy, 0, -K1.cx / K1.fx, -K1.cy / K1.fy, 1}, Ki[9] = {1.0 / K2.fx, 0, -K2.cx / K2.fx, 0, 1.0 / K2.fy, -K2.cy / K2.fy, 0, 0, 1}, A[9], Fm[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { E[r * 3 + c] = 0; for (int k = 0; k < 3; k++) E[r * 3 + c] += tx[r * 3 + k] * (double)R12.at<float>(k, c); }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { A[r * 3 + c] = 0; for (int k = 0; k < 3; k++) A[r * 3 + c] += KiT[r * 3 + k] * E[k * 3 + c]; }
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { Fm[r * 3 + c] = 0; for (int k = 0; k < 3; k++) Fm[r * 3 + c] += A[r * 3 + k] * Ki[k * 3 + c]; }
    cv::Mat F(3, 3, CV_32F);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) F.at<float>(r, c) = (float)Fm[r * 3 + c];
    return F;
}

// one round of each role; hip = the drop-in, otherwise the host restatement
static void roundT(const Plan &P, int r, bool hip, Res out[2])
{
    SharedMap &S = *P.S;
    const int a = r % P.nkf, e = a % (int)P.C->ex.size();
    KeyFrame *pKF = &S.kf[a];
    Frame cur, last;
    setupFrame(cur, *P.C, e, poseOf(a + 1 + r % 3));
    cur.mnId = 1000 + (unsigned long)(r % 10);             // ten frame ids go round: a frame's set is met again ... or evicted by then
    setupFrame(last, *P.C, e, pKF->Tcw);
    last.mnId = 2000 + (unsigned long)(r % 10);
    last.mvpMapPoints = pKF->mvpMapPoints;
    for (int i = 0; i < last.N; i += 13) last.mvbOutlier[i] = true;
    vector<MapPoint *> m;
    ORBmatcher bow(0.7f, true), proj(0.9f, true);
    out[0].n = hip ? bow.SearchByBoW(pKF, cur, m) : refSearchByBoW(pKF, cur, m, 0.7f, true);
    out[0].h = hashVec(S, m);
    const float th = r % 2 ? 15.f : 7.f;
    out[1].n = hip ? proj.SearchByProjection(cur, last, th, true) : refSearchByProjection(cur, last, th, true, true);
    out[1].h = hashVec(S, cur.mvpMapPoints);
}
static void roundM(const Plan &P, int r, bool hip, FuseMap &W, Res out[2])
{
    SharedMap &S = *P.S;
    const int a = r % P.nkf, b = kfB(r, P.nkf);
    const cv::Mat F12 = fundamental(S.kf[a], S.kf[b]);
    vector<std::pair<size_t, size_t> > pairs;
    ORBmatcher tri(0.6f, false), fuse;
    out[0].n = hip ? tri.SearchForTriangulation(&S.kf[a], &S.kf[b], F12, pairs, r % 5 == 4)
                   : refSearchForTriangulation(&S.kf[a], &S.kf[b], F12, pairs, r % 5 == 4, false);
    out[0].h = 1469598103934665603ull;
    for (size_t i = 0; i < pairs.size(); i++) out[0].h = mix(mix(out[0].h, pairs[i].first), pairs[i].second);
    if (r % 6 == 0) buildFuse(W, *P.C, 977ull + (unsigned long long)r);
    const int c = r % 6;
    const float th = r % 4 == 3 ? 6.f : 3.f;
    out[1].n = hip ? fuse.Fuse(&W.kf[c], W.cand[c], th) : refFuse(&W.kf[c], W.cand[c], th);
    out[1].h = hashFuse(W);
}
static void roundL(const Plan &P, int r, bool hip, Res out[2])
{
    SharedMap &S = *P.S;
    const int a = (r * 11 + 5) % P.nkf;
    int b = (a + 3 * (1 + r % 4)) % P.nkf;
    if (b == a) b = (b + 1) % P.nkf;
    vector<MapPoint *> m12;
    ORBmatcher bow(0.75f, true), sim(0.75f, true);
    out[0].n = hip ? bow.SearchByBoW(&S.kf[a], &S.kf[b], m12) : refSearchByBoW(&S.kf[a], &S.kf[b], m12, 0.75f, true);
    out[0].h = hashVec(S, m12);
    cv::Mat R12, t12;
    relPose(S.kf[a], S.kf[b], R12, t12);
    const float s12 = 1.f + 0.001f * (float)(r % 3);
    out[1].n = hip ? sim.SearchBySim3(&S.kf[a], &S.kf[b], m12, s12, R12, t12, 7.5f) : refSim3(&S.kf[a], &S.kf[b], m12, s12, R12, t12, 7.5f);
    out[1].h = hashVec(S, m12);
}

#ifdef ORBHIP_TSAN_MOCK
extern "C" long mock_orbhip_calls();
extern "C" long mock_orbhip_evictions();
// three "extractions" of random features: positions inside the image, octaves 0..7, descriptors from 300 base rows with a few bits
// flipped, FeatureVectors over 90 nodes, the 64 x 48 grid by the drop-in's own AssignFeaturesToGrid (host arithmetic)
static void syntheticExtractions(Common &C, int nframes, int nf)
{
    g_rng = 0xC0FFEEull;
    vector<vector<unsigned char> > base(300, vector<unsigned char>(32));
    for (size_t b = 0; b < base.size(); b++)
        for (int k = 0; k < 32; k++) base[b][k] = (unsigned char)(256 * urand());
    C.ex.resize(nframes);
    for (int e = 0; e < nframes; e++) {
        Extraction &X = C.ex[e];
        const int n = nf - 7 * e;
        X.keys.resize(n);
        X.desc = cv::Mat(n, 32, CV_8U);
        for (int i = 0; i < n; i++) {
            cv::KeyPoint &kp = X.keys[i];
            kp.pt.x = (float)(20 + (C.w - 40) * urand()); kp.pt.y = (float)(20 + (C.h - 40) * urand());
            kp.octave = (int)(8 * urand()) & 7; kp.angle = (float)(360 * urand()); kp.size = 31.f; kp.response = 20.f; kp.class_id = -1;
            const int b = (int)(base.size() * urand()) % (int)base.size();
            memcpy(X.desc.ptr(i), base[b].data(), 32);
            for (int f = (int)(4 * urand()); f > 0; f--) { const int bit = (int)(256 * urand()) & 255; X.desc.ptr(i)[bit >> 3] ^= 1 << (bit & 7); }
            X.fv[(unsigned)(1000 + b % 90)].push_back((unsigned)i);
        }
        X.grid.mvKeys = X.keys; X.grid.mvKeysUn = X.keys; X.grid.N = n;
        X.grid.mpORBextractorLeft = NULL;
        X.grid.AssignFeaturesToGrid();
    }
    C.scale.resize(8); C.sigma2.resize(8); C.invSigma2.resize(8);
    for (int l = 0; l < 8; l++) { C.scale[l] = l ? C.scale[l - 1] * 1.2f : 1.f; C.sigma2[l] = C.scale[l] * C.scale[l]; C.invSigma2[l] = 1.f / C.sigma2[l]; }
}
#endif

int main(int argc, char **argv)
{
    Common C;
#ifdef ORBHIP_TSAN_MOCK
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    C.w = 640; C.h = 480;
    Frame::fx = 517.3f; Frame::fy = 516.5f; Frame::cx = 318.6f; Frame::cy = 255.3f;
    Frame::mnMinX = 0; Frame::mnMaxX = (float)C.w; Frame::mnMinY = 0; Frame::mnMaxY = (float)C.h;
    Frame::mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / (Frame::mnMaxX - Frame::mnMinX);
    Frame::mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / (Frame::mnMaxY - Frame::mnMinY);
    syntheticExtractions(C, 3, 600);
    SharedMap S;
    buildShared(S, C, 30);
    Plan P = {rounds, 30, &C, &S};
    ORBmatcher::SetResidentSetLimit(8);
    // ---- expected: the same calls, single-threaded (this thread has a context and a table of its own) ----
    vector<Res> wantT(2 * rounds), wantM(2 * rounds), wantL(2 * rounds);
    {
        FuseMap W;
        for (int r = 0; r < rounds; r++) {
            roundT(P, r, true, &wantT[2 * r]);
            roundM(P, r, true, W, &wantM[2 * r]);
            roundL(P, r, true, &wantL[2 * r]);
        }
    }
    long sum[6] = {0, 0, 0, 0, 0, 0};
    for (int r = 0; r < rounds; r++) {
        sum[0] += wantT[2 * r].n; sum[1] += wantT[2 * r + 1].n; sum[2] += wantM[2 * r].n; sum[3] += wantM[2 * r + 1].n;
        sum[4] += wantL[2 * r].n; sum[5] += wantL[2 * r + 1].n;
    }
    printf("single-threaded against the mock, %d rounds: %ld / %ld / %ld / %ld / %ld / %ld results; %ld library calls, %ld evictions\n", rounds,
           sum[0], sum[1], sum[2], sum[3], sum[4], sum[5], mock_orbhip_calls(), mock_orbhip_evictions());
    int fails = 0;
    for (int k = 0; k < 6; k++)
        if (sum[k] <= 0) { printf("FAILED: search %d returns nothing\n", k); fails++; }
    if (mock_orbhip_evictions() < rounds) { printf("FAILED: the schedule does not evict\n"); fails++; }
#else
    if (argc < 7) { fprintf(stderr, "usage: %s w h nfeatures frames.raw nframes vocabulary.bin [rounds]\n", argv[0]); return 2; }
    C.w = atoi(argv[1]); C.h = atoi(argv[2]);
    const int nf = atoi(argv[3]), nframes = atoi(argv[5]);
    const int rounds = argc > 7 ? atoi(argv[7]) : 500;
    vector<unsigned char> raw((size_t)C.w * C.h * nframes);
    FILE *f = fopen(argv[4], "rb");
    if (!f || fread(raw.data(), 1, raw.size(), f) != raw.size()) { perror(argv[4]); return 2; }
    fclose(f);
    ORBVocabulary voc;
    if (!voc.loadFromBinaryFile(argv[6])) { fprintf(stderr, "cannot load the vocabulary %s\n", argv[6]); return 2; }

    ORBextractor ex(nf, 1.2f, 8, 20, 7);
    ex.SetPyramidDownload(false);
    Frame::fx = 517.3f; Frame::fy = 516.5f; Frame::cx = 318.6f; Frame::cy = 255.3f;
    Frame::mnMinX = 0; Frame::mnMaxX = (float)C.w; Frame::mnMinY = 0; Frame::mnMaxY = (float)C.h;
    Frame::mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / (Frame::mnMaxX - Frame::mnMinX);
    Frame::mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / (Frame::mnMaxY - Frame::mnMinY);
    C.ex.resize(nframes);
    for (int e = 0; e < nframes; e++) {
        Extraction &X = C.ex[e];
        cv::Mat im(C.h, C.w, CV_8UC1, raw.data() + (size_t)e * C.w * C.h);
        ex(im, cv::Mat(), X.keys, X.desc);
        vector<cv::Mat> rows;
        for (int j = 0; j < X.desc.rows; j++) rows.push_back(X.desc.row(j));
        voc.transform(rows, X.bow, X.fv, 4);
        X.grid.mvKeys = X.keys; X.grid.mvKeysUn = X.keys; X.grid.N = (int)X.keys.size();
        X.grid.mpORBextractorLeft = &ex;
        X.grid.AssignFeaturesToGrid();
        printf("extraction %d: %d features, %zu nodes\n", e, X.grid.N, X.fv.size());
    }
    C.scale = ex.GetScaleFactors(); C.sigma2 = ex.GetScaleSigmaSquares(); C.invSigma2 = ex.GetInverseScaleSigmaSquares();

    SharedMap S;
    buildShared(S, C, 30);
    Plan P = {rounds, 30, &C, &S};

    // ---- expected: the host restatements, single-threaded ----
    vector<Res> wantT(2 * rounds), wantM(2 * rounds), wantL(2 * rounds);
    {
        FuseMap W;
        for (int r = 0; r < rounds; r++) {
            roundT(P, r, false, &wantT[2 * r]);
            roundM(P, r, false, W, &wantM[2 * r]);
            roundL(P, r, false, &wantL[2 * r]);
        }
    }
    long sum[6] = {0, 0, 0, 0, 0, 0};
    for (int r = 0; r < rounds; r++) {
        sum[0] += wantT[2 * r].n; sum[1] += wantT[2 * r + 1].n; sum[2] += wantM[2 * r].n; sum[3] += wantM[2 * r + 1].n;
        sum[4] += wantL[2 * r].n; sum[5] += wantL[2 * r + 1].n;
    }
    printf("host restatement, %d rounds: SearchByBoW(KF,F) %ld, SearchByProjection(F,F) %ld, SearchForTriangulation %ld, Fuse %ld, "
           "SearchByBoW(KF,KF) %ld, SearchBySim3 %ld matches in total\n", rounds, sum[0], sum[1], sum[2], sum[3], sum[4], sum[5]);
    int fails = 0;
    const long floor_[6] = {50, 100, 20, 20, 20, 5};       // per round on average: the searches have something to find
    for (int k = 0; k < 6; k++)
        if (sum[k] < floor_[k] * (long)rounds) { printf("FAILED: search %d finds too little to be a test (%ld)\n", k, sum[k]); fails++; }
#endif

    // ---- the drop-in from three threads at once ----
    ORBmatcher::SetResidentSetLimit(8);
    vector<Res> gotT(2 * rounds), gotM(2 * rounds), gotL(2 * rounds);
    std::thread tT([&]() {
        for (int r = 0; r < rounds; r++) {
            if (r == rounds / 2 || r % 97 == 96) ORBmatcher::DropResidentSets();     // Tracking::Reset, mid-run
            roundT(P, r, true, &gotT[2 * r]);
        }
    });
    std::thread tM([&]() {
        FuseMap W;
        for (int r = 0; r < rounds; r++) roundM(P, r, true, W, &gotM[2 * r]);
    });
    std::thread tL([&]() {
        for (int r = 0; r < rounds; r++) roundL(P, r, true, &gotL[2 * r]);
    });
    tT.join(); tM.join(); tL.join();
    const char *names[6] = {"T SearchByBoW(KF,F)", "T SearchByProjection(F,F)", "M SearchForTriangulation", "M Fuse", "L SearchByBoW(KF,KF)", "L SearchBySim3"};
    const vector<Res> *want[3] = {&wantT, &wantM, &wantL}, *got[3] = {&gotT, &gotM, &gotL};
    for (int t = 0; t < 3; t++)
        for (int k = 0; k < 2; k++) {
            int bad = 0, first = -1;
            for (int r = 0; r < rounds; r++)
                if (!same((*want[t])[2 * r + k], (*got[t])[2 * r + k])) { if (first < 0) first = r; bad++; }
            printf("%s: %s (%d of %d rounds differ%s)\n", names[2 * t + k], bad ? "FAILED" : "ok", bad, rounds,
                   bad ? (", first at round " + std::to_string(first) + ": " + std::to_string((*got[t])[2 * first + k].n) + " vs " +
                          std::to_string((*want[t])[2 * first + k].n)).c_str() : "");
            fails += bad != 0;
        }
    printf(fails ? "FAILED (%d)\n" : "all ok\n", fails);
    return fails ? 1 : 0;
}
