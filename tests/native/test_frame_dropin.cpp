// test_frame_dropin.cpp -- the Frame constructor's sequence (ref: src/Frame.cc:518-572: ExtractORB, UndistortKeyPoints,
// ComputeImageBounds on the first frame, AssignFeaturesToGrid; ComputeBoW :739-746 from Tracking) through the drop-in classes,
// twice over the same images:
//   world A  every step its own device call (ORBextractor::operator(), Frame::UndistortKeyPoints, ...);
//   world B  ORBextractor::SetFrameBuild from the second frame on: operator() runs orbhip_frame_build (one graph launch) and
//            the same Frame helpers take its by-products.
// Every member the constructor fills must be identical in both worlds (world A's calls are checked against the oracle by the
// other drop-in tests), and so must ORBmatcher::SearchByBoW(KF, F) -- in world B the frame enters the matcher's resident sets
// from the device block of its build.  Also the two identity hazards of resident sets: an id that is used again
// for other data (Tracking::Reset restarts Frame::nNextId, ref: src/Tracking.cc:2758-2759) and a key frame first met with an
// empty FeatureVector (before KeyFrame::ComputeBoW, ref: src/KeyFrame.cc:392-400).
// usage: test_frame_dropin w h nfeatures frames.raw nframes voc.bin      prints one "name ok|FAIL" line per check; exit code
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ORBVocabulary.h"
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "orbhip.h"

using namespace ORB_SLAM2;

static int g_fail = 0;
static void check(const char *name, bool ok)
{
    printf("%s %s\n", name, ok ? "ok" : "FAIL");
    if (!ok) g_fail++;
}

static void build_frame(Frame &F, ORBextractor &ex, ORBVocabulary &voc, const cv::Mat &im, bool first)
{
    F.mpORBextractorLeft = &ex;
    F.mpORBvocabulary = &voc;
    F.mK = cv::Mat(3, 3, CV_32F);                             // Examples/Monocular/EuRoC.yaml
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) F.mK.at<float>(r, c) = r == c ? 1.f : 0.f;
    F.mK.at<float>(0, 0) = 458.654f; F.mK.at<float>(1, 1) = 457.296f; F.mK.at<float>(0, 2) = 367.215f; F.mK.at<float>(1, 2) = 248.375f;
    F.mDistCoef = cv::Mat(4, 1, CV_32F);
    F.mDistCoef.at<float>(0, 0) = -0.28340811f; F.mDistCoef.at<float>(1, 0) = 0.07395907f;
    F.mDistCoef.at<float>(2, 0) = 0.00019359f; F.mDistCoef.at<float>(3, 0) = 1.76187114e-05f;
    ex(im, cv::Mat(), F.mvKeys, F.mDescriptors);             // ExtractORB, :591-597
    F.N = (int)F.mvKeys.size();
    F.UndistortKeyPoints();
    if (first) {                                              // :551-567
        F.ComputeImageBounds(im);
        Frame::mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(Frame::mnMaxX - Frame::mnMinX);
        Frame::mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(Frame::mnMaxY - Frame::mnMinY);
    }
    F.AssignFeaturesToGrid();
    F.ComputeBoW();
}

static bool same_frame(const Frame &a, const Frame &b)
{
    if (a.N != b.N || a.N < 100) return false;
    if (memcmp(a.mvKeys.data(), b.mvKeys.data(), (size_t)a.N * sizeof(cv::KeyPoint))) return false;
    if (memcmp(a.mvKeysUn.data(), b.mvKeysUn.data(), (size_t)a.N * sizeof(cv::KeyPoint))) return false;
    for (int i = 0; i < a.N; i++)
        if (memcmp(a.mDescriptors.ptr(i), b.mDescriptors.ptr(i), 32)) return false;
    for (int i = 0; i < FRAME_GRID_COLS; i++)
        for (int j = 0; j < FRAME_GRID_ROWS; j++)
            if (a.mGrid[i][j] != b.mGrid[i][j]) return false;
    if (a.mBowVec.size() != b.mBowVec.size() || a.mBowVec.empty()) return false;
    for (DBoW2::BowVector::const_iterator x = a.mBowVec.begin(), y = b.mBowVec.begin(); x != a.mBowVec.end(); ++x, ++y)
        if (x->first != y->first || x->second != y->second) return false;
    if (a.mFeatVec.size() != b.mFeatVec.size()) return false;
    for (DBoW2::FeatureVector::const_iterator x = a.mFeatVec.begin(), y = b.mFeatVec.begin(); x != a.mFeatVec.end(); ++x, ++y)
        if (x->first != y->first || x->second != y->second) return false;
    return true;
}

static void make_keyframe(KeyFrame &kf, const Frame &F, std::vector<MapPoint> &points)
{
    kf.mvKeys = F.mvKeys;
    kf.mvKeysUn = F.mvKeysUn;
    kf.mDescriptors = F.mDescriptors.clone();
    kf.mFeatVec = F.mFeatVec;
    kf.N = F.N;
    kf.mnMinX = Frame::mnMinX; kf.mnMinY = Frame::mnMinY; kf.mnMaxX = Frame::mnMaxX; kf.mnMaxY = Frame::mnMaxY;
    kf.mfGridElementWidthInv = Frame::mfGridElementWidthInv;
    kf.mfGridElementHeightInv = Frame::mfGridElementHeightInv;
    kf.mvpMapPoints.resize(F.N);
    for (int i = 0; i < F.N; i++) kf.mvpMapPoints[i] = &points[i];
}

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s w h nfeatures frames.raw nframes voc.bin\n", argv[0]); return 2; }
    const int w = atoi(argv[1]), h = atoi(argv[2]), nf = atoi(argv[3]), nfr = atoi(argv[5]);
    std::vector<unsigned char> pix((size_t)w * h * nfr);
    FILE *f = fopen(argv[4], "rb");
    if (!f || fread(pix.data(), 1, pix.size(), f) != pix.size()) { perror(argv[4]); return 2; }
    fclose(f);
    ORBVocabulary voc;
    if (!voc.loadFromBinaryFile(argv[6])) { fprintf(stderr, "cannot load %s\n", argv[6]); return 2; }

    ORBextractor exA(nf, 1.2f, 8, 20, 7), exB(nf, 1.2f, 8, 20, 7);
    exA.SetPyramidDownload(false);
    exB.SetPyramidDownload(false);
    std::vector<Frame> A(nfr), B(nfr);
    for (int i = 0; i < nfr; i++) {
        cv::Mat im(h, w, CV_8UC1, (void *)(pix.data() + (size_t)i * w * h));
        build_frame(A[i], exA, voc, im, i == 0);
        // world B: from the second frame on the bounds are known, the whole constructor is one launch
        if (i == 1)
            exB.SetFrameBuild(A[0].mK, A[0].mDistCoef, Frame::mnMinX, Frame::mnMinY, Frame::mfGridElementWidthInv,
                              Frame::mfGridElementHeightInv, &voc, 4);
        build_frame(B[i], exB, voc, im, i == 0);
        char name[64];
        snprintf(name, sizeof name, "frame%d_members_equal", i);
        check(name, same_frame(A[i], B[i]));
        if (i >= 1) {
            snprintf(name, sizeof name, "frame%d_built_in_one_launch", i);
            check(name, exB.BuiltFrame(B[i].mvKeys) && orbhip_frame_fingerprint(exB.Context()) != 0);
        }
    }
    check("world_a_has_no_frame_build", orbhip_frame_fingerprint(exA.Context()) == 0);

    // SearchByBoW(KF, F): the key frame is frame 0; the current frame is the LAST one, the one exB still holds on the device
    std::vector<MapPoint> points(8192);
    KeyFrame kfA, kfB;
    make_keyframe(kfA, A[0], points);
    make_keyframe(kfB, B[0], points);
    ORBmatcher matcher(0.7, true);
    std::vector<MapPoint *> mA, mB;
    const int nA = matcher.SearchByBoW(&kfA, A[nfr - 1], mA);
    const int nB = matcher.SearchByBoW(&kfB, B[nfr - 1], mB);
    check("search_by_bow_equal", nA == nB && nA > 20 && mA == mB);

    // an id that comes back with other data (Tracking::Reset): frame X gets B[1]'s id and the features of B[2] cut to B[1]'s count
    {
        const int n = std::min(B[1].N, B[2].N);
        auto cut = [&](const Frame &src, Frame &dst) {
            dst.mpORBextractorLeft = src.mpORBextractorLeft;
            dst.mvKeys.assign(src.mvKeys.begin(), src.mvKeys.begin() + n);
            dst.mvKeysUn.assign(src.mvKeysUn.begin(), src.mvKeysUn.begin() + n);
            dst.mDescriptors = cv::Mat(n, 32, CV_8U);
            for (int i = 0; i < n; i++) memcpy(dst.mDescriptors.ptr(i), src.mDescriptors.ptr(i), 32);
            dst.N = n;
            for (DBoW2::FeatureVector::const_iterator it = src.mFeatVec.begin(); it != src.mFeatVec.end(); ++it)
                for (size_t k = 0; k < it->second.size(); k++)
                    if ((int)it->second[k] < n) dst.mFeatVec.addFeature(it->first, it->second[k]);
        };
        Frame first, again, fresh;
        cut(B[1], first);
        cut(B[2], again);
        cut(B[2], fresh);
        again.mnId = first.mnId;                              // the id counter was reset: same id, same count, other features
        std::vector<MapPoint *> m1, m2, m3;
        matcher.SearchByBoW(&kfB, first, m1);                 // registers the set under first.mnId
        const int n2 = matcher.SearchByBoW(&kfB, again, m2);  // must not be matched against `first`'s descriptors
        const int n3 = matcher.SearchByBoW(&kfB, fresh, m3);
        check("reused_id_is_not_a_hit", n2 == n3 && m2 == m3 && m2 != m1);
    }
    // a key frame first met before ComputeBoW
    {
        KeyFrame late;
        make_keyframe(late, B[0], points);
        late.mFeatVec.clear();
        std::vector<MapPoint *> m0, m1;
        const int before = matcher.SearchByBoW(&late, B[nfr - 1], m0);    // nothing to match yet; its set is registered with ng = 0
        late.mFeatVec = B[0].mFeatVec;                                     // KeyFrame::ComputeBoW has run
        const int after = matcher.SearchByBoW(&late, B[nfr - 1], m1);
        check("feature_vector_filled_later_is_seen", before == 0 && after == nB && m1 == mB);
    }
    ORBmatcher::DropResidentSets();
    std::vector<MapPoint *> mC;
    check("after_drop_equal", matcher.SearchByBoW(&kfB, B[nfr - 1], mC) == nB && mC == mB);
    return g_fail ? 1 : 0;
}
