// ORBextractor.cc -- host side of the drop-in ORB_SLAM2::ORBextractor (include/orbhip/ORBextractor.h).
// Mirrors the control flow of the reference's operator() (src/ORBextractor.cc:1045-1126): empty
// image -> silent return, CV_8UC1 assertion, three stage timers, keypoints cleared and refilled,
// descriptors created as N x 32 CV_8U -- with every computation delegated to liborbhip.so.
#include "ORBextractor.h"

#include <cassert>
#include <cstdio>
#include <cstring>
#include <string>

#include "ORBVocabulary.h"
#include "hiperror.h"
#include "orbhip.h"

static_assert(sizeof(cv::KeyPoint) == sizeof(orbhip_keypoint), "cv::KeyPoint must be the 28-byte OpenCV layout");

namespace ORB_SLAM2
{

static int g_device = 0;

void ORBextractor::SetDevice(int device) { g_device = device; }

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels,
         int _iniThFAST, int _minThFAST):
    nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels),
    iniThFAST(_iniThFAST), minThFAST(_minThFAST),
    mTimeOfComputePyramid(0), mTimeOfComputeKeyPointsOctTree(0), mTimeOfComputeDescriptor(0),
    mpCtx(nullptr), mCtxW(0), mCtxH(0), mbDownloadPyramid(true), mbBadParams(false),
    mbFrameBuild(false), mFbNDist(0), mFbLevelsup(-1), mpFbVoc(nullptr), mbFbVocShared(false), mnFbVocGen(0), mnBuiltN(-1), mbBuiltGrid(false),
    mbBuiltBoW(false)
{
    // scale tables, per-level quotas and umax (ref: src/ORBextractor.cc:417-471) -- host arithmetic
    // inside liborbhip, no device needed yet
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    umax.resize(16);
    // (the reference's constructor accepts anything; parameters liborbhip cannot run -- nlevels outside 1..16,
    // scaleFactor <= 1 -- leave an extractor that reports the error and returns no keypoints)
    mbBadParams = orbhip_tables(nfeatures, _scaleFactor, nlevels, iniThFAST, minThFAST, mvScaleFactor.data(),
                                mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                                mnFeaturesPerLevel.data(), umax.data()) != ORBHIP_OK;
    if (mbBadParams) hipdetail::Fail("ORBextractor::ORBextractor", orbhip_last_error(nullptr));
    mvImagePyramid.resize(nlevels > 0 ? nlevels : 0);
}

ORBextractor::~ORBextractor()
{
    if (mpCtx) orbhip_destroy(mpCtx);
}

const char *ORBextractor::LastError() const { return mpCtx ? orbhip_last_error(mpCtx) : OrbHipLastError(); }

bool ORBextractor::EnsureContext(int w, int h)
{
    if (mpCtx && w <= mCtxW && h <= mCtxH) return true;
    if (mpCtx) orbhip_destroy(mpCtx);
    mCtxW = w > mCtxW ? w : mCtxW;
    mCtxH = h > mCtxH ? h : mCtxH;
    mpCtx = orbhip_create(g_device, nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST, mCtxW, mCtxH, 1);
    mbFbVocShared = false;
    return mpCtx != nullptr;
}

void ORBextractor::SetFrameBuild(const cv::Mat &K, const cv::Mat &distCoef, float minX, float minY, float invW, float invH,
                                 const ORBVocabulary *voc, int levelsup)
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) mFbK[r * 3 + c] = K.at<float>(r, c);
    const int nd = distCoef.rows * distCoef.cols;
    mFbNDist = (nd == 4 || nd == 5 || nd == 8) ? nd : 0;
    for (int i = 0; i < 8; i++) mFbDist[i] = 0.f;
    for (int i = 0; i < mFbNDist; i++) mFbDist[i] = distCoef.rows == 1 ? distCoef.at<float>(0, i) : distCoef.at<float>(i, 0);
    mFbGrid[0] = minX; mFbGrid[1] = minY; mFbGrid[2] = invW; mFbGrid[3] = invH;
    if (voc != mpFbVoc) mbFbVocShared = false;
    mpFbVoc = (voc && !voc->empty()) ? voc : nullptr;
    mFbLevelsup = mpFbVoc ? levelsup : -1;
    mbFrameBuild = true;
    mnBuiltN = -1;
}

bool ORBextractor::BuiltFrame(const std::vector<cv::KeyPoint> &keys) const
{
    const int n = (int)keys.size();
    if (mnBuiltN < 0 || n != mnBuiltN || n == 0) return false;
    return memcmp(&keys[0], &mvKpStage[0], sizeof(cv::KeyPoint)) == 0 && memcmp(&keys[n - 1], &mvKpStage[n - 1], sizeof(cv::KeyPoint)) == 0;
}

bool ORBextractor::BuiltKeysUn(const std::vector<cv::KeyPoint> &keys, const cv::Mat &K, const cv::Mat &distCoef,
                               std::vector<cv::KeyPoint> &keysUn) const
{
    if (!BuiltFrame(keys)) return false;
    // the frame's own calibration, bit for bit what the build ran with -- otherwise Frame::UndistortKeyPoints makes its own call
    if (K.rows != 3 || K.cols != 3) return false;
    float k[9];
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) k[r * 3 + c] = K.at<float>(r, c);
    if (memcmp(k, mFbK, sizeof(k)) != 0) return false;
    const int nd = distCoef.rows * distCoef.cols;
    if (nd != mFbNDist) return false;
    for (int i = 0; i < nd; i++)
    {
        const float d = distCoef.rows == 1 ? distCoef.at<float>(0, i) : distCoef.at<float>(i, 0);
        if (memcmp(&d, &mFbDist[i], sizeof(float)) != 0) return false;
    }
    keysUn.assign(mvBuiltKeysUn.begin(), mvBuiltKeysUn.begin() + mnBuiltN);
    return true;
}

bool ORBextractor::BuiltGrid(const std::vector<cv::KeyPoint> &keys, float minX, float minY, float invW, float invH, const int **cellOff,
                             const int **cellIdx) const
{
    if (!mbBuiltGrid || !BuiltFrame(keys) || minX != mFbGrid[0] || minY != mFbGrid[1] || invW != mFbGrid[2] || invH != mFbGrid[3]) return false;
    *cellOff = mvBuiltCellOff.data();
    *cellIdx = mvBuiltCellIdx.data();
    return true;
}

bool ORBextractor::BuiltBoW(const std::vector<cv::KeyPoint> &keys, const ORBVocabulary *voc, int levelsup, const int **word,
                            const float **weight, const int **node) const
{
    if (!mbBuiltBoW || !BuiltFrame(keys) || voc != mpFbVoc || levelsup != mFbLevelsup) return false;
    *word = mvBuiltWord.data();
    *weight = mvBuiltWeight.data();
    *node = mvBuiltNode.data();
    return true;
}

void ORBextractor::operator()( cv::InputArray _image, cv::InputArray _mask, std::vector<cv::KeyPoint>& _keypoints,
                      cv::OutputArray _descriptors)
{
    (void)_mask;
    if(_image.empty())
        return;                                              // ref: :1048-1049

    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1 );                        // ref: :1052

    // every failure below leaves the caller with what the reference leaves for a frame without corners: no keypoints,
    // released descriptors (ref: :1080-1081) -- never an exception (hiperror.h)
    _keypoints.clear();
    if (mbBadParams || !EnsureContext(image.cols, image.rows))
    {
        _descriptors.release();
        hipdetail::Fail("ORBextractor::operator()", mbBadParams ? "constructed with parameters liborbhip rejects" : orbhip_last_error(nullptr));
        return;
    }

    orbhip_set_host_pyramid(mpCtx, mbDownloadPyramid ? 1 : 0);
    const int cap = orbhip_max_keypoints(mpCtx);
    mvKpStage.resize(cap);
    cv::Mat descStage(cap, 32, CV_8U);
    int n = 0;
    float t[3] = {0, 0, 0};
    int rc;
    mnBuiltN = -1;
    if (mbFrameBuild)
    {
        // extraction + UndistortKeyPoints + AssignFeaturesToGrid (+ the vocabulary transform) as one graph launch; the by-products
        // wait here for the Frame helpers (host/FrameGrid.cc)
        orbhip_frame_params fp;
        memcpy(fp.K, mFbK, sizeof(fp.K));
        memcpy(fp.dist, mFbDist, sizeof(fp.dist));
        fp.ndist = mFbNDist;
        fp.min_x = mFbGrid[0]; fp.min_y = mFbGrid[1]; fp.inv_w = mFbGrid[2]; fp.inv_h = mFbGrid[3];
        fp.levelsup = mFbLevelsup;
        // (the vocabulary object may have loaded again since its tables were borrowed: its generation says so)
        if (mpFbVoc && mbFbVocShared && orbhip_vocab_generation(mpFbVoc->Context()) != mnFbVocGen) mbFbVocShared = false;
        if (mpFbVoc && !mbFbVocShared)
        {
            if (mpFbVoc->Context() && orbhip_vocab_share(mpCtx, mpFbVoc->Context()) == ORBHIP_OK)
            {
                mbFbVocShared = true;
                mnFbVocGen = orbhip_vocab_generation(mpFbVoc->Context());
            }
            else
                fp.levelsup = -1;                            // (ComputeBoW then runs its own transform)
        }
        const bool grid = fp.inv_w > 0.f && fp.inv_h > 0.f, bow = fp.levelsup >= 0;
        mvBuiltKeysUn.resize(cap);
        mvBuiltCellOff.resize(ORBHIP_GRID_CELLS + 1);
        mvBuiltCellIdx.resize(cap);
        mvBuiltWord.resize(cap);
        mvBuiltNode.resize(cap);
        mvBuiltWeight.resize(cap);
        rc = orbhip_frame_build(mpCtx, image.data, image.cols, image.rows, (int)image.step, &fp,
                                reinterpret_cast<orbhip_keypoint *>(mvKpStage.data()),
                                reinterpret_cast<orbhip_keypoint *>(mvBuiltKeysUn.data()), descStage.data, cap, &n,
                                grid ? mvBuiltCellOff.data() : nullptr, grid ? mvBuiltCellIdx.data() : nullptr,
                                bow ? mvBuiltWord.data() : nullptr, bow ? mvBuiltWeight.data() : nullptr, bow ? mvBuiltNode.data() : nullptr);
        if (rc == ORBHIP_OK)
        {
            mnBuiltN = n;
            mbBuiltGrid = grid;
            mbBuiltBoW = bow;
            float ms[6];
            if (orbhip_get_stage_times(mpCtx, ms) == ORBHIP_OK) { t[0] = ms[0]; t[1] = ms[1] + ms[2]; t[2] = ms[3] + ms[4]; }
        }
    }
    else
        rc = orbhip_extract(mpCtx, image.data, image.cols, image.rows, (int)image.step,
                            reinterpret_cast<orbhip_keypoint *>(mvKpStage.data()), descStage.data, cap, &n, t);
    if (rc != ORBHIP_OK)
    {
        _descriptors.release();
        hipdetail::Fail("ORBextractor::operator()", orbhip_last_error(mpCtx));
        return;
    }
    mTimeOfComputePyramid = t[0];
    mTimeOfComputeKeyPointsOctTree = t[1];
    mTimeOfComputeDescriptor = t[2];

    if( n == 0 )
        _descriptors.release();                              // ref: :1080-1081
    else
    {
        _descriptors.create(n, 32, CV_8U);                   // ref: :1084
        cv::Mat descriptors = _descriptors.getMat();
        for (int i = 0; i < n; i++) memcpy(descriptors.ptr(i), descStage.ptr(i), 32);
    }
    _keypoints.clear();
    _keypoints.reserve(n);
    _keypoints.insert(_keypoints.end(), mvKpStage.begin(), mvKpStage.begin() + n);

    if (mbDownloadPyramid)
    {
        // mvImagePyramid[level] (ref: include/ORBextractor.h:103, read by src/Frame.cc:817): headers on the page-locked host
        // copy that liborbhip filled beside the kernels -- valid until the next call of this extractor, which is how the
        // reference's callers use them.  A level that is not staged there (ORBHIP_NO_GRAPH runs) is copied instead.
        for (int level = 0; level < nlevels; ++level)
        {
            const uint8_t *p = nullptr;
            int st = 0, w = 0, h = 0;
            if (orbhip_host_pyramid_level(mpCtx, 0, level, &p, &st, &w, &h) == ORBHIP_OK)
            {
                mvImagePyramid[level] = cv::Mat(h, w, CV_8UC1, (void *)p, (size_t)st);
                continue;
            }
            orbhip_get_pyramid_level(mpCtx, 0, level, nullptr, 0, &w, &h);
            mvImagePyramid[level].release();                 // (it may be a header on the page-locked block of an earlier call)
            mvImagePyramid[level].create(h, w, CV_8UC1);
            if (orbhip_get_pyramid_level(mpCtx, 0, level, mvImagePyramid[level].data, (int)mvImagePyramid[level].step,
                                         &w, &h) != ORBHIP_OK)
            {
                hipdetail::Fail("ORBextractor::operator() (mvImagePyramid download)", orbhip_last_error(mpCtx));
                return;
            }
        }
    }
}

} //namespace ORB_SLAM
