// ORBextractor.h -- drop-in replacement of the reference's include/ORBextractor.h:
// namespace ORB_SLAM2, class ORBextractor with the same constructor, operator(), getters, timing
// getters and the public member mvImagePyramid, so that src/Frame.cc (:61-67, :594-596, :817,
// :907-924) and src/Tracking.cc (:816-822) compile against it unchanged.  Everything is computed
// by liborbhip.so (HIP kernels for gfx950) through the C ABI in include/orbhip.h; there is no CPU
// path behind this class.
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <list>
#include <vector>

#include "cvlite.h"

struct orbhip_ctx;

namespace ORB_SLAM2
{
class ORBVocabulary;

class ORBextractor
{
public:
    // ref: include/ORBextractor.h:51-53 (VI-ORB-SLAM additions), milliseconds of the last call
    double GetTimeOfComputePyramid(void) { return mTimeOfComputePyramid; }
    double GetTimeOfComputeKeyPointsOctTree(void) { return mTimeOfComputeKeyPointsOctTree; }
    double GetTImeOfComputeDescriptor(void) { return mTimeOfComputeDescriptor; }

    enum {HARRIS_SCORE=0, FAST_SCORE=1 };

    // ref: include/ORBextractor.h:69-70.  The device context is sized on first use from the image.
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // ref: include/ORBextractor.h:77-79.  Mask is ignored, as in the reference.
    void operator()( cv::InputArray image, cv::InputArray mask,
      std::vector<cv::KeyPoint>& keypoints,
      cv::OutputArray descriptors);

    int inline GetLevels(){
        return nlevels;}

    float inline GetScaleFactor(){
        return scaleFactor;}

    std::vector<float> inline GetScaleFactors(){
        return mvScaleFactor;
    }

    std::vector<float> inline GetInverseScaleFactors(){
        return mvInvScaleFactor;
    }

    std::vector<float> inline GetScaleSigmaSquares(){
        return mvLevelSigma2;
    }

    std::vector<float> inline GetInverseScaleSigmaSquares(){
        return mvInvLevelSigma2;
    }

    // ref: include/ORBextractor.h:103 -- read directly by Frame::ComputeStereoMatches.
    // Valid after every operator() call until the next one: headers on a page-locked host copy of the levels that
    // liborbhip fills beside the kernels (orbhip_host_pyramid_level) -- level 0 is the staged copy of the caller's image.
    // Clone a level to keep it longer.  SetPyramidDownload(false) when the caller never reads it (monocular).
    std::vector<cv::Mat> mvImagePyramid;
    void SetPyramidDownload(bool on) { mbDownloadPyramid = on; }

    // device selection for multi-GPU processes (one process per GPU); default device 0
    static void SetDevice(int device);
    // last error text of the underlying context (empty if none).  The drop-in never throws (hiperror.h): a failed
    // operator() leaves no keypoints and released descriptors, like a frame without corners
    const char *LastError() const;
    orbhip_ctx *Context() { return mpCtx; }

    // ---- the Frame constructor's device work in one launch (orbhip_frame_build, include/orbhip.h) ----
    // Frame::Frame (ref: src/Frame.cc:518-572) calls operator(), UndistortKeyPoints, AssignFeaturesToGrid, and Tracking calls
    // ComputeBoW before the frame's first SearchByBoW: one dependency chain.  After SetFrameBuild, operator() runs the whole
    // chain as one graph launch and keeps the by-products; the Frame helpers of host/FrameGrid.cc (UndistortKeyPoints,
    // AssignFeaturesToGrid, ComputeBoW) take them from here when they are asked about the frame this extractor built last, and
    // fall back to their own device call otherwise -- the reference's call sequence stays as it is.
    //   K, distCoef     Frame::mK / mDistCoef (CV_32F)
    //   minX .. invH    Frame::mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv; invW <= 0: no grid (the first
    //                   frame of a run, whose bounds ComputeImageBounds derives after the extraction)
    //   voc, levelsup   the vocabulary ComputeBoW will use (NULL: no transform); levelsup 4 as in src/Frame.cc:744
    void SetFrameBuild(const cv::Mat &K, const cv::Mat &distCoef, float minX, float minY, float invW, float invH,
                       const ORBVocabulary *voc = 0, int levelsup = 4);
    void ClearFrameBuild() { mbFrameBuild = false; mnBuiltN = -1; }
    // Is (keys, descriptors) the frame the last operator() built?  (count, first and last keypoint, first descriptor)
    bool BuiltFrame(const std::vector<cv::KeyPoint> &keys) const;
    // by-products of that frame; false when there are none (frame build off, another frame, other parameters)
    // (K, distCoef: the asking Frame's calibration -- compared bit for bit with what SetFrameBuild was given, like the grid's bounds)
    bool BuiltKeysUn(const std::vector<cv::KeyPoint> &keys, const cv::Mat &K, const cv::Mat &distCoef,
                     std::vector<cv::KeyPoint> &keysUn) const;
    bool BuiltGrid(const std::vector<cv::KeyPoint> &keys, float minX, float minY, float invW, float invH, const int **cellOff,
                   const int **cellIdx) const;
    bool BuiltBoW(const std::vector<cv::KeyPoint> &keys, const ORBVocabulary *voc, int levelsup, const int **word, const float **weight,
                  const int **node) const;

protected:
    bool EnsureContext(int w, int h);

    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;

    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;

    std::vector<float> mvScaleFactor;
    std::vector<float> mvInvScaleFactor;
    std::vector<float> mvLevelSigma2;
    std::vector<float> mvInvLevelSigma2;

private:
    double mTimeOfComputePyramid;
    double mTimeOfComputeKeyPointsOctTree;
    double mTimeOfComputeDescriptor;

    orbhip_ctx *mpCtx;
    int mCtxW, mCtxH;
    bool mbDownloadPyramid;
    bool mbBadParams;     // the constructor arguments are outside what liborbhip runs: operator() reports and returns nothing
    std::vector<cv::KeyPoint> mvKpStage;

    // frame build (SetFrameBuild)
    bool mbFrameBuild;
    float mFbK[9], mFbDist[8], mFbGrid[4];
    int mFbNDist, mFbLevelsup;
    const ORBVocabulary *mpFbVoc;
    bool mbFbVocShared;
    unsigned long long mnFbVocGen;              // orbhip_vocab_generation of the lender when its tables were borrowed
    int mnBuiltN;                               // features of the frame whose by-products are held; -1: none
    bool mbBuiltGrid, mbBuiltBoW;
    std::vector<cv::KeyPoint> mvBuiltKeysUn;
    std::vector<int> mvBuiltCellOff, mvBuiltCellIdx, mvBuiltWord, mvBuiltNode;
    std::vector<float> mvBuiltWeight;
};

} //namespace ORB_SLAM

#endif
