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
};

} //namespace ORB_SLAM

#endif
