// FrameGrid.cc -- Frame::UndistortKeyPoints / ComputeImageBounds (ref: src/Frame.cc:748-808) through
// orbhip_undistort_keypoints, and Frame::AssignFeaturesToGrid / GetFeaturesInArea (:574-589, :671-724) on the
// device grid of liborbhip (orbhip_frame_build's by-product, orbhip_features_in_area).  mGrid keeps the
// reference's public layout (a vector of feature indices per cell).  No exceptions (hiperror.h): a failed device call is
// reported and leaves the "nothing there" result.  The one piece of arithmetic done on the host is the cell number of a
// keypoint when AssignFeaturesToGrid is called outside a frame build (r06).
#include <algorithm>
#include <cmath>
#include <string>

#include "hiperror.h"
#include "orbhip.h"
#include "ORBVocabulary.h"
#include "ORBextractor.h"
#include "slamlite.h"

namespace ORB_SLAM2
{

float Frame::fx, Frame::fy, Frame::cx, Frame::cy;                      // ref: src/Frame.cc:35-37
float Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY;
float Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv;

static orbhip_ctx *frame_ctx(const Frame *F, const char *who)
{
    if (!F->mpORBextractorLeft || !F->mpORBextractorLeft->Context())
    {
        hipdetail::Fail(who, "the frame's extractor has no device context yet");
        return nullptr;
    }
    return F->mpORBextractorLeft->Context();
}

static void calib(const Frame *F, float K[9], std::vector<float> &D)
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) K[r * 3 + c] = F->mK.at<float>(r, c);
    const int n = F->mDistCoef.rows * F->mDistCoef.cols;
    D.resize(n);
    for (int i = 0; i < n; i++) D[i] = F->mDistCoef.rows == 1 ? F->mDistCoef.at<float>(0, i) : F->mDistCoef.at<float>(i, 0);
}

void Frame::UndistortKeyPoints()
{
    if(mDistCoef.at<float>(0, 0)==0.0)
    {
        mvKeysUn=mvKeys;
        return;
    }
    mvKeysUn.resize(N);
    if (N == 0) return;
    // the frame this extractor has just built (ORBextractor::SetFrameBuild): the undistorted keypoints came with the extraction
    if (mpORBextractorLeft && mpORBextractorLeft->BuiltKeysUn(mvKeys, mK, mDistCoef, mvKeysUn)) return;
    orbhip_ctx *ctx = frame_ctx(this, "Frame::UndistortKeyPoints");
    float K[9];
    std::vector<float> D;
    calib(this, K, D);
    // cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK) on (pt.x, pt.y); the other fields are copied
    if (orbhip_undistort_keypoints(ctx, reinterpret_cast<const orbhip_keypoint *>(mvKeys.data()), N, K, D.data(), (int)D.size(),
                                   K, reinterpret_cast<orbhip_keypoint *>(mvKeysUn.data())) != ORBHIP_OK)
    {
        hipdetail::Fail("Frame::UndistortKeyPoints", orbhip_last_error(ctx));
        mvKeysUn=mvKeys;          // the frame keeps its distorted coordinates
    }
}

void Frame::ComputeImageBounds(const cv::Mat &imLeft)
{
    if(mDistCoef.at<float>(0, 0)!=0.0)
    {
        orbhip_ctx *ctx = frame_ctx(this, "Frame::ComputeImageBounds");
        float K[9];
        std::vector<float> D;
        calib(this, K, D);
        cv::KeyPoint corner[4], un[4];
        corner[0].pt = cv::Point2f(0.0f, 0.0f);
        corner[1].pt = cv::Point2f((float)imLeft.cols, 0.0f);
        corner[2].pt = cv::Point2f(0.0f, (float)imLeft.rows);
        corner[3].pt = cv::Point2f((float)imLeft.cols, (float)imLeft.rows);
        if (orbhip_undistort_keypoints(ctx, reinterpret_cast<const orbhip_keypoint *>(corner), 4, K, D.data(), (int)D.size(), K,
                                       reinterpret_cast<orbhip_keypoint *>(un)) != ORBHIP_OK)
        {
            hipdetail::Fail("Frame::ComputeImageBounds", orbhip_last_error(ctx));
            for (int i = 0; i < 4; i++) un[i] = corner[i];   // bounds of the distorted image
        }
        mnMinX = std::min(un[0].pt.x, un[2].pt.x);
        mnMaxX = std::max(un[1].pt.x, un[3].pt.x);
        mnMinY = std::min(un[0].pt.y, un[1].pt.y);
        mnMaxY = std::max(un[2].pt.y, un[3].pt.y);
    }
    else
    {
        mnMinX = 0.0f;
        mnMaxX = imLeft.cols;
        mnMinY = 0.0f;
        mnMaxY = imLeft.rows;
    }
}

void Frame::AssignFeaturesToGrid()
{
    for (unsigned int i = 0; i < FRAME_GRID_COLS; i++)
        for (unsigned int j = 0; j < FRAME_GRID_ROWS; j++) mGrid[i][j].clear();
    if (N == 0) return;
    const int *boff = 0, *bidx = 0;
    if (mpORBextractorLeft && mpORBextractorLeft->BuiltGrid(mvKeys, mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv, &boff, &bidx))
    {
        // the grid of the frame build (same kernel, same parameters)
        for (int i = 0; i < FRAME_GRID_COLS; i++)
            for (int j = 0; j < FRAME_GRID_ROWS; j++) {
                const int c = i * FRAME_GRID_ROWS + j;
                mGrid[i][j].assign(bidx + boff[c], bidx + boff[c + 1]);
            }
        return;
    }
    // Called on its own (a frame that did not come out of orbhip_frame_build): 64 x 48 counters and N cell numbers -- a
    // device round trip costs four times what the binning does on the host (profiles/r05/percall_table.md: 0.038 vs 0.010 ms),
    // so it is done here, with the float operations of k_grid_build's grid_cell (ref: src/Frame.cc:726-736 PosInGrid: the
    // difference, then the product, each rounded to float, then round()), features in index order = the reference's push_back
    // order.  The device grid (orbhip_frame_build, orbhip_set_put) is the same function of the same keypoints.
    for (int i = 0; i < N; i++)
    {
        const cv::KeyPoint &kp = mvKeysUn[i];
        const float gx = (kp.pt.x - mnMinX) * mfGridElementWidthInv;
        const float gy = (kp.pt.y - mnMinY) * mfGridElementHeightInv;
        const int cellX = (int)roundf(gx), cellY = (int)roundf(gy);
        if (cellX < 0 || cellX >= FRAME_GRID_COLS || cellY < 0 || cellY >= FRAME_GRID_ROWS) continue;   // outside the undistorted image
        mGrid[cellX][cellY].push_back((std::size_t)i);
    }
}

void Frame::ComputeBoW()
{
    if(mBowVec.empty())                                                   // ref: src/Frame.cc:741
    {
        if (!mpORBvocabulary) return;
        const int *word = 0, *node = 0;
        const float *weight = 0;
        if (mpORBextractorLeft && mpORBextractorLeft->BuiltBoW(mvKeys, mpORBvocabulary, 4, &word, &weight, &node))
        {
            mpORBvocabulary->assemble(word, weight, node, N, mBowVec, mFeatVec);
            return;
        }
        std::vector<cv::Mat> vCurrentDesc;                                // Converter::toDescriptorVector (src/Converter.cc:163-171)
        vCurrentDesc.reserve(mDescriptors.rows);
        for (int j = 0; j < mDescriptors.rows; j++) vCurrentDesc.push_back(mDescriptors.row(j));
        mpORBvocabulary->transform(vCurrentDesc,mBowVec,mFeatVec,4);      // ref: :744
    }
}

std::vector<size_t> Frame::GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel,
                                             const int maxLevel) const
{
    std::vector<size_t> vIndices;
    if (N == 0) return vIndices;
    orbhip_ctx *ctx = frame_ctx(this, "Frame::GetFeaturesInArea");
    orbhip_proj_query q = {x, y, r, 0.f, minLevel, maxLevel, 0.f, ORBHIP_Q_ACTIVE};
    std::vector<int32_t> idx(N);
    int32_t off[2] = {0, 0};
    if (orbhip_features_in_area(ctx, reinterpret_cast<const orbhip_keypoint *>(mvKeysUn.data()), N, mnMinX, mnMinY,
                                mfGridElementWidthInv, mfGridElementHeightInv, &q, 1, off, idx.data(), N) != ORBHIP_OK)
        return hipdetail::Fail("Frame::GetFeaturesInArea", orbhip_last_error(ctx)), vIndices;
    vIndices.assign(idx.begin(), idx.begin() + off[1]);
    return vIndices;
}

}  // namespace ORB_SLAM2
