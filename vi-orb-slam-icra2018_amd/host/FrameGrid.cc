// FrameGrid.cc -- Frame::AssignFeaturesToGrid and Frame::GetFeaturesInArea (ref: src/Frame.cc:574-589,
// :671-724) on the device grid of liborbhip (orbhip_grid_build, orbhip_features_in_area).  mGrid keeps the
// reference's public layout (a vector of feature indices per cell).  No CPU path: errors throw.
#include <stdexcept>
#include <string>

#include "orbhip.h"
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
        throw std::runtime_error(std::string(who) + ": the frame's extractor has no device context yet");
    return F->mpORBextractorLeft->Context();
}

void Frame::AssignFeaturesToGrid()
{
    for (unsigned int i = 0; i < FRAME_GRID_COLS; i++)
        for (unsigned int j = 0; j < FRAME_GRID_ROWS; j++) mGrid[i][j].clear();
    if (N == 0) return;
    orbhip_ctx *ctx = frame_ctx(this, "Frame::AssignFeaturesToGrid");
    std::vector<int32_t> off(ORBHIP_GRID_CELLS + 1), idx(N);
    if (orbhip_grid_build(ctx, reinterpret_cast<const orbhip_keypoint *>(mvKeysUn.data()), N, mnMinX, mnMinY,
                          mfGridElementWidthInv, mfGridElementHeightInv, off.data(), idx.data()) != ORBHIP_OK)
        throw std::runtime_error(std::string("Frame::AssignFeaturesToGrid: ") + orbhip_last_error(ctx));
    for (int i = 0; i < FRAME_GRID_COLS; i++)
        for (int j = 0; j < FRAME_GRID_ROWS; j++) {
            const int c = i * FRAME_GRID_ROWS + j;
            mGrid[i][j].assign(idx.begin() + off[c], idx.begin() + off[c + 1]);
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
        throw std::runtime_error(std::string("Frame::GetFeaturesInArea: ") + orbhip_last_error(ctx));
    vIndices.assign(idx.begin(), idx.begin() + off[1]);
    return vIndices;
}

}  // namespace ORB_SLAM2
