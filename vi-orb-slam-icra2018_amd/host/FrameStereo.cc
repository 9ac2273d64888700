// FrameStereo.cc -- Frame::ComputeStereoMatches (ref: src/Frame.cc:810-984) as one call into liborbhip:
// the row-band descriptor search (:848-893), the 11-shift SAD refinement on the two image pyramids
// (:895-955) and the median cut (:966-983) all run on the device, on the pyramid levels the two
// extractor contexts still hold from the operator() calls of this frame -- mvImagePyramid never
// crosses PCIe.  There is no CPU path and no exception (hiperror.h): on a failed
// device call no keypoint gets a depth.
#include <cstring>
#include <string>

#include "hiperror.h"
#include "orbhip.h"
#include "ORBextractor.h"
#include "slamlite.h"

namespace ORB_SLAM2
{

static const unsigned char *rows32(const cv::Mat &m, std::vector<unsigned char> &tmp)
{
    if (m.rows == 0) return nullptr;
    if (m.isContinuous() || m.rows == 1) return m.data;
    tmp.resize((size_t)m.rows * 32);
    for (int i = 0; i < m.rows; i++) memcpy(&tmp[(size_t)i * 32], m.ptr(i), 32);
    return tmp.data();
}

void Frame::ComputeStereoMatches()
{
    mvuRight.assign(N, -1.0f);                                   // ref: :812-813
    mvDepth.assign(N, -1.0f);
    if (N == 0) return;
    if (!mpORBextractorLeft || !mpORBextractorRight || !mpORBextractorLeft->Context() ||
        !mpORBextractorRight->Context())
    {
        hipdetail::Fail("Frame::ComputeStereoMatches", "both extractors must have run on this frame");
        return;
    }
    std::vector<unsigned char> tl, tr;
    int nmatch = 0;
    const int rc = orbhip_stereo_match(mpORBextractorLeft->Context(), mpORBextractorRight->Context(),
                                       reinterpret_cast<const orbhip_keypoint *>(mvKeys.data()), rows32(mDescriptors, tl), N,
                                       reinterpret_cast<const orbhip_keypoint *>(mvKeysRight.data()),
                                       rows32(mDescriptorsRight, tr), (int)mvKeysRight.size(), mb, mbf, mvuRight.data(),
                                       mvDepth.data(), &nmatch);
    if (rc != ORBHIP_OK)
    {
        hipdetail::Fail("Frame::ComputeStereoMatches", orbhip_last_error(mpORBextractorLeft->Context()));
        mvuRight.assign(N, -1.0f);
        mvDepth.assign(N, -1.0f);
    }
}

}  // namespace ORB_SLAM2
