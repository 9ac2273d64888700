// ORBmatcher.cc -- host side of the drop-in ORB_SLAM2::ORBmatcher (include/orbhip/ORBmatcher.h).
// SearchByBoW keeps the reference's interface and result conventions (src/ORBmatcher.cc:159-288,
// :522-655): the FeatureVectors are flattened to CSR in std::map order, "has a good MapPoint"
// becomes a byte mask, and the node-constrained brute force + ratio test + rotation histogram run
// in liborbhip.so (orbhip_search_by_bow).
#include "ORBmatcher.h"

#include <stdexcept>
#include <string>

#include "orbhip.h"

using namespace std;

namespace ORB_SLAM2
{

const int ORBmatcher::TH_HIGH = 100;     // ref: src/ORBmatcher.cc:37-39
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

static int g_match_device = 0;
void ORBmatcher::SetDevice(int device) { g_match_device = device; }

namespace {
// one small device context per host thread (matchers are used concurrently from Tracking,
// LocalMapping and LoopClosing; a context is not re-entrant)
struct ThreadCtx {
    orbhip_ctx *ctx = nullptr;
    ~ThreadCtx() { if (ctx) orbhip_destroy(ctx); }
    orbhip_ctx *get()
    {
        if (!ctx) {
            ctx = orbhip_create(g_match_device, 50, 1.2f, 1, 20, 7, 128, 128, 1);
            if (!ctx) throw std::runtime_error(std::string("ORBmatcher: ") + orbhip_last_error(nullptr));
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
}  // namespace

ORBmatcher::ORBmatcher(float nnratio, bool checkOri): mfNNratio(nnratio), mbCheckOrientation(checkOri)
{
}

ORBmatcher::~ORBmatcher() {}

int ORBmatcher::SearchByBoW(KeyFrame* pKF,Frame &F, vector<MapPoint*> &vpMapPointMatches)
{
    const vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = vector<MapPoint*>(F.N,static_cast<MapPoint*>(NULL));

    const int n1 = pKF->mDescriptors.rows, n2 = F.mDescriptors.rows;
    if (n1 == 0 || n2 == 0) return 0;
    vector<uint8_t> valid1(n1, 0);
    vector<float> a1(n1, 0.f), a2(n2, 0.f);
    for (int i = 0; i < n1; i++) {
        MapPoint *pMP = i < (int)vpMapPointsKF.size() ? vpMapPointsKF[i] : NULL;
        valid1[i] = (pMP && !pMP->isBad()) ? 1 : 0;          // ref: :193-199
        a1[i] = pKF->mvKeysUn[i].angle;                        // ref: :234
    }
    for (int i = 0; i < n2; i++) a2[i] = F.mvKeys[i].angle;   // ref: :238
    const Csr c1 = flatten(pKF->mFeatVec), c2 = flatten(F.mFeatVec);
    const vector<uint8_t> d1 = contiguous(pKF->mDescriptors), d2 = contiguous(F.mDescriptors);
    vector<int32_t> m12(n1), m21(n2);
    int nmatches = 0;
    const int rc = orbhip_search_by_bow(tls.get(), d1.data(), n1, valid1.data(), a1.data(), c1.node.data(),
                                        c1.off.data(), c1.idx.data(), (int)c1.node.size(), d2.data(), n2, NULL,
                                        a2.data(), c2.node.data(), c2.off.data(), c2.idx.data(), (int)c2.node.size(),
                                        TH_LOW, 0, mfNNratio, mbCheckOrientation ? 1 : 0, m12.data(), m21.data(),
                                        &nmatches);
    if (rc != ORBHIP_OK) throw std::runtime_error(std::string("ORBmatcher::SearchByBoW: ") + orbhip_last_error(tls.get()));
    for (int i2 = 0; i2 < n2 && i2 < F.N; i2++)
        if (m21[i2] >= 0) vpMapPointMatches[i2] = vpMapPointsKF[m21[i2]];   // ref: :232
    return nmatches;
}

int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint *> &vpMatches12)
{
    const vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = vector<MapPoint*>(vpMapPoints1.size(),static_cast<MapPoint*>(NULL));

    const int n1 = pKF1->mDescriptors.rows, n2 = pKF2->mDescriptors.rows;
    if (n1 == 0 || n2 == 0) return 0;
    vector<uint8_t> valid1(n1, 0), valid2(n2, 0);
    vector<float> a1(n1, 0.f), a2(n2, 0.f);
    for (int i = 0; i < n1; i++) {
        MapPoint *p = i < (int)vpMapPoints1.size() ? vpMapPoints1[i] : NULL;
        valid1[i] = (p && !p->isBad()) ? 1 : 0;               // ref: :556-560
        a1[i] = pKF1->mvKeysUn[i].angle;
    }
    for (int i = 0; i < n2; i++) {
        MapPoint *p = i < (int)vpMapPoints2.size() ? vpMapPoints2[i] : NULL;
        valid2[i] = (p && !p->isBad()) ? 1 : 0;               // ref: :572-578
        a2[i] = pKF2->mvKeysUn[i].angle;
    }
    const Csr c1 = flatten(pKF1->mFeatVec), c2 = flatten(pKF2->mFeatVec);
    const vector<uint8_t> d1 = contiguous(pKF1->mDescriptors), d2 = contiguous(pKF2->mDescriptors);
    vector<int32_t> m12(n1), m21(n2);
    int nmatches = 0;
    const int rc = orbhip_search_by_bow(tls.get(), d1.data(), n1, valid1.data(), a1.data(), c1.node.data(),
                                        c1.off.data(), c1.idx.data(), (int)c1.node.size(), d2.data(), n2,
                                        valid2.data(), a2.data(), c2.node.data(), c2.off.data(), c2.idx.data(),
                                        (int)c2.node.size(), TH_LOW, 1, mfNNratio, mbCheckOrientation ? 1 : 0,
                                        m12.data(), m21.data(), &nmatches);
    if (rc != ORBHIP_OK) throw std::runtime_error(std::string("ORBmatcher::SearchByBoW: ") + orbhip_last_error(tls.get()));
    for (int i1 = 0; i1 < n1 && i1 < (int)vpMatches12.size(); i1++)
        if (m12[i1] >= 0) vpMatches12[i1] = vpMapPoints2[m12[i1]];           // ref: :602
    return nmatches;
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
