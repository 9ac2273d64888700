// ORBmatcher.h -- drop-in replacement of the in-scope part of the reference's
// include/ORBmatcher.h:41-89: same class name, namespace, constructor, constants, static
// DescriptorDistance and both SearchByBoW overloads.  The Hamming work of SearchByBoW runs in
// liborbhip.so (k_bow_match, one wave per shared vocabulary node).
//
// SearchByProjection(Frame&, vector<MapPoint*>&, th), SearchByProjection(CurrentFrame, LastFrame, th,
// bMono) and the relocalisation variant SearchByProjection(CurrentFrame, KeyFrame*,
// sAlreadyFound, th, ORBdist) run their window search on the device grid
// (orbhip_search_by_projection); the pose arithmetic that projects the points stays on the host.
//
// SearchForInitialization (the monocular initialiser's matcher, src/ORBmatcher.cc:405-520) runs as one call
// (orbhip_search_for_initialization): windows and distances in parallel, the owner bookkeeping in the reference's order.
//
// SearchForTriangulation (LocalMapping::CreateNewMapPoints, src/ORBmatcher.cc:657-827): node-grouped matching with the
// epipolar tests, one call (orbhip_search_for_triangulation); the epipole is computed here on the host.
//
// SearchByProjection(KeyFrame*, Scw, ...), SearchBySim3 and both Fuse forms (LoopClosing / LocalMapping, src/ORBmatcher.cc:
// 290-403, 825-1100, 1102-1326): the per-point window search runs on the device (orbhip_window_best /
// orbhip_search_by_projection), the projection arithmetic and the map edits on the host in the reference's order.
// With these every public method of the reference class has a drop-in (tools/diff_dropin_headers.py compares the
// declarations with the reference's header when /root/reference is present).
#ifndef ORBMATCHER_H
#define ORBMATCHER_H

#include <set>
#include <utility>
#include <vector>

#ifdef ORBHIP_WITH_REFERENCE_HEADERS
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include "MapPoint.h"
#include "KeyFrame.h"
#include "Frame.h"
#else
#include "cvlite.h"
#include "slamlite.h"
#endif

struct orbhip_ctx;

namespace ORB_SLAM2
{

class ORBmatcher
{
public:

    ORBmatcher(float nnratio=0.6, bool checkOri=true);
    ~ORBmatcher();

    // Computes the Hamming distance between two ORB descriptors (ref: src/ORBmatcher.cc:1675-1691)
    static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b);

    // (addition) Empties the calling thread's table of device-resident key frames / frames (include/orbhip.h, orbhip_set_*).
    // Stale sets are never used -- a set is checked against the object in hand -- so this only returns their memory early:
    // call it from Tracking::Reset (ref: src/Tracking.cc:2724-2770), where the map and both id counters start over.
    static void DropResidentSets();
    // (addition) At most n key frames / frames stay resident per matcher thread from now on (4 .. 96, the default; orbhip_set_limit):
    // the device-memory budget of a thread's table, least recently used out.  Takes effect in every thread at its next search.
    static void SetResidentSetLimit(int n);

    // Search matches between MapPoints in a KeyFrame and ORB in a Frame.
    // Brute force constrained to ORB that belong to the same vocabulary node (at a certain level)
    // Used in Relocalisation and Loop Detection (ref: src/ORBmatcher.cc:159-288, 522-655)
    int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches);
    int SearchByBoW(KeyFrame *pKF1, KeyFrame* pKF2, std::vector<MapPoint*> &vpMatches12);

    // Search matches between Frame keypoints and projected MapPoints. Returns number of matches
    // Used to track the local map (Tracking) (ref: src/ORBmatcher.cc:45-129)
    int SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th=3);

    // Project MapPoints tracked in last frame into the current frame and search matches.
    // Used to track from previous frame (Tracking) (ref: src/ORBmatcher.cc:1341-1498)
    int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono);

    // Project MapPoints seen in KeyFrame into the Frame and search matches.
    // Used in relocalisation (Tracking) (ref: src/ORBmatcher.cc:1500-1627)
    int SearchByProjection(Frame &CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist);

    // Matching for the Map Initialization (only used in the monocular case) (ref: src/ORBmatcher.cc:405-520)
    int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize=10);

    // Matching to triangulate new MapPoints. Check Epipolar Constraint. (ref: src/ORBmatcher.cc:657-827)
    int SearchForTriangulation(KeyFrame *pKF1, KeyFrame* pKF2, cv::Mat F12,
                               std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo);

    // Project MapPoints using a Similarity Transformation and search matches.
    // Used in loop detection (Loop Closing) (ref: src/ORBmatcher.cc:290-403)
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*> &vpPoints, std::vector<MapPoint*> &vpMatched, int th);

    // Search matches between MapPoints seen in KF1 and KF2 transforming by a Sim3 [s12*R12|t12]
    // In the stereo and RGB-D case, s12=1 (ref: src/ORBmatcher.cc:1102-1326)
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th);

    // Project MapPoints into KeyFrame and search for duplicated MapPoints. (ref: src/ORBmatcher.cc:825-975)
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint *> &vpMapPoints, const float th=3.0);

    // Project MapPoints into KeyFrame using a given Sim3 and search for duplicated MapPoints. (ref: src/ORBmatcher.cc:977-1100)
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint);

    // Device context used for matching.  By default one small context per thread is created on
    // first use (matchers are stack objects in the reference and are used from three threads).
    static void SetDevice(int device);

public:

    static const int TH_LOW;
    static const int TH_HIGH;
    static const int HISTO_LENGTH;

protected:

    float RadiusByViewingCos(const float &viewCos);

    void ComputeThreeMaxima(std::vector<int>* histo, const int L, int &ind1, int &ind2, int &ind3);

    float mfNNratio;
    bool mbCheckOrientation;
};

// MapPoint::ComputeDistinctiveDescriptors (ref: src/MapPoint.cc:283-349) for many map points in one device call -- the
// batched form of the loops LocalMapping runs over a key frame's points (src/LocalMapping.cc ProcessNewKeyFrame,
// CreateNewMapPoints, SearchInNeighbors: "pMP->ComputeDistinctiveDescriptors()" per point).  Bad points, points without
// observations and points whose observers are all bad are left alone, as in the reference.  Returns the number of points
// whose descriptor was set.  Inside the reference tree MapPoint needs a setter for its protected mDescriptor
// (INTEGRATION.md).
int ComputeDistinctiveDescriptors(const std::vector<MapPoint*> &vpMapPoints);

}// namespace ORB_SLAM

#endif // ORBMATCHER_H
