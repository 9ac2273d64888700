// ORBVocabulary.h -- drop-in for the part of the reference's include/ORBVocabulary.h
// (typedef DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> ORBVocabulary) that the ORB hot path uses:
// loadFromBinaryFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1680-1721, called from
// src/System.cc:336-339) and transform(features, BowVector&, FeatureVector&, levelsup) (:1167-1258, called
// from Frame::ComputeBoW src/Frame.cc:739-746 and KeyFrame::ComputeBoW src/KeyFrame.cc:392-400).
// The tree descent runs in liborbhip.so (k_vocab_transform); the BowVector / FeatureVector maps are
// assembled on the host in ascending feature order (the canonical order; the fork's own transform fills
// them from four racing threads, TemplatedVocabulary.h:1202-1211).
#ifndef ORBVOCABULARY_H
#define ORBVOCABULARY_H

#include <map>
#include <string>
#include <vector>

#ifdef ORBHIP_WITH_REFERENCE_HEADERS
#include "Thirdparty/DBoW2/DBoW2/BowVector.h"
#include "Thirdparty/DBoW2/DBoW2/FeatureVector.h"
#include <opencv2/core/core.hpp>
#else
#include "cvlite.h"
#include "slamlite.h"
#endif

struct orbhip_ctx;

namespace ORB_SLAM2
{

class ORBVocabulary
{
public:
    ORBVocabulary();
    ~ORBVocabulary();
    ORBVocabulary(const ORBVocabulary &) = delete;
    ORBVocabulary &operator=(const ORBVocabulary &) = delete;

    // ref: TemplatedVocabulary.h:1564-1647 -- the loader src/System.cc:335-336 picks for a ".txt" vocabulary (the stock
    // ORBvoc.txt).  Returns false if the file cannot be read or is malformed.  Node weights are kept as the doubles of
    // the text (Node::weight), so BowVector values equal the reference's for a text-loaded vocabulary.
    bool loadFromTextFile(const std::string &filename);
    bool loadFromText(const char *text, size_t nbytes);
    // ref: TemplatedVocabulary.h:1680 -- returns false if the file cannot be read or is malformed
    bool loadFromBinaryFile(const std::string &filename);
    // same from memory (e.g. the blob received through orbhip_bcast_blob_device)
    bool loadFromBinaryBlob(const void *blob, size_t nbytes);
    bool empty() const { return mnNodes == 0; }
    unsigned int size() const { return (unsigned int)mnWords; }   // number of words, ref :1109

    // ref: TemplatedVocabulary.h:1167-1258
    void transform(const std::vector<cv::Mat> &features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv,
                   int levelsup) const;

    // the host half of transform (BowVector / FeatureVector from the per-feature word ids, weights and node ids of the
    // descent): Frame::ComputeBoW uses it on the by-products of orbhip_frame_build (ORBextractor::BuiltBoW)
    void assemble(const int *word, const float *weight, const int *node, int n, DBoW2::BowVector &v,
                  DBoW2::FeatureVector &fv) const;
    orbhip_ctx *Context() const { return mpCtx; }

    static void SetDevice(int device);

private:
    orbhip_ctx *mpCtx;
    int mnNodes, mnWords, mK, mL, mScoring, mWeighting;
    std::vector<double> mvWordWeight;   // by word id; only for a text-loaded vocabulary (empty otherwise)
};

}  // namespace ORB_SLAM2

#endif
