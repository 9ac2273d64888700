// ORBVocabulary.cc -- host side of the drop-in ORB_SLAM2::ORBVocabulary (include/orbhip/ORBVocabulary.h).
#include "ORBVocabulary.h"

#include <cmath>
#include <cstdio>
#include <cstring>

#include "hiperror.h"
#include "orbhip.h"

namespace ORB_SLAM2
{

static int g_voc_device = 0;
void ORBVocabulary::SetDevice(int device) { g_voc_device = device; }

ORBVocabulary::ORBVocabulary() : mpCtx(nullptr), mnNodes(0), mnWords(0), mK(0), mL(0), mScoring(0), mWeighting(0) {}

ORBVocabulary::~ORBVocabulary()
{
    if (mpCtx) orbhip_destroy(mpCtx);
}

bool ORBVocabulary::loadFromBinaryBlob(const void *blob, size_t nbytes)
{
    if (!mpCtx) {
        mpCtx = orbhip_create(g_voc_device, 50, 1.2f, 1, 20, 7, 128, 128, 1);
        if (!mpCtx) return hipdetail::Fail("ORBVocabulary (device context)", orbhip_last_error(nullptr));   // = "failed to load", src/System.cc:340-346
    }
    mvWordWeight.clear();
    if (orbhip_vocab_load(mpCtx, blob, nbytes) != ORBHIP_OK) return false;
    orbhip_vocab_info(mpCtx, &mK, &mL, &mScoring, &mWeighting, &mnNodes, &mnWords);
    return true;
}

static bool slurp(const std::string &filename, std::vector<unsigned char> &buf)
{
    FILE *f = fopen(filename.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    const bool ok = n > 0 && fread(buf.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

bool ORBVocabulary::loadFromTextFile(const std::string &filename)
{
    std::vector<unsigned char> buf;
    return slurp(filename, buf) && loadFromText((const char *)buf.data(), buf.size());
}

bool ORBVocabulary::loadFromText(const char *text, size_t nbytes)
{
    size_t need = 0;
    if (orbhip_vocab_text_to_binary(text, nbytes, NULL, 0, &need, NULL, 0) != ORBHIP_OK) return false;   // ref :1585-1589
    const size_t n = (need - 24) / 41;
    std::vector<unsigned char> blob(need);
    std::vector<double> w(n);
    if (orbhip_vocab_text_to_binary(text, nbytes, blob.data(), blob.size(), &need, w.data(), w.size()) != ORBHIP_OK) return false;
    if (!loadFromBinaryBlob(blob.data(), blob.size())) return false;
    // words are numbered in node order among the leaves (ref :1633-1640); the weight of word i as the text's double
    mvWordWeight.clear();
    for (size_t i = 0; i < n; i++)
        if (blob[24 + i * 41 + 40]) mvWordWeight.push_back(w[i]);
    return true;
}

bool ORBVocabulary::loadFromBinaryFile(const std::string &filename)
{
    std::vector<unsigned char> buf;
    return slurp(filename, buf) && loadFromBinaryBlob(buf.data(), buf.size());
}

void ORBVocabulary::transform(const std::vector<cv::Mat> &features, DBoW2::BowVector &v, DBoW2::FeatureVector &fv,
                              int levelsup) const
{
    v.clear();
    fv.clear();
    if (empty() || features.empty()) return;           // ref :1175-1178
    const int n = (int)features.size();
    std::vector<unsigned char> desc((size_t)n * 32);
    for (int i = 0; i < n; i++) memcpy(&desc[(size_t)i * 32], features[i].ptr(0), 32);
    std::vector<int32_t> word(n), node(n);
    std::vector<float> weight(n);
    if (orbhip_vocab_transform(mpCtx, desc.data(), n, levelsup, word.data(), weight.data(), node.data()) != ORBHIP_OK)
    {
        hipdetail::Fail("ORBVocabulary::transform", orbhip_last_error(mpCtx));
        return;                   // empty BowVector / FeatureVector
    }
    assemble(word.data(), weight.data(), node.data(), n, v, fv);
}

void ORBVocabulary::assemble(const int *word, const float *weight, const int *node, int n, DBoW2::BowVector &v,
                             DBoW2::FeatureVector &fv) const
{
    v.clear();
    fv.clear();
    // BowVector.h: WeightingType TF_IDF=0, TF=1, IDF=2, BINARY=3; ScoringType L1_NORM=0, L2_NORM=1,
    // CHI_SQUARE=2, KL=3, BHATTACHARYYA=4, DOT_PRODUCT=5
    const bool accumulate = (mWeighting == 0 || mWeighting == 1);
    const bool must = (mScoring != 5);
    const bool textWeights = !mvWordWeight.empty();     // text-loaded: Node::weight is the text's double
    for (int i = 0; i < n; i++) {
        const double wi = textWeights ? mvWordWeight[(size_t)word[i]] : (double)weight[i];
        if (!(wi > 0)) continue;                        // "stopped" word, ref :1334
        const DBoW2::WordId id = (DBoW2::WordId)word[i];
        DBoW2::BowVector::iterator it = v.lower_bound(id);
        if (it != v.end() && !(v.key_comp()(id, it->first))) {
            if (accumulate) it->second += wi;           // addWeight; addIfNotExist keeps the first
        } else {
            v.insert(it, DBoW2::BowVector::value_type(id, wi));
        }
        fv.addFeature((DBoW2::NodeId)node[i], (unsigned int)i);
    }
    if (accumulate && !v.empty() && !must) {            // ref :1226-1232
        const double nd = (double)v.size();
        for (DBoW2::BowVector::iterator it = v.begin(); it != v.end(); ++it) it->second /= nd;
    }
    if (must) {                                         // BowVector::normalize, BowVector.cpp:62-86
        double norm = 0.0;
        if (mScoring == 1) {
            for (DBoW2::BowVector::iterator it = v.begin(); it != v.end(); ++it) norm += it->second * it->second;
            norm = sqrt(norm);
        } else {
            for (DBoW2::BowVector::iterator it = v.begin(); it != v.end(); ++it) norm += fabs(it->second);
        }
        if (norm > 0.0)
            for (DBoW2::BowVector::iterator it = v.begin(); it != v.end(); ++it) it->second /= norm;
    }
}

}  // namespace ORB_SLAM2
