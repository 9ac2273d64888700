// test_nothrow_dropin.cpp -- the drop-in classes keep the reference's no-throw contract (include/orbhip/hiperror.h):
// the reference's operator() / Search* never throw and Tracking / LocalMapping / LoopClosing have no try block
// (src/ORBextractor.cc:1048-1052, src/Tracking.cc:935-976).  Failures must come back as "nothing found" plus a message.
// Also the text loader of the vocabulary as src/System.cc:335-336 calls it.  Prints key=value lines for
// tests/test_gpu_dropin.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "ORBVocabulary.h"
#include "hiperror.h"

using namespace ORB_SLAM2;

static std::vector<unsigned char> slurp(const char *path)
{
    std::vector<unsigned char> v;
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n);
    if (fread(v.data(), 1, n, f) != (size_t)n) exit(2);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s w h frame.raw voc.txt desc.bin\n", argv[0]); return 2; }
    const int w = atoi(argv[1]), h = atoi(argv[2]);
    std::vector<unsigned char> raw = slurp(argv[3]);
    std::vector<unsigned char> dsc = slurp(argv[5]);      // n x 32 descriptor bytes for the vocabulary transform

    ORBextractor ex(1000, 1.2f, 8, 20, 7);
    std::vector<cv::KeyPoint> keys(5);                    // stale content must be cleared
    cv::Mat desc(3, 32, CV_8U);
    // 1. an image whose top pyramid level has no 30-pixel cell (the reference divides by zero there): an error, not a throw
    std::vector<unsigned char> tiny(90 * 90, 100);
    cv::Mat imTiny(90, 90, CV_8UC1, tiny.data());
    const unsigned long e0 = OrbHipErrorCount();
    ex(imTiny, cv::Mat(), keys, desc);
    printf("tiny_keys=%d\ntiny_desc_rows=%d\ntiny_errors=%lu\ntiny_msg=%s\n", (int)keys.size(), desc.rows, OrbHipErrorCount() - e0,
           OrbHipLastError());
    // 2. the same extractor object keeps working afterwards
    cv::Mat im(h, w, CV_8UC1, raw.data());
    ex(im, cv::Mat(), keys, desc);
    printf("good_keys=%d\ngood_desc_rows=%d\n", (int)keys.size(), desc.rows);
    // 3. empty image: silent return, outputs untouched (ref: src/ORBextractor.cc:1048-1049)
    const int before = (int)keys.size();
    ex(cv::Mat(), cv::Mat(), keys, desc);
    printf("empty_keys_unchanged=%d\n", (int)keys.size() == before ? 1 : 0);
    // 4. constructor arguments liborbhip cannot run (scale factor 1: the reference's own quota formula divides 0 by 0)
    ORBextractor bad(1000, 1.0f, 8, 20, 7);
    std::vector<cv::KeyPoint> k2(2);
    cv::Mat d2(2, 32, CV_8U);
    bad(im, cv::Mat(), k2, d2);
    printf("bad_keys=%d\nbad_desc_rows=%d\n", (int)k2.size(), d2.rows);
    // 5. matcher: inconsistent arguments
    {
        Frame F1, F2;
        F1.N = F2.N = (int)keys.size();
        F1.mvKeys = F1.mvKeysUn = keys;
        F2.mvKeys = F2.mvKeysUn = keys;
        F1.mDescriptors = desc.clone();
        F2.mDescriptors = desc.clone();
        std::vector<cv::Point2f> prev(3);                 // shorter than F1.mvKeysUn
        std::vector<int> m12;
        ORBmatcher matcher(0.9f, true);
        const int n = matcher.SearchForInitialization(F1, F2, prev, m12, 100);
        int assigned = 0;
        for (size_t i = 0; i < m12.size(); i++) assigned += m12[i] >= 0;
        printf("init_short_prev=%d\ninit_assigned=%d\ninit_size=%d\n", n, assigned, (int)m12.size());
    }
    // 6. vocabulary: missing file, malformed text, then the text fixture (System.cc:335-336)
    ORBVocabulary voc;
    printf("voc_missing=%d\n", voc.loadFromBinaryFile("/nonexistent/ORBvoc.bin") ? 1 : 0);
    printf("voc_missing_txt=%d\n", voc.loadFromTextFile("/nonexistent/ORBvoc.txt") ? 1 : 0);
    printf("voc_bad_txt=%d\n", voc.loadFromText("42 1  0 0\n", 10) ? 1 : 0);
    const bool textOk = voc.loadFromTextFile(argv[4]);
    printf("voc_text=%d\nvoc_words=%u\n", textOk ? 1 : 0, voc.size());
    const int nd = (int)(dsc.size() / 32);
    std::vector<cv::Mat> feats(nd);
    for (int i = 0; i < nd; i++) feats[i] = cv::Mat(1, 32, CV_8U, dsc.data() + (size_t)i * 32);
    DBoW2::BowVector bv;
    DBoW2::FeatureVector fv;
    voc.transform(feats, bv, fv, 1);
    printf("bow_n=%d\n", (int)bv.size());
    for (DBoW2::BowVector::const_iterator it = bv.begin(); it != bv.end(); ++it) printf("bow=%u %.17g\n", (unsigned)it->first, it->second);
    for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
        printf("fv=%u", (unsigned)it->first);
        for (size_t k = 0; k < it->second.size(); k++) printf(" %u", it->second[k]);
        printf("\n");
    }
    printf("errors_total=%lu\ndone=1\n", OrbHipErrorCount());
    return 0;
}
