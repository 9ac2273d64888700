// latency_dropin.cpp -- per-call latency of the extractor as a C++ caller sees it (no Python in the loop):
//   (a) orbhip_extract through the C ABI (host image in, keypoints + descriptors out),
//   (b) ORB_SLAM2::ORBextractor::operator() without and with mvImagePyramid on the host (Tracking.cc / Frame.cc usage).
//   (c) with a vocabulary file and a second frame: what Tracking::TrackReferenceKeyFrame adds per frame --
//       ORBVocabulary::transform (Frame::ComputeBoW, src/Frame.cc:739-746) and ORBmatcher::SearchByBoW(KF, F)
//       (src/Tracking.cc:1881-1885).
// usage: latency_dropin w h nfeatures frames.raw [iterations [voc.bin]]      (frames.raw: one frame, or two for (c))
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ORBVocabulary.h"
#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "orbhip.h"

using namespace ORB_SLAM2;
typedef std::chrono::steady_clock Clock;

static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s w h nfeatures frame.raw [iterations]\n", argv[0]); return 2; }
    const int w = atoi(argv[1]), h = atoi(argv[2]), nf = atoi(argv[3]), iters = argc > 5 ? atoi(argv[5]) : 2000;
    std::vector<unsigned char> pix((size_t)w * h), pix2((size_t)w * h);
    FILE *f = fopen(argv[4], "rb");
    if (!f || fread(pix.data(), 1, pix.size(), f) != pix.size()) { perror(argv[4]); return 2; }
    const bool two = fread(pix2.data(), 1, pix2.size(), f) == pix2.size();
    fclose(f);

    // (a) C ABI
    orbhip_ctx *c = orbhip_create(0, nf, 1.2f, 8, 20, 7, w, h, 1);
    if (!c) { fprintf(stderr, "orbhip_create: %s\n", orbhip_last_error(nullptr)); return 1; }
    const int cap = orbhip_max_keypoints(c);
    std::vector<orbhip_keypoint> kps(cap);
    std::vector<unsigned char> desc((size_t)cap * 32);
    int n = 0;
    float t[3];
    for (int i = 0; i < 20; i++) orbhip_extract(c, pix.data(), w, h, w, kps.data(), desc.data(), cap, &n, t);
    Clock::time_point t0 = Clock::now();
    for (int i = 0; i < iters; i++)
        if (orbhip_extract(c, pix.data(), w, h, w, kps.data(), desc.data(), cap, &n, t) != ORBHIP_OK) return 1;
    const double capi = ms_since(t0) / iters;
    // (d) the Frame constructor's device work: four calls (extract, UndistortKeyPoints, AssignFeaturesToGrid, transform) against
    //     ONE (orbhip_frame_build), EuRoC cam0 calibration, the vocabulary of argv[6]
    double four = -1, one = -1, oneNoBow = -1;
    if (argc > 6) {
        std::vector<unsigned char> blob;
        if (FILE *vf = fopen(argv[6], "rb")) {
            fseek(vf, 0, SEEK_END);
            blob.resize((size_t)ftell(vf));
            fseek(vf, 0, SEEK_SET);
            if (fread(blob.data(), 1, blob.size(), vf) != blob.size()) blob.clear();
            fclose(vf);
        }
        if (!blob.empty() && orbhip_vocab_load(c, blob.data(), blob.size()) == ORBHIP_OK) {
            orbhip_frame_params fp = {{458.654f, 0, 367.215f, 0, 457.296f, 248.375f, 0, 0, 1}, {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f},
                                      4, 0.f, 0.f, 64.f / w, 48.f / h, 4};
            std::vector<orbhip_keypoint> kun(cap);
            std::vector<int32_t> off(ORBHIP_GRID_CELLS + 1), idx(cap), word(cap), node(cap);
            std::vector<float> wt(cap);
            auto sep = [&]() {
                return orbhip_extract(c, pix.data(), w, h, w, kps.data(), desc.data(), cap, &n, nullptr) ||
                       orbhip_undistort_keypoints(c, kps.data(), n, fp.K, fp.dist, fp.ndist, fp.K, kun.data()) ||
                       orbhip_grid_build(c, kun.data(), n, fp.min_x, fp.min_y, fp.inv_w, fp.inv_h, off.data(), idx.data()) ||
                       orbhip_vocab_transform(c, desc.data(), n, 4, word.data(), wt.data(), node.data());
            };
            auto fb = [&]() {
                return orbhip_frame_build(c, pix.data(), w, h, w, &fp, kps.data(), kun.data(), desc.data(), cap, &n, off.data(), idx.data(),
                                          word.data(), wt.data(), node.data());
            };
            for (int i = 0; i < 20; i++) if (sep() || fb()) { fprintf(stderr, "frame build: %s\n", orbhip_last_error(c)); return 1; }
            t0 = Clock::now();
            for (int i = 0; i < iters; i++) sep();
            four = ms_since(t0) / iters;
            t0 = Clock::now();
            for (int i = 0; i < iters; i++) fb();
            one = ms_since(t0) / iters;
            fp.levelsup = -1;
            for (int i = 0; i < 20; i++) fb();
            t0 = Clock::now();
            for (int i = 0; i < iters; i++) fb();
            oneNoBow = ms_since(t0) / iters;
        }
    }
    orbhip_destroy(c);

    // (b) the drop-in class
    double cls[2];
    int nk = 0;
    for (int mode = 0; mode < 2; mode++) {
        ORBextractor ex(nf, 1.2f, 8, 20, 7);
        ex.SetPyramidDownload(mode == 1);
        cv::Mat im(h, w, CV_8UC1, (void *)pix.data());
        std::vector<cv::KeyPoint> keys;
        cv::Mat d;
        for (int i = 0; i < 20; i++) ex(im, cv::Mat(), keys, d);
        t0 = Clock::now();
        for (int i = 0; i < iters; i++) ex(im, cv::Mat(), keys, d);
        cls[mode] = ms_since(t0) / iters;
        nk = (int)keys.size();
    }
    printf("{\"w\": %d, \"h\": %d, \"nfeatures\": %d, \"keypoints\": %d, \"iterations\": %d, \"orbhip_extract_ms\": %.4f, "
           "\"dropin_operator_ms\": %.4f, \"dropin_operator_with_pyramid_ms\": %.4f",
           w, h, nf, nk, iters, capi, cls[0], cls[1]);
    if (one >= 0)
        printf(", \"frame_four_calls_ms\": %.4f, \"orbhip_frame_build_ms\": %.4f, \"orbhip_frame_build_no_transform_ms\": %.4f", four, one, oneNoBow);
    if (argc > 6 && two) {
        ORBVocabulary voc;
        if (!voc.loadFromBinaryFile(argv[6])) { fprintf(stderr, "cannot load %s\n", argv[6]); return 1; }
        ORBextractor ex(nf, 1.2f, 8, 20, 7);
        ex.SetPyramidDownload(false);
        KeyFrame kf;
        Frame F;
        std::vector<MapPoint> points(8192);
        {
            cv::Mat im1(h, w, CV_8UC1, (void *)pix.data()), im2(h, w, CV_8UC1, (void *)pix2.data());
            ex(im1, cv::Mat(), kf.mvKeys, kf.mDescriptors);
            kf.mvKeysUn = kf.mvKeys;
            kf.mvpMapPoints.resize(kf.mvKeys.size());
            for (size_t i = 0; i < kf.mvKeys.size(); i++) kf.mvpMapPoints[i] = &points[i];
            ex(im2, cv::Mat(), F.mvKeys, F.mDescriptors);
            F.mvKeysUn = F.mvKeys;
            F.N = (int)F.mvKeys.size();
        }
        // Converter::toDescriptorVector (src/Converter.cc:163-171): one 1x32 Mat per descriptor row
        std::vector<cv::Mat> d1, d2;
        for (int j = 0; j < kf.mDescriptors.rows; j++) d1.push_back(kf.mDescriptors.row(j));
        for (int j = 0; j < F.mDescriptors.rows; j++) d2.push_back(F.mDescriptors.row(j));
        DBoW2::BowVector bv;
        voc.transform(d1, bv, kf.mFeatVec, 4);
        const int it2 = iters / 4 > 0 ? iters / 4 : 1;
        for (int i = 0; i < 5; i++) voc.transform(d2, bv, F.mFeatVec, 4);
        t0 = Clock::now();
        for (int i = 0; i < it2; i++) voc.transform(d2, bv, F.mFeatVec, 4);
        const double tr = ms_since(t0) / it2;
        ORBmatcher matcher(0.7, true);
        std::vector<MapPoint *> vpMapPointMatches;
        int nm = 0;
        for (int i = 0; i < 5; i++) nm = matcher.SearchByBoW(&kf, F, vpMapPointMatches);
        t0 = Clock::now();
        for (int i = 0; i < it2; i++) nm = matcher.SearchByBoW(&kf, F, vpMapPointMatches);
        const double sb = ms_since(t0) / it2;
        printf(", \"vocabulary_words\": %u, \"transform_ms\": %.4f, \"search_by_bow_ms\": %.4f, \"bow_matches\": %d", voc.size(), tr, sb, nm);
    }
    printf("}\n");
    return n == nk ? 0 : 1;
}
