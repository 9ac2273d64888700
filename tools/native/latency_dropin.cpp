// latency_dropin.cpp -- per-call latency of the extractor as a C++ caller sees it (no Python in the loop):
//   (a) orbhip_extract through the C ABI (host image in, keypoints + descriptors out),
//   (b) ORB_SLAM2::ORBextractor::operator() without and with mvImagePyramid on the host (Tracking.cc / Frame.cc usage).
// usage: latency_dropin w h nfeatures frame.raw [iterations]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ORBextractor.h"
#include "orbhip.h"

using namespace ORB_SLAM2;
typedef std::chrono::steady_clock Clock;

static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s w h nfeatures frame.raw [iterations]\n", argv[0]); return 2; }
    const int w = atoi(argv[1]), h = atoi(argv[2]), nf = atoi(argv[3]), iters = argc > 5 ? atoi(argv[5]) : 2000;
    std::vector<unsigned char> pix((size_t)w * h);
    FILE *f = fopen(argv[4], "rb");
    if (!f || fread(pix.data(), 1, pix.size(), f) != pix.size()) { perror(argv[4]); return 2; }
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
           "\"dropin_operator_ms\": %.4f, \"dropin_operator_with_pyramid_ms\": %.4f}\n",
           w, h, nf, nk, iters, capi, cls[0], cls[1]);
    return n == nk ? 0 : 1;
}
