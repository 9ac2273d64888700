// test_stereo_dropin.cpp -- the stereo Frame constructor's sequence (ref: src/Frame.cc:413-437):
//   thread threadLeft(&Frame::ExtractORB,this,0,imLeft); thread threadRight(&Frame::ExtractORB,this,1,imRight);
//   join both; N = mvKeys.size(); ComputeStereoMatches();
// with the drop-in ORBextractor (two instances used concurrently from two host threads) and the
// device-side Frame::ComputeStereoMatches.  Output is raw binary for tests/test_gpu_dropin.py.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "ORBextractor.h"
#include "slamlite.h"

using namespace ORB_SLAM2;

static void ExtractORB(ORBextractor *ex, const unsigned char *pix, int w, int h, std::vector<cv::KeyPoint> *keys,
                       cv::Mat *desc)
{
    cv::Mat im(h, w, CV_8UC1, (void *)pix);
    (*ex)(im, cv::Mat(), *keys, *desc);
}

int main(int argc, char **argv)
{
    if (argc < 8) { fprintf(stderr, "usage: %s w h nfeatures mb mbf pair.raw out.bin\n", argv[0]); return 2; }
    const int w = atoi(argv[1]), h = atoi(argv[2]), nf = atoi(argv[3]);
    std::vector<unsigned char> raw((size_t)2 * w * h);
    FILE *f = fopen(argv[6], "rb");
    if (!f || fread(raw.data(), 1, raw.size(), f) != raw.size()) { perror(argv[6]); return 2; }
    fclose(f);

    Frame F;
    F.mb = (float)atof(argv[4]);
    F.mbf = (float)atof(argv[5]);
    F.mpORBextractorLeft = new ORBextractor(nf, 1.2f, 8, 20, 7);
    F.mpORBextractorRight = new ORBextractor(nf, 1.2f, 8, 20, 7);
    F.mpORBextractorLeft->SetPyramidDownload(false);      // the pyramids stay on the device
    F.mpORBextractorRight->SetPyramidDownload(false);
    for (int rep = 0; rep < 2; rep++) {                     // twice: contexts are reused frame after frame
        std::thread threadLeft(ExtractORB, F.mpORBextractorLeft, raw.data(), w, h, &F.mvKeys, &F.mDescriptors);
        std::thread threadRight(ExtractORB, F.mpORBextractorRight, raw.data() + (size_t)w * h, w, h, &F.mvKeysRight,
                                &F.mDescriptorsRight);
        threadLeft.join();
        threadRight.join();
        F.N = (int)F.mvKeys.size();
        F.ComputeStereoMatches();
    }
    FILE *out = fopen(argv[7], "wb");
    int nr = (int)F.mvKeysRight.size();
    fwrite(&F.N, 4, 1, out);
    fwrite(&nr, 4, 1, out);
    fwrite(F.mvKeys.data(), sizeof(cv::KeyPoint), F.N, out);
    fwrite(F.mvuRight.data(), 4, F.N, out);
    fwrite(F.mvDepth.data(), 4, F.N, out);
    fclose(out);
    delete F.mpORBextractorLeft;
    delete F.mpORBextractorRight;
    return 0;
}
