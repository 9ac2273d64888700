// test_dropin.cpp -- uses the drop-in C++ classes exactly the way the reference's callers do:
//   Tracking.cc:816-822   new ORBextractor(nFeatures, fScaleFactor, nLevels, fIniThFAST, fMinThFAST)
//   Frame.cc:591-597      (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors)
//   Frame.cc:526-532      the scale getters
//   Frame.cc:817          mpORBextractorLeft->mvImagePyramid[l]
//   Tracking.cc:1881-1885 ORBmatcher matcher(0.7,true); matcher.SearchByBoW(pKF, F, vpMapPointMatches)
//   LoopClosing.cc:291    ORBmatcher matcher(0.75,true); matcher.SearchByBoW(pKF1, pKF2, vpMatches12)
//   System.cc:336-339     mpVocabulary->loadFromBinaryFile(strVocFile)
//   Frame.cc:739-746      mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)
// Inputs/outputs are raw binary files so that tests/test_gpu_dropin.py can compare with the oracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"
#include "ORBVocabulary.h"

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

static void extract(ORBextractor *ex, const unsigned char *pix, int w, int h, std::vector<cv::KeyPoint> &keys,
                    cv::Mat &desc)
{
    cv::Mat im(h, w, CV_8UC1, (void *)pix);
    (*ex)(im, cv::Mat(), keys, desc);
}

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s w h nfeatures frames.raw groups.bin out.bin\n", argv[0]); return 2; }
    const int w = atoi(argv[1]), h = atoi(argv[2]), nf = atoi(argv[3]);
    std::vector<unsigned char> raw = slurp(argv[4]);     // two frames, w*h each
    std::vector<unsigned char> grp = slurp(argv[5]);     // int32: node id per feature slot (enough entries)
    const int *nodeOf = (const int *)grp.data();
    const int ngrp = (int)(grp.size() / 4);
    FILE *out = fopen(argv[6], "wb");

    ORBextractor *mpORBextractorLeft = new ORBextractor(nf, 1.2f, 8, 20, 7);
    // getters as read by the Frame constructor
    int nlev = mpORBextractorLeft->GetLevels();
    float sf = mpORBextractorLeft->GetScaleFactor();
    std::vector<float> sfs = mpORBextractorLeft->GetScaleFactors(), inv = mpORBextractorLeft->GetInverseScaleFactors(),
                       s2 = mpORBextractorLeft->GetScaleSigmaSquares(), is2 = mpORBextractorLeft->GetInverseScaleSigmaSquares();
    fwrite(&nlev, 4, 1, out);
    fwrite(&sf, 4, 1, out);
    fwrite(sfs.data(), 4, nlev, out);
    fwrite(inv.data(), 4, nlev, out);
    fwrite(s2.data(), 4, nlev, out);
    fwrite(is2.data(), 4, nlev, out);

    KeyFrame kf, kf2;
    Frame F;
    std::vector<MapPoint> points(8192);
    for (int fi = 0; fi < 2; fi++) {
        std::vector<cv::KeyPoint> mvKeys;
        cv::Mat mDescriptors;
        extract(mpORBextractorLeft, raw.data() + (size_t)fi * w * h, w, h, mvKeys, mDescriptors);
        int n = (int)mvKeys.size();
        fwrite(&n, 4, 1, out);
        fwrite(mvKeys.data(), sizeof(cv::KeyPoint), n, out);
        for (int i = 0; i < n; i++) fwrite(mDescriptors.ptr(i), 1, 32, out);
        // mvImagePyramid as ComputeStereoMatches reads it
        for (int l = 0; l < nlev; l++) {
            const cv::Mat &lv = mpORBextractorLeft->mvImagePyramid[l];
            int dims[2] = {lv.cols, lv.rows};
            fwrite(dims, 4, 2, out);
            unsigned long long sum = 0;   // position-weighted checksum: sum pix[i] * (i % 251 + 1)
            for (int y = 0; y < lv.rows; y++)
                for (int x = 0; x < lv.cols; x++) {
                    const unsigned long long i = (unsigned long long)y * lv.cols + x;
                    sum += (unsigned long long)lv.at<uchar>(y, x) * (i % 251 + 1);
                }
            fwrite(&sum, 8, 1, out);
        }
        double t[3] = {mpORBextractorLeft->GetTimeOfComputePyramid(), mpORBextractorLeft->GetTimeOfComputeKeyPointsOctTree(),
                       mpORBextractorLeft->GetTImeOfComputeDescriptor()};
        fwrite(t, 8, 3, out);
        if (fi == 0) {
            kf.mvKeys = kf.mvKeysUn = mvKeys;
            kf.mDescriptors = mDescriptors.clone();
            kf.mvpMapPoints.resize(n);
            for (int i = 0; i < n; i++) {
                kf.mvpMapPoints[i] = (i % 7 == 3) ? NULL : &points[i];     // some features without a MapPoint
                if (i % 11 == 5) points[i].SetBadFlag();                   // some bad MapPoints
                if (i < ngrp) kf.mFeatVec.addFeature(nodeOf[i], i);
            }
        } else {
            F.N = n;
            F.mvKeys = F.mvKeysUn = mvKeys;
            F.mDescriptors = mDescriptors.clone();
            kf2.mvKeys = kf2.mvKeysUn = mvKeys;
            kf2.mDescriptors = mDescriptors.clone();
            kf2.mvpMapPoints.resize(n);
            for (int i = 0; i < n; i++) {
                if (ngrp >= 4096 + i + 1) {
                    F.mFeatVec.addFeature(nodeOf[4096 + i], i);
                    kf2.mFeatVec.addFeature(nodeOf[4096 + i], i);
                }
                kf2.mvpMapPoints[i] = (i % 5 == 1) ? NULL : &points[4096 + i];
            }
        }
    }
    {
        ORBmatcher matcher(0.7, true);
        std::vector<MapPoint *> vpMapPointMatches;
        int nmatches = matcher.SearchByBoW(&kf, F, vpMapPointMatches);
        int n = (int)vpMapPointMatches.size();
        fwrite(&nmatches, 4, 1, out);
        fwrite(&n, 4, 1, out);
        for (int i = 0; i < n; i++) {
            int v = vpMapPointMatches[i] ? (int)(vpMapPointMatches[i] - &points[0]) : -1;
            fwrite(&v, 4, 1, out);
        }
    }
    {
        ORBmatcher matcher(0.75, true);
        std::vector<MapPoint *> vpMatches12;
        int nmatches = matcher.SearchByBoW(&kf, &kf2, vpMatches12);
        int n = (int)vpMatches12.size();
        fwrite(&nmatches, 4, 1, out);
        fwrite(&n, 4, 1, out);
        for (int i = 0; i < n; i++) {
            int v = vpMatches12[i] ? (int)(vpMatches12[i] - &points[0]) : -1;
            fwrite(&v, 4, 1, out);
        }
    }
    {
        cv::Mat a = kf.mDescriptors.row(0), b = F.mDescriptors.row(0);
        int d = ORBmatcher::DescriptorDistance(a, b);
        fwrite(&d, 4, 1, out);
    }
    {
        // LocalMapping::CreateNewMapPoints: ORBmatcher matcher(0.6,false); matcher.SearchForTriangulation(mpCurrentKeyFrame,
        // pKF2, F12, vMatchedIndices, false) -- here with the orientation check on as well; monocular key frames
        kf.N = (int)kf.mvKeysUn.size();
        kf2.N = (int)kf2.mvKeysUn.size();
        const float K[4] = {458.654f, 457.296f, 367.215f, 248.375f};
        for (int which = 0; which < 2; which++) {
            KeyFrame &k = which ? kf2 : kf;
            k.fx = K[0]; k.fy = K[1]; k.cx = K[2]; k.cy = K[3];
            k.mvuRight = std::vector<float>(k.N, -1.0f);
            k.mvScaleFactors = sfs;
            k.mvLevelSigma2 = s2;
            k.Tcw = cv::Mat::zeros(4, 4, CV_32F);
            for (int d = 0; d < 4; d++) k.Tcw.at<float>(d, d) = 1.0f;
            k.Ow = cv::Mat::zeros(3, 1, CV_32F);
        }
        // key frame 2: a little rotation about y and a baseline along x (Tcw given, Ow = -R' t)
        const float ang = 0.01f, tx = -0.2f, ty = 0.01f, tz = 0.05f;
        kf2.Tcw.at<float>(0, 0) = cosf(ang); kf2.Tcw.at<float>(0, 2) = sinf(ang);
        kf2.Tcw.at<float>(2, 0) = -sinf(ang); kf2.Tcw.at<float>(2, 2) = cosf(ang);
        kf2.Tcw.at<float>(0, 3) = tx; kf2.Tcw.at<float>(1, 3) = ty; kf2.Tcw.at<float>(2, 3) = tz;
        cv::Mat F12(3, 3, CV_32F);
        const float Fv[9] = {0.f, 0.f, 0.f, 0.f, 0.f, -1.f, 0.f, 1.f, 0.f};   // epipolar lines = image rows
        for (int i = 0; i < 9; i++) F12.at<float>(i / 3, i % 3) = Fv[i];
        for (int ori = 0; ori < 2; ori++) {
            ORBmatcher matcher(0.6, ori != 0);
            std::vector<std::pair<size_t, size_t> > vMatchedIndices;
            int nmatches = matcher.SearchForTriangulation(&kf, &kf2, F12, vMatchedIndices, false);
            int n = (int)vMatchedIndices.size();
            fwrite(&nmatches, 4, 1, out);
            fwrite(&n, 4, 1, out);
            for (int i = 0; i < n; i++) {
                int v[2] = {(int)vMatchedIndices[i].first, (int)vMatchedIndices[i].second};
                fwrite(v, 4, 2, out);
            }
        }
    }
    if (argc > 7) {
        ORBVocabulary *mpVocabulary = new ORBVocabulary();
        const bool bVocLoad = mpVocabulary->loadFromBinaryFile(argv[7]);
        int okflag = bVocLoad ? 1 : 0, nwords = (int)mpVocabulary->size();
        fwrite(&okflag, 4, 1, out);
        fwrite(&nwords, 4, 1, out);
        // Converter::toDescriptorVector (src/Converter.cc:163-171): one 1x32 Mat per descriptor row
        std::vector<cv::Mat> vCurrentDesc;
        for (int j = 0; j < kf.mDescriptors.rows; j++) vCurrentDesc.push_back(kf.mDescriptors.row(j));
        DBoW2::BowVector mBowVec;
        DBoW2::FeatureVector mFeatVec;
        mpVocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4);
        int nb = (int)mBowVec.size();
        fwrite(&nb, 4, 1, out);
        for (DBoW2::BowVector::const_iterator it = mBowVec.begin(); it != mBowVec.end(); ++it) {
            int w = (int)it->first;
            double val = it->second;
            fwrite(&w, 4, 1, out);
            fwrite(&val, 8, 1, out);
        }
        int nfv = (int)mFeatVec.size();
        fwrite(&nfv, 4, 1, out);
        for (DBoW2::FeatureVector::const_iterator it = mFeatVec.begin(); it != mFeatVec.end(); ++it) {
            int node = (int)it->first, cnt = (int)it->second.size();
            fwrite(&node, 4, 1, out);
            fwrite(&cnt, 4, 1, out);
            for (int j = 0; j < cnt; j++) {
                int idx = (int)it->second[j];
                fwrite(&idx, 4, 1, out);
            }
        }
        delete mpVocabulary;
    }
    delete mpORBextractorLeft;
    fclose(out);
    return 0;
}
