// test_guided_dropin.cpp -- the tracking front end's use of the grid and the guided search with the drop-in
// classes (ref: src/Frame.cc:518-572 frame construction -> AssignFeaturesToGrid; src/Tracking.cc
// TrackWithMotionModel -> matcher.SearchByProjection(mCurrentFrame, mLastFrame, th, bMono); SearchLocalPoints
// -> matcher.SearchByProjection(mCurrentFrame, vpMapPoints, th)).  Raw binary in/out for tests/test_gpu_dropin.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <set>
#include <vector>

#include "ORBextractor.h"
#include "ORBmatcher.h"

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

struct Header {
    float fx, fy, cx, cy, minX, maxX, minY, maxY, mb, mbf, thLast, thLocal, bMono;
    float TcwLast[16], TcwCur[16];
    float D[4];
};
struct MpRec {
    float X, Y, Z, nObs, outlier, projX, projY, projXR, viewCos, level, inView, bad, minDist, maxDist;
};

static void buildFrame(Frame &F, ORBextractor *ex, const unsigned char *pix, int w, int h, const float *Tcw, const Header &H,
                       bool first)
{
    cv::Mat im(h, w, CV_8UC1, (void *)pix);
    F.mpORBextractorLeft = ex;
    // ref: src/Frame.cc:518-572 -- extract, undistort, (first frame) image bounds and grid constants, grid
    (*ex)(im, cv::Mat(), F.mvKeys, F.mDescriptors);
    F.N = (int)F.mvKeys.size();
    F.mK = cv::Mat::zeros(3, 3, CV_32F);
    F.mK.at<float>(0, 0) = H.fx; F.mK.at<float>(1, 1) = H.fy; F.mK.at<float>(0, 2) = H.cx; F.mK.at<float>(1, 2) = H.cy;
    F.mK.at<float>(2, 2) = 1.0f;
    F.mDistCoef = cv::Mat(4, 1, CV_32F);
    for (int i = 0; i < 4; i++) F.mDistCoef.at<float>(i, 0) = H.D[i];
    F.UndistortKeyPoints();
    if (first) {
        F.ComputeImageBounds(im);
        Frame::mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(Frame::mnMaxX - Frame::mnMinX);
        Frame::mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(Frame::mnMaxY - Frame::mnMinY);
    }
    F.mvuRight = std::vector<float>(F.N, -1);
    F.mvpMapPoints = std::vector<MapPoint *>(F.N, static_cast<MapPoint *>(NULL));
    F.mvbOutlier = std::vector<bool>(F.N, false);
    F.mvScaleFactors = ex->GetScaleFactors();
    F.mTcw = cv::Mat(4, 4, CV_32F);
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) F.mTcw.at<float>(r, c) = Tcw[r * 4 + c];
    F.AssignFeaturesToGrid();
}

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s w h nfeatures frames.raw params.bin out.bin\n", argv[0]); return 2; }
    const int w = atoi(argv[1]), h = atoi(argv[2]), nf = atoi(argv[3]);
    std::vector<unsigned char> raw = slurp(argv[4]), par = slurp(argv[5]);
    Header H;
    memcpy(&H, par.data(), sizeof(H));
    const MpRec *rec = (const MpRec *)(par.data() + sizeof(H));
    const int nrec = (int)((par.size() - sizeof(H)) / sizeof(MpRec));
    FILE *out = fopen(argv[6], "wb");

    Frame::fx = H.fx; Frame::fy = H.fy; Frame::cx = H.cx; Frame::cy = H.cy;

    ORBextractor *ex = new ORBextractor(nf, 1.2f, 8, 20, 7);
    ex->SetPyramidDownload(false);
    Frame mLastFrame, mCurrentFrame;
    buildFrame(mLastFrame, ex, raw.data(), w, h, H.TcwLast, H, true);
    buildFrame(mCurrentFrame, ex, raw.data() + (size_t)w * h, w, h, H.TcwCur, H, false);
    mCurrentFrame.mb = H.mb;
    mCurrentFrame.mbf = H.mbf;

    // image bounds, undistorted keypoints and grid of the current frame, cell by cell
    const float bounds[4] = {Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY};
    fwrite(bounds, 4, 4, out);
    fwrite(&mCurrentFrame.N, 4, 1, out);
    fwrite(mCurrentFrame.mvKeysUn.data(), sizeof(cv::KeyPoint), mCurrentFrame.N, out);
    for (int i = 0; i < FRAME_GRID_COLS; i++)
        for (int j = 0; j < FRAME_GRID_ROWS; j++) {
            int c = (int)mCurrentFrame.mGrid[i][j].size();
            fwrite(&c, 4, 1, out);
            for (int k = 0; k < c; k++) { int v = (int)mCurrentFrame.mGrid[i][j][k]; fwrite(&v, 4, 1, out); }
        }
    // two windows
    const float wx[2] = {310.5f, 20.f}, wy[2] = {200.25f, 470.f}, wr[2] = {45.f, 60.f};
    const int wmin[2] = {1, -1}, wmax[2] = {3, -1};
    for (int t = 0; t < 2; t++) {
        std::vector<size_t> v = mCurrentFrame.GetFeaturesInArea(wx[t], wy[t], wr[t], wmin[t], wmax[t]);
        int c = (int)v.size();
        fwrite(&c, 4, 1, out);
        for (int k = 0; k < c; k++) { int e = (int)v[k]; fwrite(&e, 4, 1, out); }
    }

    {
        // KeyFrame::KeyFrame(Frame&,...) copies the grid; KeyFrame::GetFeaturesInArea (no level filter) -- used by Fuse / Sim3
        KeyFrame kfg;
        kfg.mvKeysUn = mCurrentFrame.mvKeysUn;
        kfg.CopyGridFrom(mCurrentFrame);
        int same = 1;
        for (int i = 0; i < FRAME_GRID_COLS; i++)
            for (int j = 0; j < FRAME_GRID_ROWS; j++) same &= (kfg.mGrid[i][j] == mCurrentFrame.mGrid[i][j]) ? 1 : 0;
        int inimg[2] = {kfg.IsInImage(10.f, 10.f) ? 1 : 0, kfg.IsInImage(-5000.f, 10.f) ? 1 : 0};
        fwrite(&same, 4, 1, out);
        fwrite(inimg, 4, 2, out);
        for (int t = 0; t < 2; t++) {
            std::vector<size_t> v = kfg.GetFeaturesInArea(wx[t], wy[t], wr[t]);
            int c = (int)v.size();
            fwrite(&c, 4, 1, out);
            for (int k = 0; k < c; k++) { int e = (int)v[k]; fwrite(&e, 4, 1, out); }
        }
    }

    // map points seen in the last frame
    const int nmp = nrec < mLastFrame.N ? nrec : mLastFrame.N;
    std::vector<MapPoint> points(nmp);
    for (int i = 0; i < nmp; i++) {
        MapPoint &p = points[i];
        p.mWorldPos = cv::Mat(3, 1, CV_32F);
        p.mWorldPos.at<float>(0, 0) = rec[i].X;
        p.mWorldPos.at<float>(1, 0) = rec[i].Y;
        p.mWorldPos.at<float>(2, 0) = rec[i].Z;
        p.nObs = (int)rec[i].nObs;
        p.mDescriptor = mLastFrame.mDescriptors.row(i).clone();
        p.mTrackProjX = rec[i].projX;
        p.mTrackProjY = rec[i].projY;
        p.mTrackProjXR = rec[i].projXR;
        p.mTrackViewCos = rec[i].viewCos;
        p.mnTrackScaleLevel = (int)rec[i].level;
        p.mbTrackInView = rec[i].inView != 0;
        p.mfMinDistance = rec[i].minDist;
        p.mfMaxDistance = rec[i].maxDist;
        if (rec[i].bad != 0) p.SetBadFlag();
        if (i % 9 != 4) mLastFrame.mvpMapPoints[i] = &p;     // some last-frame features carry no point
        mLastFrame.mvbOutlier[i] = rec[i].outlier != 0;
    }
    {
        // Tracking::TrackWithMotionModel: ORBmatcher matcher(0.9,true); th = 15 mono / 7 stereo
        ORBmatcher matcher(0.9, true);
        int nmatches = matcher.SearchByProjection(mCurrentFrame, mLastFrame, H.thLast, H.bMono != 0);
        fwrite(&nmatches, 4, 1, out);
        for (int i = 0; i < mCurrentFrame.N; i++) {
            int v = mCurrentFrame.mvpMapPoints[i] ? (int)(mCurrentFrame.mvpMapPoints[i] - &points[0]) : -1;
            fwrite(&v, 4, 1, out);
        }
    }
    {
        // Tracking::SearchLocalPoints: ORBmatcher matcher(0.8); matcher.SearchByProjection(mCurrentFrame, vpMapPoints, th)
        // on top of the assignments above (features holding an observed point are closed)
        std::vector<MapPoint *> vpMapPoints;
        for (int i = nmp - 1; i >= 0; i--) vpMapPoints.push_back(&points[i]);
        ORBmatcher matcher(0.8);
        int nmatches = matcher.SearchByProjection(mCurrentFrame, vpMapPoints, H.thLocal);
        fwrite(&nmatches, 4, 1, out);
        for (int i = 0; i < mCurrentFrame.N; i++) {
            int v = mCurrentFrame.mvpMapPoints[i] ? (int)(mCurrentFrame.mvpMapPoints[i] - &points[0]) : -1;
            fwrite(&v, 4, 1, out);
        }
    }
    {
        // Tracking::Relocalization (src/Tracking.cc: matcher2.SearchByProjection(mCurrentFrame, vpCandidateKFs[i], sFound,
        // 10, 100)): the last frame plays the candidate keyframe; every third feature of the current frame is opened
        // again, the points still held are the ones "already found"
        KeyFrame kf;
        kf.mvKeysUn = mLastFrame.mvKeysUn;
        kf.mvpMapPoints = mLastFrame.mvpMapPoints;
        std::set<MapPoint *> sFound;
        for (int i = 0; i < mCurrentFrame.N; i++) {
            if (i % 3 == 0) mCurrentFrame.mvpMapPoints[i] = static_cast<MapPoint *>(NULL);
            if (mCurrentFrame.mvpMapPoints[i]) sFound.insert(mCurrentFrame.mvpMapPoints[i]);
        }
        mCurrentFrame.mnScaleLevels = ex->GetLevels();
        mCurrentFrame.mfScaleFactor = ex->GetScaleFactor();
        mCurrentFrame.mfLogScaleFactor = log(mCurrentFrame.mfScaleFactor);
        ORBmatcher matcher2(0.9, true);
        int nmatches = matcher2.SearchByProjection(mCurrentFrame, &kf, sFound, 10, 100);
        fwrite(&nmatches, 4, 1, out);
        for (int i = 0; i < mCurrentFrame.N; i++) {
            int v = mCurrentFrame.mvpMapPoints[i] ? (int)(mCurrentFrame.mvpMapPoints[i] - &points[0]) : -1;
            fwrite(&v, 4, 1, out);
        }
    }
    {
        // Tracking::MonocularInitialization (src/Tracking.cc:1636-1671): mvbPrevMatched = the initial frame's undistorted
        // keypoints, ORBmatcher matcher(0.9,true), window 100 -- and once more with the updated mvbPrevMatched, as the
        // next frame of a failed initialisation would
        Frame &mInitialFrame = mLastFrame;
        std::vector<cv::Point2f> mvbPrevMatched(mInitialFrame.mvKeysUn.size());
        for (size_t i = 0; i < mInitialFrame.mvKeysUn.size(); i++) mvbPrevMatched[i] = mInitialFrame.mvKeysUn[i].pt;
        std::vector<int> mvIniMatches;
        ORBmatcher matcher(0.9, true);
        for (int round = 0; round < 2; round++) {
            int nmatches = matcher.SearchForInitialization(mInitialFrame, mCurrentFrame, mvbPrevMatched, mvIniMatches, 100);
            fwrite(&nmatches, 4, 1, out);
            fwrite(mvIniMatches.data(), 4, mvIniMatches.size(), out);
            fwrite(mvbPrevMatched.data(), 8, mvbPrevMatched.size(), out);
        }
    }
    fclose(out);
    delete ex;
    return 0;
}
