/*
 * orb_oracle.h -- CPU restatement (ORACLE) of the ORB front end and Hamming matching of
 * hwb0314/VI-ORB-SLAM-ICRA2018.
 *
 * THIS IS TEST INFRASTRUCTURE.  It is the parity checker and the timed CPU baseline
 * ("cpu_baseline.kind = port"); nothing in the product path (vi-orb-slam-icra2018_amd/)
 * includes, links or calls it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * PARITY STATUS: "parity unpinned" at the OpenCV boundary.  The reference has no tests,
 * golden vectors or fixtures, and OpenCV (pinned 2.4.10,
 * CMakeLists.txt:41-42) is not vendored and not installed, so cv::resize / cv::FAST /
 * cv::GaussianBlur / cv::fastAtan2 / cvRound are restated here from their published
 * OpenCV-2.4 algorithms (details in the .c file).  Everything that IS in
 * the reference tree is restated from the cited lines.  The cos/sin used for BRIEF steering
 * is libm's cosf/sinf exactly as the reference calls them (src/ORBextractor.cc:115).
 *
 * Canonicalisation: the quadtree's sort on (size, node pointer)
 * (src/ORBextractor.cc:686) is made deterministic by replacing the pointer with the node's
 * creation sequence number (a later-created node compares greater).
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBO_MAX_LEVELS 16

/* Same field order and size (28 B) as cv::KeyPoint (pt.x, pt.y, size, angle, response,
 * octave, class_id). */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbo_keypoint;

/* A FAST candidate before quadtree distribution: coordinates relative to (16,16) of the level
 * (src/ORBextractor.cc:822-827) and the FAST score. */
typedef struct {
    int32_t x, y, score;
} orbo_cand;

typedef struct {
    int nfeatures, nlevels, iniThFAST, minThFAST;
    double scaleFactor;                         /* include/ORBextractor.h:116 (double member) */
    float mvScaleFactor[ORBO_MAX_LEVELS];
    float mvInvScaleFactor[ORBO_MAX_LEVELS];
    float mvLevelSigma2[ORBO_MAX_LEVELS];
    float mvInvLevelSigma2[ORBO_MAX_LEVELS];
    int mnFeaturesPerLevel[ORBO_MAX_LEVELS];
    int umax[16];
} orbo_params;

/* ---- E0: constructor arithmetic (src/ORBextractor.cc:412-472) ---- */
int orbo_params_init(orbo_params *p, int nfeatures, float scaleFactor, int nlevels,
                     int iniThFAST, int minThFAST);
/* Level size, src/ORBextractor.cc:1132-1133. */
void orbo_level_size(const orbo_params *p, int cols, int rows, int level, int *w, int *h);

/* ---- primitives, each usable on its own by the tests ---- */
int orbo_cvround(double v);                                           /* OpenCV cvRound (SSE2) */
float orbo_fast_atan2(float y, float x);                              /* OpenCV 2.4 fastAtan2 */
void orbo_resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst,
                           int dw, int dh, int dstride);              /* cv::resize INTER_LINEAR */
void orbo_gaussian_blur7_u8(const uint8_t *src, int w, int h, int sstride, uint8_t *dst,
                            int dstride);                             /* GaussianBlur 7x7 s=2 */
/* cv::FAST(img, kps, th, nonmax=true) TYPE_9_16 on a sub-image; returns count, raster order. */
int orbo_fast9_16(const uint8_t *img, int stride, int w, int h, int th, orbo_cand *out, int cap);
/* FAST corner score of one pixel (needs a 3-px margin). */
int orbo_fast_corner_score(const uint8_t *p, int stride, int th);
/* E3 cell loop for one level image; returns number of candidates or <0 on error. */
int orbo_level_candidates(const uint8_t *img, int w, int h, int stride, int iniTh, int minTh,
                          orbo_cand *out, int cap);
/* E4 quadtree; in: candidates (order matters), region size (maxX-minX, maxY-minY), N.
 * out: indices of the retained candidates in result (list) order.  Returns count. */
int orbo_distribute_octtree(const orbo_cand *cands, int ncand, int width, int height, int N,
                            int *out_idx, int cap);
/* E5 */
float orbo_ic_angle(const uint8_t *img, int stride, int x, int y, const int *umax);
/* E7: one descriptor from the blurred level. */
void orbo_brief(const uint8_t *blurred, int stride, int x, int y, float angle_deg,
                uint8_t desc[32]);

/* ---- E1: whole extractor ---- */
typedef struct orbo_extractor orbo_extractor;
orbo_extractor *orbo_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                            int minThFAST);
void orbo_destroy(orbo_extractor *e);
const orbo_params *orbo_get_params(const orbo_extractor *e);
/* Runs operator() (src/ORBextractor.cc:1045-1126).  kps/desc capacity `cap` keypoints.
 * Returns number of keypoints, or <0 on error (-1 bad args, -2 image too small, -3 capacity). */
int orbo_extract(orbo_extractor *e, const uint8_t *img, int w, int h, int stride,
                 orbo_keypoint *kps, uint8_t *desc, int cap);
/* Stage outputs of the last orbo_extract call (valid until the next call). */
const uint8_t *orbo_pyramid_level(const orbo_extractor *e, int level, int *w, int *h, int *stride);
const uint8_t *orbo_blurred_level(const orbo_extractor *e, int level, int *w, int *h, int *stride);
int orbo_level_cands(const orbo_extractor *e, int level, const orbo_cand **c);
int orbo_level_keypoints(const orbo_extractor *e, int level, const orbo_keypoint **k);

/* ---- M0..M3: matching ---- */
int orbo_descriptor_distance(const uint8_t *a, const uint8_t *b);    /* ORBmatcher.cc:1675-1691 */
/* best / second best over a database, strict '<' (lowest index wins ties), initial 256/-1/256. */
void orbo_knn2(const uint8_t *q, int nq, const uint8_t *db, int ndb, int32_t *best_idx,
               int32_t *best_d, int32_t *second_d);
/* M3: same over explicit candidate lists (CSR): for query i, candidates cand[off[i]..off[i+1]). */
void orbo_knn2_lists(const uint8_t *q, int nq, const uint8_t *db, const int32_t *off,
                     const int32_t *cand, int32_t *best_idx, int32_t *best_d, int32_t *second_d);

/* M1/M2: SearchByBoW core on node-grouped features (ORBmatcher.cc:159-288, 522-655).
 * Side 1 = key frame, side 2 = frame (M1) or second key frame (M2).
 * groups: sorted node ids + CSR offsets into feature-index arrays (DBoW2::FeatureVector).
 * valid1[i] != 0  <=> feature i of side 1 has a good MapPoint; valid2 may be NULL (M1: no test).
 * th_mode 0: accept best <= th (M1, :228)   1: accept best < th (M2, :598).
 * Outputs: match12[n1] (index into side 2 or -1) and match21[n2] (index into side 1 or -1),
 * after the rotation-consistency filter when check_ori != 0.  Returns nmatches. */
int orbo_search_by_bow(const uint8_t *desc1, int n1, const uint8_t *valid1, const float *angle1,
                       const int32_t *node1, const int32_t *off1, const int32_t *idx1, int ng1,
                       const uint8_t *desc2, int n2, const uint8_t *valid2, const float *angle2,
                       const int32_t *node2, const int32_t *off2, const int32_t *idx2, int ng2,
                       int th, int th_mode, float nnratio, int check_ori, int32_t *match12,
                       int32_t *match21);
/* ORBmatcher::ComputeThreeMaxima (ORBmatcher.cc:1629-1670) on bin sizes. */
void orbo_three_maxima(const int *histo_sizes, int L, int *ind1, int *ind2, int *ind3);

/* ---- next row: ORB vocabulary tree ----
 * Binary format of Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1680-1751 (loadFromBinaryFile /
 * saveToBinaryFile): u32 nb_nodes, u32 size_node(=41), i32 k, i32 L, i32 scoring, i32 weighting, then per
 * node id 1..nb_nodes-1: i32 parent, u8 desc[32], f32 weight, u8 is_leaf.  Children keep file order;
 * words are numbered in the order their leaves appear. */
typedef struct orbo_vocab orbo_vocab;
orbo_vocab *orbo_vocab_load(const void *blob, size_t nbytes);
void orbo_vocab_free(orbo_vocab *v);
int orbo_vocab_info(const orbo_vocab *v, int *k, int *L, int *scoring, int *weighting, int *nnodes, int *nwords);
/* Per-feature transform, TemplatedVocabulary.h:1443-1485: descend from the root, at every level take
 * the child with the smallest Hamming distance (first wins ties); word id and weight of the leaf;
 * node id at level L - levelsup (0 when that level is <= 0 or is never reached). */
void orbo_vocab_transform(const orbo_vocab *v, const uint8_t *desc, int n, int levelsup, int32_t *word_id,
                          float *weight, int32_t *node_id);
/* BowVector of a feature set (TemplatedVocabulary.h:1167-1258, BowVector.cpp:34-86), features visited in
 * ascending index order (canonical): out_word/out_value sorted by word id; returns
 * the number of entries (<= n). */
int orbo_vocab_bow(const orbo_vocab *v, const int32_t *word_id, const float *weight, int n, int32_t *out_word,
                   double *out_value);

/* ---- next row: Frame::ComputeStereoMatches (src/Frame.cc:810-984) ----
 * pyrL / pyrR: the extractor pyramids (continuous, stride = width), lw / lh their sizes.  Outputs
 * mvuRight / mvDepth [nL] (-1 = no match).  Returns the number of matches before the outlier cut. */
int orbo_stereo_matches(const orbo_keypoint *kL, const uint8_t *dL, int nL, const orbo_keypoint *kR,
                        const uint8_t *dR, int nR, const uint8_t *const *pyrL, const uint8_t *const *pyrR,
                        const int *lw, const int *lh, const float *mvScaleFactors, const float *mvInvScaleFactors,
                        float mb, float mbf, float *mvuRight, float *mvDepth);


/* ---- frame grid + guided search ---- */
#define ORBO_Q_ACTIVE 1    /* the query takes part (mbTrackInView && !isBad / LastFrame point valid and in bounds) */
#define ORBO_Q_OBSERVED 2  /* its MapPoint has Observations() > 0: a matched feature is closed to later queries */
typedef struct {
    float u, v, radius, proj_xr;
    int32_t min_level, max_level;
    float angle;
    int32_t flags;
} orbo_proj_query;
void orbo_grid_build(const orbo_keypoint *kps, int n, float minX, float minY, float invW, float invH,
                     int32_t *cell_off /* 64*48+1 */, int32_t *cell_idx /* n */);
int orbo_features_in_area(const orbo_keypoint *kps, const int32_t *cell_off, const int32_t *cell_idx, float minX,
                          float minY, float invW, float invH, float x, float y, float r, int minLevel, int maxLevel,
                          int32_t *out, int cap);
int orbo_search_by_projection(const orbo_keypoint *kps, const uint8_t *desc, int n, const float *u_right,
                              const uint8_t *occupied_in, float minX, float minY, float invW, float invH,
                              const orbo_proj_query *q, const uint8_t *qdesc, int nq, int use_ratio, float nnratio,
                              int check_ori, int th_high, int32_t *match);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:283-349), P points in CSR form; best[p] = row of least median
 * distance to the others (first wins), -1 for an empty list */
void orbo_distinctive_descriptors(const uint8_t *desc, const int32_t *off, int P, int32_t *best, int32_t *best_median);

/* inner loop of ORBmatcher::Fuse (src/ORBmatcher.cc:887-950; :1044-1075) and SearchBySim3 (:1190-1224): the first
 * feature of smallest distance in the window on levels [min_level, max_level]; inv_level_sigma2 != NULL adds Fuse's
 * chi-square gate (7.8 with a right coordinate >= 0, else 5.99).  -1 / 256 when none. */
void orbo_window_best(const orbo_keypoint *kps, const uint8_t *desc, int n, const float *u_right,
                      const float *inv_level_sigma2, float minX, float minY, float invW, float invH,
                      const orbo_proj_query *q, const uint8_t *qdesc, int nq, int32_t *best_idx, int32_t *best_dist);

/* ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:405-520); prev_matched n1 x 2 floats, in/out */
int orbo_search_for_initialization(const orbo_keypoint *kps1, const uint8_t *desc1, int n1, const orbo_keypoint *kps2,
                                   const uint8_t *desc2, int n2, float minX, float minY, float invW, float invH,
                                   float *prev_matched, int window_size, float nnratio, int check_ori, int th_low,
                                   int32_t *matches12);

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:657-827) with CheckDistEpipolarLine (:140-157); F12 row-major */
int orbo_search_for_triangulation(const orbo_keypoint *kps1, const uint8_t *desc1, int n1, const uint8_t *skip1,
                                  const float *u_right1, const int32_t *node1, const int32_t *off1, const int32_t *idx1,
                                  int ng1, const orbo_keypoint *kps2, const uint8_t *desc2, int n2, const uint8_t *skip2,
                                  const float *u_right2, const int32_t *node2, const int32_t *off2, const int32_t *idx2,
                                  int ng2, const float *F12, float ex, float ey, const float *scale_factors2,
                                  const float *level_sigma2_2, int only_stereo, int check_ori, int th_low,
                                  int32_t *matches12);

/* ---- undistortion / rectification ---- */
/* cv::undistortPoints(src, dst, K, D, Mat(), P) as Frame::UndistortKeyPoints calls it (src/Frame.cc:767);
 * K, P: 3x3 row-major float (P may be NULL = identity), D: nD in {0,4,5,8} coefficients. */
void orbo_undistort_points(const float *xy_in, int n, const float *K, const float *D, int nD, const float *P,
                           float *xy_out);
/* cv::initUndistortRectifyMap(K, D, R, P(0:3,0:3), size, CV_32F, map1, map2) (stereo_euroc.cc:96-98) */
void orbo_init_undistort_rectify_map(const double *K, const double *D, int nD, const double *R, const double *P,
                                     int w, int h, float *mapx, float *mapy);
/* cv::remap(src, dst, map1, map2, INTER_LINEAR) (stereo_euroc.cc:136-137), split as OpenCV does it */
void orbo_remap_prepare(const float *mapx, const float *mapy, int w, int h, int16_t *xy, uint16_t *frac);
void orbo_remap_linear_u8(const uint8_t *src, int sw, int sh, int sstride, const int16_t *xy, const uint16_t *frac,
                          int dw, int dh, uint8_t *dst, int dstride);

#ifdef __cplusplus
}
#endif
#endif
