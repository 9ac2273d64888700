/*
 * orbhip.h -- C ABI of liborbhip.so: the MI355X (gfx950) ORB front end and Hamming matcher.
 *
 * This is the drop-in boundary for ONE hot path of hwb0314/VI-ORB-SLAM-ICRA2018:
 * ORBextractor::operator() and ORBmatcher's descriptor matching.  The C++ classes
 * ORB_SLAM2::ORBextractor / ORB_SLAM2::ORBmatcher in include/orbhip/ keep the reference's
 * signatures and forward to these entry points (INTEGRATION.md shows the binding).
 *
 * Plain pointers and sizes only; no C++/torch types.  Unless a name ends in _device every
 * pointer is a HOST pointer and the call is synchronous.  *_device entry points take device
 * pointers, enqueue on the context's HIP stream and return without synchronising; call
 * orbhip_sync() before reading results.
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference
 * repository root).
 *
 * Errors: 0 = ok, negative = error (see ORBHIP_E_*); orbhip_last_error() gives the text.
 * There is no CPU fallback: without a usable HIP device orbhip_create() fails.
 * Threading: one context = one extractor instance = one HIP stream; a context is not
 * re-entrant (same as the reference class, which mutates mvImagePyramid), different contexts
 * may be used concurrently from different host threads (src/Frame.cc:422-425 does this for
 * stereo).
 */
#ifndef ORBHIP_H
#define ORBHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBHIP_OK 0
#define ORBHIP_E_ARG (-1)       /* bad argument */
#define ORBHIP_E_SIZE (-2)      /* image larger than the context or too small for the cell grid */
#define ORBHIP_E_CAPACITY (-3)  /* output capacity too small */
#define ORBHIP_E_HIP (-4)       /* HIP runtime error */
#define ORBHIP_E_NODEVICE (-5)  /* no HIP device / wrong architecture */
#define ORBHIP_E_COMM (-6)      /* RCCL error */

#define ORBHIP_MAX_LEVELS 16

typedef struct orbhip_ctx orbhip_ctx;

/* Binary layout of cv::KeyPoint (28 bytes): pt.x, pt.y, size, angle, response, octave,
 * class_id.  Replaces std::vector<cv::KeyPoint>& of include/ORBextractor.h:77-79. */
typedef struct orbhip_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orbhip_keypoint;

/* A FAST candidate before quadtree distribution (debug/parity access): coordinates relative
 * to (16,16) of the level as in src/ORBextractor.cc:822-827, and the FAST score. */
typedef struct orbhip_cand {
    int32_t x, y, score;
} orbhip_cand;

/* Number of visible HIP devices (0 if none). */
int orbhip_device_count(void);

/* Replaces ORBextractor::ORBextractor(int nfeatures, float scaleFactor, int nlevels,
 * int iniThFAST, int minThFAST) (include/ORBextractor.h:69-70, src/ORBextractor.cc:412-472).
 * max_w/max_h bound the image size, max_batch the frames per batched call; all device
 * buffers are allocated here, none in the per-frame calls.  Returns NULL on failure
 * (orbhip_last_error(NULL) has the reason).  Limits (ORBHIP_E_SIZE): every pyramid level must hold at least one
 * 30-pixel cell and one quadtree root (the reference divides by zero there); levels up to 4128 x 4128; scaleFactor > 1.
 * There is no bound on a level's feature quota (node tables beyond the LDS move to a device scratch block). */
orbhip_ctx *orbhip_create(int device, int nfeatures, float scaleFactor, int nlevels,
                          int iniThFAST, int minThFAST, int max_w, int max_h, int max_batch);
void orbhip_destroy(orbhip_ctx *ctx);
const char *orbhip_last_error(const orbhip_ctx *ctx);
int orbhip_sync(orbhip_ctx *ctx);
/* The context's hipStream_t (as void*), so a caller can order its own work against it. */
void *orbhip_stream(orbhip_ctx *ctx);

/* Replaces GetLevels/GetScaleFactor(s)/GetInverseScaleFactors/GetScaleSigmaSquares/
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:81-101) and exposes
 * mnFeaturesPerLevel/umax (:122,:124).  Any output pointer may be NULL.  Arrays need nlevels
 * entries (umax: 16). */
int orbhip_get_tables(const orbhip_ctx *ctx, int *nlevels, double *scaleFactor, float *mvScaleFactor,
                      float *mvInvScaleFactor, float *mvLevelSigma2, float *mvInvLevelSigma2,
                      int *mnFeaturesPerLevel, int *umax);
/* The same tables without a context or a device (pure host arithmetic of
 * src/ORBextractor.cc:417-471), for constructing the C++ class before the first image arrives.
 * Arrays need nlevels entries (umax: 16); any output pointer may be NULL. */
int orbhip_tables(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST,
                  float *mvScaleFactor, float *mvInvScaleFactor, float *mvLevelSigma2,
                  float *mvInvLevelSigma2, int *mnFeaturesPerLevel, int *umax);
/* Upper bound on keypoints per frame (nfeatures plus the quadtree's overshoot); use as `cap`. */
int orbhip_max_keypoints(const orbhip_ctx *ctx);
/* Size of pyramid level `level` for a w x h input (src/ORBextractor.cc:1132-1133). */
int orbhip_level_size(const orbhip_ctx *ctx, int w, int h, int level, int *lw, int *lh);

/* Replaces ORBextractor::operator()(InputArray image, InputArray mask, vector<KeyPoint>&,
 * OutputArray descriptors) (include/ORBextractor.h:77-79, src/ORBextractor.cc:1045-1126).
 * img: 8-bit single channel, `stride` bytes per row.  kps/desc: capacity `cap` keypoints
 * (desc is cap x 32 bytes, row i = descriptor of keypoint i).  timings_ms (may be NULL)
 * receives {pyramid, keypoints, descriptors} in ms = GetTimeOfComputePyramid /
 * GetTimeOfComputeKeyPointsOctTree / GetTImeOfComputeDescriptor (include/ORBextractor.h:51-53). */
int orbhip_extract(orbhip_ctx *ctx, const uint8_t *img, int w, int h, int stride,
                   orbhip_keypoint *kps, uint8_t *desc, int cap, int *n_out, float timings_ms[3]);

/* Batched mode (new; the reference processes one image per call): B independent frames of the
 * same size per call, one launch per stage.  imgs[b] are host pointers; outputs are
 * kps[b*cap + i], desc[(b*cap + i)*32], n_out[b]. */
int orbhip_extract_batch(orbhip_ctx *ctx, const uint8_t *const *imgs, int B, int w, int h,
                         int stride, orbhip_keypoint *kps, uint8_t *desc, int cap, int *n_out);

/* Same with everything resident in device memory: d_imgs = B frames, frame b at
 * d_imgs + b*frame_stride, rows `stride` bytes apart (stride % 4 == 0 and 4-byte aligned
 * base required).  d_kps [B*cap] orbhip_keypoint, d_desc [B*cap*32] bytes, d_counts [B]
 * int32.  Asynchronous on the context stream.  Level 0 of the pyramid aliases d_imgs until the
 * next extract call on this context. */
int orbhip_extract_batch_device(orbhip_ctx *ctx, const void *d_imgs, int B, int w, int h,
                                int stride, size_t frame_stride, void *d_kps, void *d_desc, int cap,
                                void *d_counts);

/* ---- host-fed pipeline (new; the reference's frames always arrive from host memory: cv::imread at
 * Examples/Monocular/mono_euroc.cc:73, then ORBextractor::operator() via src/Frame.cc:591-597) ----
 * A ring of `depth` (2..8) slots per context: batch n + 1 is copied to the device and batch n - 1's keypoints /
 * descriptors are copied back while batch n computes (one copy-in stream, the context's compute stream, one copy-out
 * stream; the results land in pinned host memory owned by the context).  Frames of w x h, up to B per batch.
 *   orbhip_pipe_submit  enqueues one batch and returns at once.  `frames`: B images, image b at frames + b * frame_stride,
 *                       rows `stride` bytes apart.  The copy is one asynchronous DMA when the memory is pinned
 *                       (orbhip_host_alloc, or the caller's own hipHostMalloc / hipHostRegister); pageable memory works
 *                       but is staged by the driver.  The caller must leave the frames alone until the batch's wait returns.
 *                       ORBHIP_E_CAPACITY when `depth` batches are in flight and none was collected.
 *   orbhip_pipe_wait    blocks until the OLDEST outstanding batch is complete and returns pointers into that slot's pinned
 *                       result block: kps[b * cap + i], desc[(b * cap + i) * 32], n_out[b].  They stay valid until the
 *                       NEXT orbhip_pipe_wait (or orbhip_pipe_destroy) on this context, however many batches are submitted
 *                       in between: the ring has depth + 1 host result blocks, and no submit it admits reuses the block
 *                       the last wait returned.
 * Results are those of orbhip_extract_batch for the same frames (same kernels, same order). */
void *orbhip_host_alloc(size_t nbytes);
void orbhip_host_free(void *p);
int orbhip_pipe_create(orbhip_ctx *ctx, int depth, int B, int w, int h);
int orbhip_pipe_destroy(orbhip_ctx *ctx);
int orbhip_pipe_submit(orbhip_ctx *ctx, const uint8_t *frames, int B, int stride, size_t frame_stride);
int orbhip_pipe_wait(orbhip_ctx *ctx, const orbhip_keypoint **kps, const uint8_t **desc, const int32_t **n_out, int *B,
                     int *cap);
/* Optional matching stage of the pipeline: after the extraction of a batch, Frame::ComputeBoW (src/Frame.cc:739-746,
 * orbhip_vocab_transform_device with `levelsup`) and ORBmatcher(nnratio, check_ori)::SearchByBoW of every frame b >= 1
 * against frame b - 1 of the same batch (Tracking::TrackReferenceKeyFrame, src/Tracking.cc:1881-1885;
 * orbhip_search_by_bow_seq_device with lag 1, th_mode 0) run on the device-resident outputs and their results travel back
 * with the keypoints.  Needs a vocabulary in the context.  orbhip_pipe_matches returns, for the batch the last
 * orbhip_pipe_wait returned, match12[b * cap + i1] (feature of frame b matched by feature i1 of frame b - 1, or -1),
 * match21[b * cap + i2] and nmatches[b] (frame 0 of a batch: all -1 / 0). */
int orbhip_pipe_enable_bow(orbhip_ctx *ctx, int levelsup, float nnratio, int check_ori);
int orbhip_pipe_matches(orbhip_ctx *ctx, const int32_t **match12, const int32_t **match21, const int32_t **nmatches);

/* Replaces reads of the public member std::vector<cv::Mat> mvImagePyramid
 * (include/ORBextractor.h:103; read by src/Frame.cc:817,907,919,924).  Copies level `level`
 * of frame `frame` of the last extract call to dst (rows dst_stride apart). */
int orbhip_get_pyramid_level(orbhip_ctx *ctx, int frame, int level, uint8_t *dst, int dst_stride,
                             int *w, int *h);

/* The same member without a copy per level: with orbhip_set_host_pyramid(ctx, 1) every following orbhip_extract /
 * orbhip_extract_batch also lands levels 1.. of its frames in page-locked host memory (one device-to-host copy that runs
 * beside the kernels), and orbhip_host_pyramid_level returns a pointer into that block (rows *stride apart); level 0 is
 * the page-locked copy of the caller's frame that the single-frame path uploads from.  The pointers stay valid until the
 * next extract call on this context or orbhip_destroy -- the drop-in's mvImagePyramid[level] are headers on them, read
 * right after the extraction as src/Frame.cc:817 does.  Returns ORBHIP_E_ARG when a level is not staged (host pyramid
 * off, device-pointer entry points, or level 0 of a batch of 8 or more frames: use the caller's image). */
int orbhip_set_host_pyramid(orbhip_ctx *ctx, int on);
int orbhip_host_pyramid_level(orbhip_ctx *ctx, int frame, int level, const uint8_t **ptr, int *stride, int *w, int *h);

/* ---- parity/debug access to stage outputs of the last extract call ---- */
/* blurred level (cv::GaussianBlur at src/ORBextractor.cc:1103-1104) */
int orbhip_debug_get_blurred_level(orbhip_ctx *ctx, int frame, int level, uint8_t *dst,
                                   int dst_stride, int *w, int *h);
/* FAST candidates of one level in the reference's order (cells row-major, raster inside a
 * cell; src/ORBextractor.cc:791-831). */
int orbhip_debug_get_candidates(orbhip_ctx *ctx, int frame, int level, orbhip_cand *out, int cap,
                                int *n_out);
/* keypoints of one level after DistributeOctTree + orientation, level coordinates
 * (src/ORBextractor.cc:833-854). */
int orbhip_debug_get_level_keypoints(orbhip_ctx *ctx, int frame, int level, orbhip_keypoint *out,
                                     int cap, int *n_out);

/* ---- matching ---- */
/* Replaces ORBmatcher::DescriptorDistance (include/ORBmatcher.h:47, src/ORBmatcher.cc:1675-1691)
 * applied to a whole query set against a whole database with the best / second-best
 * bookkeeping of every search routine (src/ORBmatcher.cc:205-226 etc.): for each query the
 * lowest-index minimum (strict '<'), its distance and the second smallest distance; initial
 * values 256 / -1 / 256.  Descriptors are rows of 32 bytes.  Device pointers of any alignment are accepted; the matrix-pipe
 * kernels want d_q 16-byte and d_db 4-byte aligned (what hipMalloc and rows of 32 bytes give) and hand other pointers to
 * the scalar kernels (same results, about a third of the rate). */
int orbhip_hamming_knn2(orbhip_ctx *ctx, const uint8_t *q, int nq, const uint8_t *db, int ndb,
                        int32_t *best_idx, int32_t *best_d, int32_t *second_d);
int orbhip_hamming_knn2_device(orbhip_ctx *ctx, const void *d_q, int nq, const void *d_db, int ndb,
                               void *d_best_idx, void *d_best_d, void *d_second_d);

/* Batched form for a sequence held on the device (new; the reference matches one frame pair per
 * call): descriptor sets laid out as the outputs of orbhip_extract_batch_device (set b at
 * d_desc + b*cap*32 with d_counts[b] rows).  For b >= lag the queries are set b and the database
 * is set b-lag; outputs at [b*cap + i]; for b < lag the outputs are -1 / 256 / 256.  One launch for
 * all B sets. */
int orbhip_hamming_knn2_seq_device(orbhip_ctx *ctx, const void *d_desc, const void *d_counts, int cap,
                                   int B, int lag, void *d_best_idx, void *d_best_d, void *d_second_d);

/* Same bookkeeping over explicit candidate lists in CSR form (query i examines
 * cand[off[i] .. off[i+1])), the shape of SearchByProjection / SearchForInitialization /
 * Fuse / SearchBySim3 inner loops (src/ORBmatcher.cc:76-125, 432-461, 901-949, 1199-1224). */
int orbhip_hamming_knn2_lists(orbhip_ctx *ctx, const uint8_t *q, int nq, const uint8_t *db, int ndb,
                              const int32_t *off, const int32_t *cand, int32_t *best_idx,
                              int32_t *best_d, int32_t *second_d);

/* Replaces the matching core of ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...)
 * (src/ORBmatcher.cc:159-288; th_mode 0: accept best <= th) and
 * SearchByBoW(KeyFrame*, KeyFrame*, ...) (:522-655; th_mode 1: accept best < th, valid2 given).
 * Side 1/2 FeatureVectors in CSR form: sorted node ids, offsets [ng+1], feature indices.
 * valid1[i] != 0 <=> feature i has a good MapPoint; valid2 may be NULL.
 * match12[n1] / match21[n2] receive the matched index on the other side or -1, after the
 * rotation histogram filter (ComputeThreeMaxima, :1629-1670) when check_ori != 0.
 * *nmatches receives the return value of the reference routine. */
int orbhip_search_by_bow(orbhip_ctx *ctx, const uint8_t *desc1, int n1, const uint8_t *valid1,
                         const float *angle1, const int32_t *node1, const int32_t *off1,
                         const int32_t *idx1, int ng1, const uint8_t *desc2, int n2,
                         const uint8_t *valid2, const float *angle2, const int32_t *node2,
                         const int32_t *off2, const int32_t *idx2, int ng2, int th, int th_mode,
                         float nnratio, int check_ori, int32_t *match12, int32_t *match21,
                         int *nmatches);

/* ---- ORB vocabulary ----
 * Replaces ORBVocabulary::loadFromBinaryFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1680-1721,
 * called from src/System.cc:336-339): parses the binary vocabulary (header u32 nb_nodes, u32 size_node=41,
 * i32 k, L, scoring, weighting; then per node i32 parent, u8 desc[32], f32 weight, u8 is_leaf) and keeps
 * it as device tables in the context.  _device: the blob is already in device memory (e.g. after
 * orbhip_bcast_blob_device); it is copied back once for parsing. */
int orbhip_vocab_load(orbhip_ctx *ctx, const void *blob, size_t nbytes);
int orbhip_vocab_load_device(orbhip_ctx *ctx, const void *d_blob, size_t nbytes);
/* The vocabulary tables of `src` serve `dst` too (contexts of one device; borrowed, not copied -- 58 MB for the stock tree).
 * What lets the extractor's context run the transform inside orbhip_frame_build on the vocabulary the ORBVocabulary drop-in
 * loaded into its own context.  The device block is reference-counted: src may load another vocabulary or be destroyed while
 * dst still uses what it borrowed -- dst then keeps running on the OLD tables until it shares again.
 * orbhip_vocab_generation: a process-wide counter of orbhip_vocab_load calls as seen by this context's tables (0 = no
 * vocabulary); a borrower compares it with the lender's to learn that it should share again (no counterpart in the reference:
 * System.cc:336-339 loads the vocabulary once).  orbhip_vocab_share / orbhip_vocab_generation on `src` may run on another
 * thread than an orbhip_vocab_load on it: the (block, tables) pair is swapped under a lock, the borrower sees the old pair or
 * the new one.  Work already queued on src's OWN stream against the old tables is the caller's to order, as with any two
 * calls on one context. */
int orbhip_vocab_share(orbhip_ctx *dst, const orbhip_ctx *src);
unsigned long long orbhip_vocab_generation(const orbhip_ctx *ctx);
/* Replaces ORBVocabulary::loadFromTextFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1564-1647; chosen by
 * src/System.cc:335-336 for a ".txt" vocabulary such as the stock ORBvoc.txt): host-only conversion of the text (first
 * line "k L scoring weighting", then per node "parent is_leaf d0 .. d31 weight") to the binary layout above, which
 * orbhip_vocab_load takes.  Call with blob = NULL to get *blob_bytes = 24 + 41 * nodes, then with a buffer.  node_weight
 * (may be NULL; (*blob_bytes - 24) / 41 entries, node ids 1..n) receives the weights as the doubles the reference's text
 * loader keeps (Node::weight is a double; the binary format narrows it to float, so BowVector values of a text-loaded
 * vocabulary come from this array).  Lines without a token are skipped (the reference turns the empty line after the final
 * newline into a node built from uninitialised memory).  ORBHIP_E_ARG for a malformed text, ORBHIP_E_CAPACITY for a
 * buffer that is too small. */
int orbhip_vocab_text_to_binary(const char *text, size_t nbytes, void *blob, size_t blob_cap, size_t *blob_bytes,
                                double *node_weight, size_t weight_cap);
int orbhip_vocab_info(const orbhip_ctx *ctx, int *k, int *L, int *scoring, int *weighting, int *nnodes,
                      int *nwords);
/* Replaces the per-feature ORBVocabulary::transform (TemplatedVocabulary.h:1443-1485) as used by
 * Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:739-746, src/KeyFrame.cc:392-400): word id and
 * weight of the leaf reached, node id at level L - levelsup (0 if that level is <= 0).  The caller builds
 * BowVector (sum of weights per word, normalised) and FeatureVector (indices per node, weight > 0 only)
 * from these arrays. */
int orbhip_vocab_transform(orbhip_ctx *ctx, const uint8_t *desc, int n, int levelsup, int32_t *word_id,
                           float *weight, int32_t *node_id);
int orbhip_vocab_transform_device(orbhip_ctx *ctx, const void *d_desc, int n, int levelsup, void *d_word_id,
                                  void *d_weight, void *d_node_id);
/* Batched SearchByBoW (src/ORBmatcher.cc:159-288 / :522-655 with th_mode 0 / 1) for a sequence on the
 * device, laid out like the outputs of orbhip_extract_batch_device; d_node / d_weight [B*cap] from
 * orbhip_vocab_transform_device; d_valid [B*cap] bytes ("has a good MapPoint") or NULL = all valid.
 * For b >= lag side 1 is set b-lag (the key frame) and side 2 is set b; d_match12[b*cap + i1] = matched
 * side-2 feature or -1, d_match21[b*cap + i2] = matched side-1 feature or -1, d_nmatches[b] = the
 * reference routine's return value.  TH_LOW = 50.  One launch for all B pairs.  cap <= 4096 (ORBHIP_E_SIZE beyond: the
 * per-pair tables live in LDS). */
int orbhip_search_by_bow_seq_device(orbhip_ctx *ctx, const void *d_desc, const void *d_kps, const void *d_counts,
                                    const void *d_node, const void *d_weight, const void *d_valid, int cap,
                                    int B, int lag, int th_mode, float nnratio, int check_ori, void *d_match12,
                                    void *d_match21, void *d_nmatches);

/* ---- stereo ----
 * Replaces Frame::ComputeStereoMatches (src/Frame.cc:810-984): for every left keypoint the best right
 * keypoint in its row band (octave +-1, u in [uL - mbf/mb, uL], distance < 75), refined by the 11-shift SAD
 * on the pyramid level of the keypoint and a parabola fit; outputs mvuRight / mvDepth (-1 = none) after the
 * median-based outlier cut.  `left` and `right` are the two extractor contexts that have just extracted the
 * left / right image(s) of the same size: their device-resident pyramids are read in place (the reference
 * reads mpORBextractorLeft/Right->mvImagePyramid, :817,:907,:919,:924).  *nmatch = matches before the cut. */
int orbhip_stereo_match(orbhip_ctx *left, orbhip_ctx *right, const orbhip_keypoint *kpsL, const uint8_t *descL,
                        int nL, const orbhip_keypoint *kpsR, const uint8_t *descR, int nR, float mb, float mbf,
                        float *mvuRight, float *mvDepth, int *nmatch);
/* Batched, device-resident form (B stereo pairs; arrays laid out like orbhip_extract_batch_device outputs). */
int orbhip_stereo_match_device(orbhip_ctx *left, orbhip_ctx *right, const void *d_kpsL, const void *d_descL,
                               const void *d_cntL, const void *d_kpsR, const void *d_descR, const void *d_cntR,
                               int cap, int B, float mb, float mbf, void *d_uRight, void *d_depth, void *d_nmatch);

/* ---- frame grid and guided search ----
 * The 64 x 48 grid of include/Frame.h:41-42 as CSR: cell id = ix * 48 + iy (the order
 * Frame::GetFeaturesInArea visits cells in), feature indices ascending inside a cell (the push_back
 * order of Frame::AssignFeaturesToGrid, src/Frame.cc:574-589; cell of a feature by PosInGrid,
 * :726-736).  min_x, min_y, inv_w, inv_h are Frame::mnMinX, mnMinY, mfGridElementWidthInv,
 * mfGridElementHeightInv (:556-557).  d_cell_off [B][ORBHIP_GRID_CELLS + 1], d_cell_idx [B][cap]. */
#define ORBHIP_GRID_COLS 64
#define ORBHIP_GRID_ROWS 48
#define ORBHIP_GRID_CELLS (ORBHIP_GRID_COLS * ORBHIP_GRID_ROWS)
int orbhip_grid_build_device(orbhip_ctx *ctx, const void *d_kps_un, const void *d_counts, int cap, int B, float min_x,
                             float min_y, float inv_w, float inv_h, void *d_cell_off, void *d_cell_idx);
int orbhip_grid_build(orbhip_ctx *ctx, const orbhip_keypoint *kps_un, int n, float min_x, float min_y, float inv_w,
                      float inv_h, int32_t *cell_off, int32_t *cell_idx);

/* One projected point of a guided search: where it falls in the frame (u, v), the window half-size, the
 * octave range GetFeaturesInArea filters on (min_level <= 0 and max_level < 0: no filter, :693-707), the
 * projected right coordinate (compared with mvuRight when that is > 0) and the orientation of the source
 * keypoint.  flags: ORBHIP_Q_ACTIVE = the point takes part (mbTrackInView && !isBad(), :55-59; or the
 * LastFrame point is valid, not an outlier and projects inside the image, :1371-1398);
 * ORBHIP_Q_OBSERVED = its MapPoint has Observations() > 0, so a feature it takes is closed to later points. */
#define ORBHIP_Q_ACTIVE 1
#define ORBHIP_Q_OBSERVED 2
typedef struct {
    float u, v, radius, proj_xr;
    int32_t min_level, max_level;
    float angle;
    int32_t flags;
} orbhip_proj_query;

/* Replaces Frame::GetFeaturesInArea (src/Frame.cc:671-724) for nq windows at once (only u, v, radius,
 * min_level, max_level of a query are read): out_off[nq + 1], out_idx[out_off[nq]] in the reference's
 * order.  Returns ORBHIP_E_ARG if out_cap is too small (out_off is still filled). */
int orbhip_features_in_area(orbhip_ctx *ctx, const orbhip_keypoint *kps_un, int n, float min_x, float min_y,
                            float inv_w, float inv_h, const orbhip_proj_query *queries, int nq, int32_t *out_off,
                            int32_t *out_idx, int out_cap);

/* Replaces the search loops of ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)
 * (src/ORBmatcher.cc:45-129; use_ratio = 1: second best and the ratio test when best and second lie on the
 * same octave) and SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) (:1341-1498;
 * use_ratio = 0: best only, rotation histogram when check_ori).  Queries are processed in index order, as
 * the reference processes its points: occupied[i] != 0 marks frame features that already hold a MapPoint
 * with observations (:88-90, :1413-1415).  match[i] = index of the query assigned to frame feature i (the
 * last one, as in the reference), -1 if none was, -2 if one was and the rotation check removed it (the
 * reference stores NULL there, :1489); *nmatches = the reference routine's return value.  The caller
 * projects the points (pose arithmetic stays on the host) and fills the queries.  TH_HIGH = th_high. */
int orbhip_search_by_projection(orbhip_ctx *ctx, const orbhip_keypoint *kps_un, const uint8_t *desc, int n,
                                const float *u_right, const uint8_t *occupied, float min_x, float min_y,
                                float inv_w, float inv_h, const orbhip_proj_query *queries, const uint8_t *qdesc,
                                int nq, int use_ratio, float nnratio, int check_ori, int th_high, int32_t *match,
                                int *nmatches);
/* Batched, device-resident form: B frames laid out like orbhip_extract_batch_device outputs, the grid from
 * orbhip_grid_build_device, d_queries [B][cap_q], d_qdesc [B][cap_q][32], d_nq [B]; d_u_right / d_occupied
 * [B][cap] or NULL; d_match [B][cap], d_nmatches [B]. */
int orbhip_search_by_projection_device(orbhip_ctx *ctx, const void *d_kps_un, const void *d_desc,
                                       const void *d_counts, int cap, int B, const void *d_u_right,
                                       const void *d_occupied, float min_x, float min_y, float inv_w, float inv_h,
                                       const void *d_cell_off, const void *d_cell_idx, const void *d_queries,
                                       const void *d_qdesc, const void *d_nq, int cap_q, int use_ratio,
                                       float nnratio, int check_ori, int th_high, void *d_match, void *d_nmatches);

/* Replaces the body of MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:283-349; the remaining caller of
 * ORBmatcher::DescriptorDistance, run by LocalMapping on every point of a new key frame and after every fusion) for P map
 * points at once: point p's observed descriptors (rows of the key frames that see it, in the order of its observation map)
 * are rows off[p] .. off[p + 1] of desc; off[0] = 0.  best[p] = the row within that list with the least median distance
 * to the list (the reference's sorted_row[0.5 * (N - 1)], self-distance included; the first such row), -1 for an empty
 * list; best_median[p] (may be NULL in the host form) = that median, INT_MAX for an empty list. */
int orbhip_distinctive_descriptors(orbhip_ctx *ctx, const uint8_t *desc, const int32_t *off, int P, int32_t *best,
                                   int32_t *best_median);
int orbhip_distinctive_descriptors_device(orbhip_ctx *ctx, const void *d_desc, const void *d_off, int P, void *d_best,
                                          void *d_best_median);

/* Replaces the per-point inner loop of ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th)
 * (src/ORBmatcher.cc:887-950), Fuse(KeyFrame*, cv::Mat Scw, ...) (:1044-1075) and both directions of SearchBySim3
 * (:1190-1224, :1270-1304): for every projected point, KeyFrame::GetFeaturesInArea(u, v, radius) (src/KeyFrame.cc:1138-1177),
 * features on levels outside [min_level, max_level] (= [predicted - 1, predicted]) skipped, the first feature of smallest
 * Hamming distance wins.  inv_level_sigma2 != NULL (host array, nlevels <= 16 entries = KeyFrame::mvInvLevelSigma2) turns on
 * the first Fuse's chi-square gate on the reprojection error: 7.8 over (u, v, proj_xr) when u_right[idx] >= 0, else 5.99
 * over (u, v) (:908-934).  Points do not close features to each other here, so best_idx[q] / best_dist[q] are per query:
 * -1 / 256 when the query is not ORBHIP_Q_ACTIVE or no feature is closer than 256; the caller applies <= TH_LOW (50) /
 * TH_HIGH (100) and the map updates.  Only u, v, radius, proj_xr, min_level, max_level and flags of a query are read. */
int orbhip_window_best(orbhip_ctx *ctx, const orbhip_keypoint *kps_un, const uint8_t *desc, int n, const float *u_right,
                       const float *inv_level_sigma2, int nlevels, float min_x, float min_y, float inv_w, float inv_h,
                       const orbhip_proj_query *queries, const uint8_t *qdesc, int nq, int32_t *best_idx,
                       int32_t *best_dist);
/* Batched, device-resident form (layouts as orbhip_search_by_projection_device); d_best_idx / d_best_dist [B][cap_q]. */
int orbhip_window_best_device(orbhip_ctx *ctx, const void *d_kps_un, const void *d_desc, int cap, int B,
                              const void *d_u_right, const float *inv_level_sigma2, int nlevels, float min_x, float min_y,
                              float inv_w, float inv_h, const void *d_cell_off, const void *d_cell_idx,
                              const void *d_queries, const void *d_qdesc, const void *d_nq, int cap_q, void *d_best_idx,
                              void *d_best_dist);

/* Replaces the body of ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched,
 * vector<int> &vnMatches12, int windowSize) (src/ORBmatcher.cc:405-520; the monocular initialiser, called at
 * src/Tracking.cc MonocularInitialization with nnratio 0.9, windowSize 100).  kps1_un / kps2_un are mvKeysUn of the two
 * frames; only octave-0 features of frame 1 search (:420-421), in index order, among the octave-0 features of frame 2
 * inside the window around prev_matched[i1] (:424); a feature of frame 2 that is already matched at a distance <= the
 * new one is skipped (:443-444), a better match displaces the earlier owner (:462-466); TH_LOW = 50 and
 * bestDist < bestDist2 * nnratio (:458-460); rotation histogram and three maxima when check_ori (:471-509).
 * matches12[i1] = feature of frame 2 or -1; prev_matched (n1 x 2 floats, in/out) is updated for the matched
 * features (:512-515); *nmatches = the reference's return value. */
int orbhip_search_for_initialization(orbhip_ctx *ctx, const orbhip_keypoint *kps1_un, const uint8_t *desc1, int n1,
                                     const orbhip_keypoint *kps2_un, const uint8_t *desc2, int n2, float min_x, float min_y,
                                     float inv_w, float inv_h, float *prev_matched, int window_size, float nnratio,
                                     int check_ori, int32_t *matches12, int *nmatches);
/* Batched, device-resident form: B frame pairs; frame 1 arrays [B][cap1], frame 2 arrays [B][cap2] with the grid
 * of frame 2 from orbhip_grid_build_device; d_prev_matched [B][cap1][2] floats (in/out), d_matches12 [B][cap1],
 * d_nmatches [B]. */
int orbhip_search_for_initialization_device(orbhip_ctx *ctx, const void *d_kps1_un, const void *d_desc1,
                                            const void *d_counts1, int cap1, const void *d_kps2_un, const void *d_desc2,
                                            const void *d_counts2, int cap2, int B, float min_x, float min_y, float inv_w,
                                            float inv_h, const void *d_cell_off2, const void *d_cell_idx2,
                                            void *d_prev_matched, int window_size, float nnratio, int check_ori,
                                            void *d_matches12, void *d_nmatches);

/* Replaces the matching core of ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12,
 * vector<pair<size_t,size_t>> &vMatchedPairs, bool bOnlyStereo) (src/ORBmatcher.cc:657-827, with
 * CheckDistEpipolarLine :140-157; called by LocalMapping::CreateNewMapPoints).  Both key frames as mvKeysUn,
 * descriptors, skip[i] != 0 <=> GetMapPoint(i) != NULL, mvuRight (NULL = monocular) and their FeatureVectors in CSR
 * form; F12 3x3 row-major float; (ex, ey) the epipole in the second image (:664-671, computed by the caller);
 * scale_factors2 / level_sigma2_2 = pKF2->mvScaleFactors / mvLevelSigma2.  TH_LOW = 50.  matches12[i1] = feature of
 * key frame 2 or -1 after the rotation histogram; *nmatches = the reference's return value.  (This fork does not mark
 * matched features of key frame 2, so one of them can be matched by several features of key frame 1.) */
int orbhip_search_for_triangulation(orbhip_ctx *ctx, const orbhip_keypoint *kps1_un, const uint8_t *desc1, int n1,
                                    const uint8_t *skip1, const float *u_right1, const int32_t *node1, const int32_t *off1,
                                    const int32_t *idx1, int ng1, const orbhip_keypoint *kps2_un, const uint8_t *desc2, int n2,
                                    const uint8_t *skip2, const float *u_right2, const int32_t *node2, const int32_t *off2,
                                    const int32_t *idx2, int ng2, const float F12[9], float ex, float ey,
                                    const float *scale_factors2, const float *level_sigma2_2, int nlevels2, int only_stereo,
                                    int check_ori, int32_t *matches12, int *nmatches);

/* ---- undistortion and rectification ----
 * Replaces the body of Frame::UndistortKeyPoints (src/Frame.cc:748-778): cv::undistortPoints(mat, mat, mK,
 * mDistCoef, cv::Mat(), mK) on the keypoint coordinates; every other field of a keypoint is copied.  K, P: 3x3
 * row-major float (P = NULL: normalised coordinates are returned, as OpenCV does without P; the reference passes
 * mK); dist: ndist in {0, 4, 5, 8} coefficients (k1, k2, p1, p2[, k3[, k4, k5, k6]]).  The caller keeps the
 * reference's shortcut "mDistCoef(0) == 0 -> mvKeysUn = mvKeys" (:750-754).  d_counts may be NULL (= cap points
 * per set).  Also serves Frame::ComputeImageBounds (:780-808: the four image corners as keypoints). */
int orbhip_undistort_keypoints(orbhip_ctx *ctx, const orbhip_keypoint *kps, int n, const float K[9], const float *dist,
                               int ndist, const float *P, orbhip_keypoint *kps_un);
int orbhip_undistort_keypoints_device(orbhip_ctx *ctx, const void *d_kps, const void *d_counts, int cap, int B,
                                      const float K[9], const float *dist, int ndist, const float *P, void *d_kps_un);
/* Replaces cv::initUndistortRectifyMap(K, D, R, P.rowRange(0,3).colRange(0,3), size, CV_32F, M1, M2)
 * (Examples/Stereo/stereo_euroc.cc:96-98): a once-per-run table built on the host in double. */
int orbhip_init_undistort_rectify_map(const double K[9], const double *dist, int ndist, const double R[9],
                                      const double P[9], int w, int h, float *map_x, float *map_y);
/* Replaces cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) (stereo_euroc.cc:136-137; 8-bit single channel,
 * BORDER_CONSTANT 0).  The maps (w x h floats each) are uploaded once per context; every call rectifies B images
 * with them.  The destination has the maps' size. */
int orbhip_remap_set_maps(orbhip_ctx *ctx, const float *map_x, const float *map_y, int w, int h);
int orbhip_remap(orbhip_ctx *ctx, const uint8_t *src, int src_w, int src_h, int src_stride, uint8_t *dst,
                 int dst_stride);
int orbhip_remap_device(orbhip_ctx *ctx, const void *d_src, int B, int src_w, int src_h, int src_stride,
                        size_t src_frame_stride, void *d_dst, int dst_stride, size_t dst_frame_stride);

/* Device time of the stages of the last extract call on this context, in ms:
 * {pyramid, FAST, quadtree, blur, describe} and, at [5], of the last orbhip_hamming_knn2*_device
 * call.  Measured with HIP events on the context's stream; synchronises the stream. */
int orbhip_get_stage_times(orbhip_ctx *ctx, float ms[6]);
/* Which of those events the extract / match calls record: 2 (default) all of them, 1 only the pair around the FAST launch
 * (the other entries read 0), 0 none.  An event between two kernels of a stream costs a few microseconds of device time;
 * a loop that only wants its throughput (and bench.py, which wants the FAST launch time) narrows the set. */
int orbhip_set_stage_timing(orbhip_ctx *ctx, int mode);
/* Scheduling of the batched path (affects speed only, never a result): where the Gaussian blur (ref:
 * src/ORBextractor.cc:1103-1104) runs.  0 (default): on the context's second stream behind FAST -- FAST, the kernel whose
 * roofline is reported, owns the device while it runs; the quadtree is cut into two half-batches, the first beside the blur,
 * the second beside the describe kernel of the first half; 1: one launch from the end of the pyramid, beside FAST and the
 * quadtree; 2: alone on the main stream between FAST and the quadtree (every kernel owns the device: per-kernel counters).
 * Also ORBHIP_BLUR_PLACE at context creation. */
int orbhip_set_blur_placement(orbhip_ctx *ctx, int place);

/* ---- resident feature sets (new) ----
 * A key frame's descriptors, keypoints, FeatureVector and feature grid never change after KeyFrame::KeyFrame / ComputeBoW
 * (ref: src/KeyFrame.cc:53-87, 392-400), but the reference's matcher entry points take the KeyFrame itself, so a per-call
 * C entry point has to upload that data again on every call.  A set keeps it on the device under a caller-chosen key
 * (non-zero; the drop-in classes use the KeyFrame's / Frame's mnId); at most 96 sets per context, least recently used out.
 *   orbhip_set_put   kps[n] / desc[n * 32]; the FeatureVector as CSR over ascending node ids (node[ng], off[ng + 1], idx[off[ng]];
 *                    ng may be 0); inv_w, inv_h > 0: the 64 x 48 grid of Frame::AssignFeaturesToGrid is built too (needed
 *                    by orbhip_window_best_set).  Replaces a set of the same key.
 *   orbhip_set_has   1 if a set with this key and this number of features is resident, else 0.
 *   orbhip_set_drop  key 0: all sets. */
int orbhip_set_put(orbhip_ctx *ctx, uint64_t key, const orbhip_keypoint *kps, const uint8_t *desc, int n, const int32_t *node,
                   const int32_t *off, const int32_t *idx, int ng, float min_x, float min_y, float inv_w, float inv_h);
int orbhip_set_has(orbhip_ctx *ctx, uint64_t key, int n);
int orbhip_set_drop(orbhip_ctx *ctx, uint64_t key);
/* Bounds the table: at most max_sets sets (clamped to 4 .. 96, the default) stay resident in this context, least recently used
 * out -- ~90 KB of device memory per set of 1000 features.  Returns the limit in force.  (The reference keeps every KeyFrame's
 * descriptors in host memory for the life of the map, src/KeyFrame.cc:55-89; this is the device-side counterpart's budget.) */
int orbhip_set_limit(orbhip_ctx *ctx, int max_sets);
/* Is the set what the caller thinks it is?  Ids are not identities: Tracking::Reset restarts KeyFrame::nNextId and
 * Frame::nNextId (ref: src/Tracking.cc:2758-2759), and a key frame met before KeyFrame::ComputeBoW (ref: src/KeyFrame.cc:392-400)
 * has an empty FeatureVector.
 *   orbhip_set_info         1 and the set's feature count, FeatureVector nodes and fingerprint when `key` is resident, else 0
 *                           (any of the three pointers may be NULL).  The wrappers of the *_sets entry points size their
 *                           buffers from it.
 *   orbhip_set_fingerprint  the fingerprint orbhip_set_put / orbhip_set_put_from_frame store: a hash of n, the first
 *                           keypoint's position and the first and last descriptor.  A caller compares the one of its own
 *                           data with the resident one and puts the set again when they differ.
 * The integration calls orbhip_set_drop(ctx, 0) from Tracking::Reset. */
int orbhip_set_info(orbhip_ctx *ctx, uint64_t key, int *n, int *ng, uint64_t *fingerprint);
uint64_t orbhip_set_fingerprint(const orbhip_keypoint *kps, const uint8_t *desc, int n);
/* the same from the three things it hashes (descriptor rows that are not contiguous: a cv::Mat with a step) */
uint64_t orbhip_set_fingerprint_rows(const orbhip_keypoint *first_kp, const uint8_t *first_desc, const uint8_t *last_desc, int n);
/* orbhip_search_by_bow (above) between two resident sets: valid1[n1] (and valid2[n2] or NULL) are the only per-feature
 * inputs that travel.  Same results as orbhip_search_by_bow on the sets' data. */
int orbhip_search_by_bow_sets(orbhip_ctx *ctx, uint64_t key1, const uint8_t *valid1, uint64_t key2, const uint8_t *valid2, int th,
                              int th_mode, float nnratio, int check_ori, int32_t *match12, int32_t *match21, int *nmatches);
/* orbhip_window_best (above) into a resident set (with a grid): the projected points travel, the key frame does not. */
int orbhip_window_best_set(orbhip_ctx *ctx, uint64_t key, const float *u_right, const float *inv_level_sigma2, int nlevels,
                           const orbhip_proj_query *queries, const uint8_t *qdesc, int nq, int32_t *best_idx, int32_t *best_dist);

/* ---- the Frame constructor's device work as one launch (new) ----
 * Frame::Frame (ref: src/Frame.cc:518-572) runs ExtractORB (:591-597), UndistortKeyPoints (:748-778) and AssignFeaturesToGrid
 * (:574-589) one after the other, and Tracking asks for Frame::ComputeBoW (:739-746) before the frame's first SearchByBoW
 * (ref: src/Tracking.cc, TrackReferenceKeyFrame).  As four entry points that is four launch + synchronise round trips for
 * one dependency chain; orbhip_frame_build runs the chain as ONE captured graph with one synchronisation and one packed
 * result block -- the kernels are those of orbhip_extract, orbhip_undistort_keypoints, orbhip_grid_build and
 * orbhip_vocab_transform, the results are theirs.
 *   fp->K, dist, ndist  mK / mDistCoef as for orbhip_undistort_keypoints (P = K); ndist == 0 or dist[0] == 0: kps_un = kps, the
 *                       reference's shortcut (:750-754)
 *   fp->min_x ... inv_h the grid of Frame::AssignFeaturesToGrid (mnMinX, mnMinY, mfGridElementWidthInv, ...HeightInv);
 *                       inv_w <= 0: no grid (cell_off / cell_idx may be NULL) -- the first frame of a run, whose image
 *                       bounds Frame::ComputeImageBounds derives after the extraction
 *   fp->levelsup        >= 0: ORBVocabulary::transform of the descriptors with this levelsup (a vocabulary must be
 *                       loaded); < 0: none (word_id / weight / node_id may be NULL)
 * Outputs: kps, kps_un, desc (capacity `cap` features), n_out; cell_off[3073] / cell_idx[n] as orbhip_grid_build;
 * word_id / weight / node_id [n] as orbhip_vocab_transform.  The result block stays on the device until the context's
 * next orbhip_frame_build: orbhip_set_put_from_frame makes it a resident set of any context of the same device (the
 * matcher's per-thread context in the drop-in classes) without the keypoints and descriptors travelling again;
 * orbhip_frame_fingerprint is the orbhip_set_fingerprint of the frame it holds (0: none).  The graph is captured at the first
 * call of an (image size, parameters) pair and replayed afterwards; ORBHIP_NO_GRAPH=1 keeps the eager sequence. */
typedef struct orbhip_frame_params {
    float K[9];
    float dist[8];
    int ndist;
    float min_x, min_y, inv_w, inv_h;
    int levelsup;
} orbhip_frame_params;
int orbhip_frame_build(orbhip_ctx *ctx, const uint8_t *img, int w, int h, int stride, const orbhip_frame_params *fp,
                       orbhip_keypoint *kps, orbhip_keypoint *kps_un, uint8_t *desc, int cap, int *n_out, int32_t *cell_off,
                       int32_t *cell_idx, int32_t *word_id, float *weight, int32_t *node_id);
uint64_t orbhip_frame_fingerprint(const orbhip_ctx *ctx);
/* orbhip_set_put with the frame that `src` built last: the FeatureVector (CSR as for orbhip_set_put; ng may be 0) is all
 * that travels.  ctx and src must be contexts of the same device (they may be the same context).  Several contexts may copy
 * the same frame, each from its own thread: src's next orbhip_frame_build waits for every one of the copies.  What the caller
 * orders itself, as with any two calls on one context: a copy of src's frame must not START (this call) while another thread
 * is inside orbhip_frame_build on src -- in the reference the thread that builds a Frame is the one that makes it a KeyFrame
 * (src/Tracking.cc, CreateNewKeyFrame). */
int orbhip_set_put_from_frame(orbhip_ctx *ctx, uint64_t key, orbhip_ctx *src, const int32_t *node, const int32_t *off,
                              const int32_t *idx, int ng);

/* The floor under a per-call entry point, in microseconds per call on this context's stream: mode 0 = an empty kernel and
 * one synchronisation; 1 = 4 KB copied in, the kernel, 4 KB copied out, one synchronisation; 2 = the kernel stores its
 * result to page-locked memory, one synchronisation.  (Measurement aid: tools/percall_latency.py.) */
int orbhip_debug_roundtrip(orbhip_ctx *ctx, int mode, int iters, double *us_per_call);

/* Which kernel variants this PROCESS has launched since the last reset (test aid: a test that claims to reach a fallback path
 * checks the bit).  Bits: 0 k_fast_fix, 1 k_fast (generic grid), 2 k_resize_fit, 3 k_resize<32> / <8>, 4 k_pyramid_chain,
 * 5 k_quadtree (tables in LDS), 6 k_quadtree (tables and candidates in LDS, a frame or two), 7 k_quadtree (tables in global
 * memory), 8 k_bow_lane, 9 k_bow_seq (descriptors in LDS), 10 k_bow_seq (descriptors in global memory), 11 k_fast_fix on the
 * tall-cell instance.  Returns the mask; reset != 0 clears it. */
unsigned orbhip_debug_path_mask(int reset);

/* ---- multi-GPU (one process per GPU) ----
 * The reference is a single process (no distributed back end); these entry points are what a
 * multi-GPU host adds around the unchanged per-frame path: frames or whole sequences are sharded over ranks with no
 * per-frame collective, the vocabulary travels once, and database-sharded brute force has one exchange step. */
/* RCCL communicator over the ranks of one node.  uid: 128-byte ncclUniqueId produced by
 * orbhip_comm_unique_id() on rank 0 and distributed by the caller (file, env, torch store).  nranks = 1 is valid (the
 * collectives then run on a one-rank communicator).  The communicator is destroyed by orbhip_comm_destroy or
 * orbhip_destroy. */
int orbhip_comm_unique_id(uint8_t uid[128]);
int orbhip_comm_init(orbhip_ctx *ctx, int rank, int nranks, const uint8_t uid[128]);
int orbhip_comm_destroy(orbhip_ctx *ctx);
/* Rank and size as the communicator reports them (ncclCommUserRank / ncclCommCount); 0 and 1 without a communicator. */
int orbhip_comm_info(orbhip_ctx *ctx, int *rank, int *nranks);
/* Broadcast of the ORB vocabulary blob (binary format of
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1727-1751, loaded at src/System.cc:331-346)
 * from `root` to every rank over xGMI.  d_buf: device pointer, nbytes on every rank.  Asynchronous on the context
 * stream. */
int orbhip_bcast_blob_device(orbhip_ctx *ctx, void *d_buf, size_t nbytes, int root);
/* Database-sharded brute force (the bookkeeping of src/ORBmatcher.cc:205-226 over a database
 * split by rows): every rank runs orbhip_hamming_knn2_device of the same nq queries against ITS rows (shard_offset =
 * global index of its first row; ranks hold increasing row ranges), then this call all-gathers the nq x 3 int32 results
 * (ncclAllGather, 12 bytes per query per rank) and min-merges them on the device with the reference's tie rule (strict
 * '<': lowest global row wins).  Outputs as orbhip_hamming_knn2_device, global indices, identical on every rank.
 * Asynchronous on the context stream.  Without a communicator (nranks = 1) it reduces to the index shift. */
int orbhip_knn2_allgather_merge_device(orbhip_ctx *ctx, const void *d_best_idx_local, const void *d_best_d_local,
                                       const void *d_second_d_local, int nq, int shard_offset, void *d_best_idx,
                                       void *d_best_d, void *d_second_d);
/* The merge alone, for hosts that exchange through their own collective (torch.distributed in bench.py / the tests):
 * d_parts = nshards consecutive parts of 3 * nq + 1 int32 each: best_idx[nq] | best_d[nq] | second_d[nq] | shard_offset,
 * in order of increasing shard_offset. */
int orbhip_knn2_merge_device(orbhip_ctx *ctx, const void *d_parts, int nshards, int nq, void *d_best_idx, void *d_best_d,
                             void *d_second_d);

#ifdef __cplusplus
}
#endif
#endif /* ORBHIP_H */
