// api_stereo.hip -- C ABI, part 5: Frame::ComputeStereoMatches, keypoint undistortion and stereo rectification.
#include "api_common.h"

// ------------------------------------------------------------------------------------------------
// stereo
// ------------------------------------------------------------------------------------------------
extern "C" int orbhip_stereo_match_device(orbhip_ctx *L, orbhip_ctx *R, const void *d_kpsL, const void *d_descL,
                                          const void *d_cntL, const void *d_kpsR, const void *d_descR,
                                          const void *d_cntR, int cap, int B, float mb, float mbf, void *d_uRight,
                                          void *d_depth, void *d_nmatch)
{
    if (!L || !R || !d_kpsL || !d_descL || !d_cntL || !d_kpsR || !d_descR || !d_cntR || cap <= 0 || B <= 0 ||
        !d_uRight || !d_depth || !d_nmatch || !(mb > 0.f) || !(mbf > 0.f))
        return fail(L, ORBHIP_E_ARG, "orbhip_stereo_match_device: bad argument");
    if (!L->last_lvl0 || !R->last_lvl0 || L->cur_w != R->cur_w || L->cur_h != R->cur_h || L->nlevels != R->nlevels ||
        B > L->last_B || B > R->last_B || L->device != R->device || L->cur_h > 4095 || cap > 65535)
        return fail(L, ORBHIP_E_ARG, "orbhip_stereo_match_device: both contexts must have just extracted images of the "
                                     "same size (at most 4095 rows, at most 65535 keypoints per image) on the same device");
    HIPCHK(L, hipSetDevice(L->device));
    int rc;
    if ((rc = orb_match_scratch(L, stereo_scratch_bytes(B, cap)))) return rc;
    // the right pyramid / keypoints are produced on the right context's stream
    HIPCHK(L, hipEventRecord(R->evx[0], R->stream));
    HIPCHK(L, hipStreamWaitEvent(L->stream, R->evx[0], 0));
    launch_stereo(L, R, (const orbhip_keypoint *)d_kpsL, (const uint8_t *)d_descL, (const int32_t *)d_cntL,
                  (const orbhip_keypoint *)d_kpsR, (const uint8_t *)d_descR, (const int32_t *)d_cntR, cap, B, mb, mbf,
                  (float *)d_uRight, (float *)d_depth, (int32_t *)L->d_match, (int32_t *)d_nmatch);
    HIPCHK(L, hipGetLastError());
    // ... and the right context must not overwrite its pyramid before these kernels have read it
    HIPCHK(L, hipEventRecord(L->evx[0], L->stream));
    HIPCHK(L, hipStreamWaitEvent(R->stream, L->evx[0], 0));
    return ORBHIP_OK;
}

extern "C" int orbhip_stereo_match(orbhip_ctx *L, orbhip_ctx *R, const orbhip_keypoint *kpsL, const uint8_t *descL, int nL,
                                   const orbhip_keypoint *kpsR, const uint8_t *descR, int nR, float mb, float mbf,
                                   float *mvuRight, float *mvDepth, int *nmatch)
{
    if (!L || !R || nL < 0 || nR < 0 || (nL > 0 && (!kpsL || !descL || !mvuRight || !mvDepth)) || (nR > 0 && (!kpsR || !descR)))
        return fail(L, ORBHIP_E_ARG, "orbhip_stereo_match: bad argument");
    if (nmatch) *nmatch = 0;
    for (int i = 0; i < nL; i++) mvuRight[i] = mvDepth[i] = -1.0f;
    if (nL == 0 || nR == 0) return ORBHIP_OK;
    HIPCHK(L, hipSetDevice(L->device));
    const int cap = std::max(nL, nR);
    Packed P(L);
    int rc;
    if ((rc = P.begin((size_t)cap * (28 + 32) * 2 + (size_t)cap * 8 + 16 * 256))) return rc;
    // (device arrays of `cap` slots each; only the first nL / nR entries travel)
    const int32_t cnts[4] = {nL, nR, 0, 0};
    const orbhip_keypoint *dkL = (const orbhip_keypoint *)P.in(kpsL, (size_t)nL * 28);
    P.off += (size_t)(cap - nL) * 28;
    const orbhip_keypoint *dkR = (const orbhip_keypoint *)P.in(kpsR, (size_t)nR * 28);
    P.off += (size_t)(cap - nR) * 28;
    const uint8_t *ddL = (const uint8_t *)P.in(descL, (size_t)nL * 32);
    P.off += (size_t)(cap - nL) * 32;
    const uint8_t *ddR = (const uint8_t *)P.in(descR, (size_t)nR * 32);
    P.off += (size_t)(cap - nR) * 32;
    int32_t *dc = (int32_t *)P.in(cnts, 16);                     // nL | nR | matches before the median cut (comes back)
    float *du = (float *)P.out((size_t)cap * 4), *dz = (float *)P.out((size_t)cap * 4);
    P.inEnd = (size_t)((uint8_t *)dc - P.d) + 16;
    if ((rc = P.upload())) return rc;
    if ((rc = orbhip_stereo_match_device(L, R, dkL, ddL, dc, dkR, ddR, dc + 1, cap, 1, mb, mbf, du, dz, dc + 2))) return rc;
    if ((rc = P.download(dc))) return rc;
    memcpy(mvuRight, P.host(du), (size_t)nL * 4);
    memcpy(mvDepth, P.host(dz), (size_t)nL * 4);
    if (nmatch) *nmatch = ((const int32_t *)P.host(dc))[2];
    return ORBHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// undistortion / rectification
// ------------------------------------------------------------------------------------------------
static bool dist_ok(const float *K, const float *dist, int ndist)
{
    return K && K[0] != 0.f && K[4] != 0.f && (ndist == 0 || ((ndist == 4 || ndist == 5 || ndist == 8) && dist));
}

extern "C" int orbhip_undistort_keypoints_device(orbhip_ctx *c, const void *d_kps, const void *d_counts, int cap, int B,
                                                 const float K[9], const float *dist, int ndist, const float *P,
                                                 void *d_kps_un)
{
    if (!c || !d_kps || !d_kps_un || cap <= 0 || B <= 0 || !dist_ok(K, dist, ndist))
        return fail(c, ORBHIP_E_ARG, "orbhip_undistort_keypoints_device: bad argument");
    HIPCHK(c, orb_enter(c));
    launch_undistort(c->stream, (const orbhip_keypoint *)d_kps, (const int32_t *)d_counts, cap, B, K, dist, ndist, P,
                     (orbhip_keypoint *)d_kps_un);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}

extern "C" int orbhip_undistort_keypoints(orbhip_ctx *c, const orbhip_keypoint *kps, int n, const float K[9],
                                          const float *dist, int ndist, const float *P, orbhip_keypoint *kps_un)
{
    if (!c || n < 0 || (n > 0 && (!kps || !kps_un)) || !dist_ok(K, dist, ndist))
        return fail(c, ORBHIP_E_ARG, "orbhip_undistort_keypoints: bad argument");
    if (n == 0) return ORBHIP_OK;
    HIPCHK(c, orb_enter(c));
    TmpDev T(c);
    int rc;
    if ((rc = T.reserve((size_t)n * 56 + 1024))) return rc;
    orbhip_keypoint *di = (orbhip_keypoint *)T.take((size_t)n * 28), *dout = (orbhip_keypoint *)T.take((size_t)n * 28);
    TMPCHK(c, T);
    HIPCHK(c, hipMemcpyAsync(di, kps, (size_t)n * 28, hipMemcpyHostToDevice, c->stream));
    if ((rc = orbhip_undistort_keypoints_device(c, di, nullptr, n, 1, K, dist, ndist, P, dout))) return rc;
    HIPCHK(c, hipMemcpyAsync(kps_un, dout, (size_t)n * 28, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBHIP_OK;
}

extern "C" int orbhip_init_undistort_rectify_map(const double K[9], const double *dist, int ndist, const double R[9],
                                                 const double P[9], int w, int h, float *map_x, float *map_y)
{
    if (!K || !R || !P || w <= 0 || h <= 0 || !map_x || !map_y || ndist < 0 || (ndist > 0 && !dist)) return ORBHIP_E_ARG;
    orb_init_undistort_rectify_map(K, dist, ndist, R, P, w, h, map_x, map_y);
    return ORBHIP_OK;
}

extern "C" int orbhip_remap_set_maps(orbhip_ctx *c, const float *map_x, const float *map_y, int w, int h)
{
    if (!c || !map_x || !map_y || w <= 0 || h <= 0) return fail(c, ORBHIP_E_ARG, "orbhip_remap_set_maps: bad argument");
    HIPCHK(c, orb_enter(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_maps) HIPCHK(c, hipFree(c->d_maps));
    c->d_maps = nullptr;
    c->map_w = c->map_h = 0;
    const size_t n = (size_t)w * h;
    HIPCHK(c, hipMalloc((void **)&c->d_maps, 2 * n * sizeof(float) + 64));
    HIPCHK(c, hipMemcpy(c->d_maps, map_x, n * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_maps + n, map_y, n * sizeof(float), hipMemcpyHostToDevice));
    c->map_w = w;
    c->map_h = h;
    return ORBHIP_OK;
}

extern "C" int orbhip_remap_device(orbhip_ctx *c, const void *d_src, int B, int src_w, int src_h, int src_stride,
                                   size_t src_frame_stride, void *d_dst, int dst_stride, size_t dst_frame_stride)
{
    if (!c || !d_src || !d_dst || B <= 0 || src_w <= 0 || src_h <= 0 || src_stride < src_w || src_w > 32767 || src_h > 32767)
        return fail(c, ORBHIP_E_ARG, "orbhip_remap_device: bad argument");
    if (!c->d_maps) return fail(c, ORBHIP_E_ARG, "orbhip_remap_device: no maps (orbhip_remap_set_maps)");
    if (dst_stride < c->map_w) return fail(c, ORBHIP_E_ARG, "orbhip_remap_device: dst_stride smaller than the map width");
    HIPCHK(c, orb_enter(c));
    launch_remap(c->stream, (const uint8_t *)d_src, B, src_w, src_h, src_stride, src_frame_stride, c->d_maps,
                 c->d_maps + (size_t)c->map_w * c->map_h, c->map_w, c->map_h, (uint8_t *)d_dst, dst_stride, dst_frame_stride);
    HIPCHK(c, hipGetLastError());
    return ORBHIP_OK;
}

extern "C" int orbhip_remap(orbhip_ctx *c, const uint8_t *src, int src_w, int src_h, int src_stride, uint8_t *dst,
                            int dst_stride)
{
    if (!c || !src || !dst || src_w <= 0 || src_h <= 0 || src_stride < src_w)
        return fail(c, ORBHIP_E_ARG, "orbhip_remap: bad argument");
    if (!c->d_maps) return fail(c, ORBHIP_E_ARG, "orbhip_remap: no maps (orbhip_remap_set_maps)");
    if (dst_stride < c->map_w) return fail(c, ORBHIP_E_ARG, "orbhip_remap: dst_stride smaller than the map width");
    HIPCHK(c, orb_enter(c));
    TmpDev T(c);
    int rc;
    const size_t sbytes = (size_t)src_stride * src_h, dpitch = align_up((size_t)c->map_w, 64), dbytes = dpitch * c->map_h;
    if ((rc = T.reserve(sbytes + dbytes + 1024))) return rc;
    uint8_t *ds = (uint8_t *)T.take(sbytes), *dd = (uint8_t *)T.take(dbytes);
    TMPCHK(c, T);
    HIPCHK(c, hipMemcpyAsync(ds, src, sbytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = orbhip_remap_device(c, ds, 1, src_w, src_h, src_stride, sbytes, dd, (int)dpitch, dbytes))) return rc;
    HIPCHK(c, hipMemcpy2DAsync(dst, dst_stride, dd, dpitch, c->map_w, c->map_h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ORBHIP_OK;
}

