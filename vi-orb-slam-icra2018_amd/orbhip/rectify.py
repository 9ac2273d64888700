"""Python mirror of the steps either side of extraction:
Frame::UndistortKeyPoints / ComputeImageBounds (src/Frame.cc:748-808) and the stereo driver's rectification
(Examples/Stereo/stereo_euroc.cc:96-98, :136-137).  All per-frame arithmetic runs in liborbhip."""
import numpy as np

from . import capi
from .capi import KP_DTYPE, _p, check

f32 = np.float32


def UndistortKeyPoints(ctx, kps, K, dist_coef):
    """mvKeysUn from mvKeys: cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK)."""
    kps = np.ascontiguousarray(kps, KP_DTYPE)
    D = np.ascontiguousarray(dist_coef, f32).ravel()
    if len(D) == 0 or D[0] == 0.0:                          # ref: src/Frame.cc:750-754
        return kps.copy()
    K = np.ascontiguousarray(K, f32).reshape(9)
    out = np.empty_like(kps)
    check(capi.load().orbhip_undistort_keypoints(ctx.handle, _p(kps), len(kps), _p(K), _p(D), len(D), _p(K), _p(out)),
          ctx.handle, "orbhip_undistort_keypoints")
    return out


def undistort_points(ctx, xy, K, dist_coef, P=None):
    """cv::undistortPoints on bare (x, y) pairs."""
    xy = np.ascontiguousarray(xy, f32).reshape(-1, 2)
    kps = np.zeros(len(xy), KP_DTYPE)
    kps["x"], kps["y"] = xy[:, 0], xy[:, 1]
    K = np.ascontiguousarray(K, f32).reshape(9)
    D = np.ascontiguousarray(dist_coef, f32).ravel()
    Pm = None if P is None else np.ascontiguousarray(P, f32).reshape(9)
    out = np.empty_like(kps)
    check(capi.load().orbhip_undistort_keypoints(ctx.handle, _p(kps), len(kps), _p(K), _p(D) if len(D) else None, len(D),
                                                 _p(Pm), _p(out)), ctx.handle, "orbhip_undistort_keypoints")
    return np.stack([out["x"], out["y"]], 1)


def ComputeImageBounds(ctx, cols, rows, K, dist_coef):
    """(mnMinX, mnMaxX, mnMinY, mnMaxY) -- src/Frame.cc:780-808."""
    D = np.ascontiguousarray(dist_coef, f32).ravel()
    if len(D) == 0 or D[0] == 0.0:
        return f32(0), f32(cols), f32(0), f32(rows)
    m = undistort_points(ctx, [[0, 0], [cols, 0], [0, rows], [cols, rows]], K, D, K)
    return min(m[0, 0], m[2, 0]), max(m[1, 0], m[3, 0]), min(m[0, 1], m[1, 1]), max(m[2, 1], m[3, 1])


def initUndistortRectifyMap(K, D, R, P, w, h):
    """cv::initUndistortRectifyMap(K, D, R, P.rowRange(0,3).colRange(0,3), (w, h), CV_32F) -> (M1, M2)."""
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    D = np.ascontiguousarray(D, np.float64).ravel()
    R = np.ascontiguousarray(R, np.float64).reshape(9)
    P = np.ascontiguousarray(np.asarray(P, np.float64).reshape(3, -1)[:, :3]).reshape(9)
    mx = np.empty((h, w), f32)
    my = np.empty((h, w), f32)
    rc = capi.load().orbhip_init_undistort_rectify_map(_p(K), _p(D) if len(D) else None, len(D), _p(R), _p(P), w, h, _p(mx),
                                                       _p(my))
    if rc != 0:
        raise capi.OrbHipError("orbhip_init_undistort_rectify_map failed (%d)" % rc)
    return mx, my


class Rectifier:
    """cv::remap(im, imRect, M1, M2, cv::INTER_LINEAR) with the maps resident on the device."""

    def __init__(self, ctx, map_x, map_y):
        self._ctx = ctx
        self._L = capi.load()
        mx = np.ascontiguousarray(map_x, f32)
        my = np.ascontiguousarray(map_y, f32)
        assert mx.shape == my.shape and mx.ndim == 2
        self.h, self.w = mx.shape
        check(self._L.orbhip_remap_set_maps(ctx.handle, _p(mx), _p(my), self.w, self.h), ctx.handle, "orbhip_remap_set_maps")

    def __call__(self, image):
        img = np.asarray(image)
        if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1 or img.strides[0] < img.shape[1]:
            img = np.ascontiguousarray(image, np.uint8)          # row-padded views are passed as they are
        out = np.empty((self.h, self.w), np.uint8)
        check(self._L.orbhip_remap(self._ctx.handle, _p(img), img.shape[1], img.shape[0], img.strides[0], _p(out), self.w),
              self._ctx.handle, "orbhip_remap")
        return out

    def remap_device(self, d_src, B, src_w, src_h, src_stride, src_frame_stride, d_dst, dst_stride, dst_frame_stride):
        check(self._L.orbhip_remap_device(self._ctx.handle, d_src, B, src_w, src_h, src_stride, src_frame_stride, d_dst,
                                          dst_stride, dst_frame_stride), self._ctx.handle, "orbhip_remap_device")
