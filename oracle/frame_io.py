"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the frame I/O either side of the hot path.

Reference call sites: Full_model_inference/Codes/test_online_tra.py:252-278 (cv2.imread -> HR fp32 CHW in 0..255 and
`cv2.resize(img, (480, 360))` -> fp32 CHW `/127.5 - 1.0`) and :409-417 (`stable_list[k].astype(np.uint8)` -> VideoWriter).

PARITY UNPINNED for `cv2_resize_linear_u8`: the algorithm lives in a third-party dependency that is absent from
/root/reference AND from this image (environment.yml:343 pins opencv-python-headless==4.5.1.48; no cv2 module or
libopencv exists here), so this function restates OpenCV 4.5.1's published algorithm
(modules/imgproc/src/resize.cpp: cv::hal::resize -> resizeGeneric_<HResizeLinear<uchar,int,short,2048,...>,
VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>,...>>, and resizeAreaFast_ for the exact 2x2 case) from the
source as documented, and is checked through properties (identity, exact 2x2 area, +-1 LSB of real bilinear,
constant images) and against a SECOND, scalar-by-scalar hand derivation of the same published arithmetic
(tests/golden/make_cv2_handworked.py -> cv2_resize_handworked.json: 2x2 area route, 5:3 / 8:3 / 3:2 ratios, an
enlargement, a 0/255 checkerboard, the 1280 -> 480 tap table).  Neither is an output of cv2.  IPP is not in play: for 8-bit linear resize
OpenCV skips IPP unless `ipp::useIPP_NotExact()` (resize.cpp, ipp_resize).
"""
import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def _sat_short_round(v32):
    """saturate_cast<short>(float): cvRound (ties to even) then clamp."""
    return np.clip(np.rint(v32.astype(np.float32)), -32768, 32767).astype(np.int32)


def linear_tables(src, dst):
    """(ofs[dst], w0[dst], w1[dst]) along one axis as resize.cpp builds them for the x axis
    (fx clamped to 0 where the 2-tap window leaves the image); the y axis uses `linear_tables_y`."""
    scale = 1.0 / (float(dst) / float(src))                       # double, as `1./inv_scale_x`
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0
    s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0.0
    s[hi] = src - 1
    w0 = _sat_short_round((np.float32(1.0) - f) * np.float32(INTER_RESIZE_COEF_SCALE))
    w1 = _sat_short_round(f * np.float32(INTER_RESIZE_COEF_SCALE))
    return s, w0, w1


def linear_tables_y(src, dst):
    """y axis: the fraction is NOT clamped, the two source rows are (clip(sy), clip(sy+1))."""
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    b0 = _sat_short_round((np.float32(1.0) - f) * np.float32(INTER_RESIZE_COEF_SCALE))
    b1 = _sat_short_round(f * np.float32(INTER_RESIZE_COEF_SCALE))
    r0 = np.clip(s, 0, src - 1)
    r1 = np.clip(s + 1, 0, src - 1)
    return r0, r1, b0, b1


def cv2_resize_linear_u8(img, dsize):
    """cv2.resize(img, dsize=(w, h)) with the default INTER_LINEAR for uint8 HWC images (OpenCV 4.5.1)."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    sh, sw, _ = img.shape
    dw, dh = int(dsize[0]), int(dsize[1])
    if (dw, dh) == (sw, sh):
        return img.copy()
    if sw == 2 * dw and sh == 2 * dh:      # INTER_LINEAR with iscale 2x2 is routed to INTER_AREA (fast): (a+b+c+d+2)>>2
        s = img.astype(np.int32)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    xo, a0, a1 = linear_tables(sw, dw)
    r0, r1, b0, b1 = linear_tables_y(sh, dh)
    s = img.astype(np.int32)
    x1 = np.minimum(xo + 1, sw - 1)
    hrow = s[:, xo, :] * a0[None, :, None] + s[:, x1, :] * a1[None, :, None]          # [sh, dw, c] ints (x 2048)
    top = (b0[:, None, None] * (hrow[r0] >> 4)) >> 16
    bot = (b1[:, None, None] * (hrow[r1] >> 4)) >> 16
    return np.clip((top + bot + 2) >> 2, 0, 255).astype(np.uint8)


def load_frame(img_u8, lr_h=360, lr_w=480):
    """test_online_tra.py:252-264 for one decoded frame (HWC uint8, BGR as cv2 returns it)
    -> (hr [3,H,W] fp32 in 0..255, lr [3,lr_h,lr_w] fp32 in [-1,1])."""
    hr = np.transpose(img_u8.astype(np.float32), [2, 0, 1])
    lr = cv2_resize_linear_u8(img_u8, (lr_w, lr_h)).astype(np.float32)
    lr = np.transpose(lr, [2, 0, 1])
    lr = (lr / np.float32(127.5)) - np.float32(1.0)
    return hr, lr.astype(np.float32)


def to_video_frame(fused_chw):
    """test_online_tra.py:151 + :413: `[3,H,W]` fp32 -> HWC -> `.astype(np.uint8)`.  For the values the path
    produces (0 <= v < 256) this is truncation toward zero; outside that range numpy's cast goes through int32
    on x86-64 (cvttss2si) and keeps the low 8 bits, which is what is restated here."""
    hwc = np.transpose(np.asarray(fused_chw, dtype=np.float32), [1, 2, 0])
    return (np.trunc(hwc).astype(np.int64) & 0xFF).astype(np.uint8)
