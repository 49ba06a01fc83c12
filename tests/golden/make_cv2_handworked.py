"""Hand-worked vectors for OpenCV 4.5.1's uint8 INTER_LINEAR resize (cv2.resize default), written out scalar by scalar
from the algorithm as published in modules/imgproc/src/resize.cpp -- an independent second derivation, NOT an output of
cv2 (absent from this image and from /root/reference; environment.yml:343 pins opencv-python-headless==4.5.1.48), so
`ss_ingest_u8` / oracle.frame_io stay "parity unpinned against cv2" until a cv2-produced vector exists.  What these pin
is that the vectorised numpy restatement (oracle/frame_io.py) and the HIP kernel (csrc/frameio.hip) implement THIS
arithmetic:

  resize.cpp, cv::resize():        scale_x = 1 / inv_scale_x (double);  INTER_LINEAR with an exact integer 2x2
                                   decimation is routed to INTER_AREA (resizeAreaFast_: (a + b + c + d + 2) >> 2)
  resize.cpp, general linear path: for every destination x:  fx = (float)((dx + 0.5) * scale_x - 0.5); sx = floor(fx);
                                   fx -= sx;  sx < 0 -> (sx, fx) = (0, 0);  sx >= ssize.width - 1 -> (sx, fx) = (width-1, 0)
                                   ialpha = saturate_cast<short>(fx' * 2048) with cvRound (ties to even), pair (1-fx, fx)
                                   for every destination y the same for (sy, fy) WITHOUT the clamp of the fraction; the two
                                   source rows are clip(sy) and clip(sy + 1)
  HResizeLinear<uchar,int,short>:  D[dx] = S[sx] * a0 + S[sx + 1] * a1                     (int, scaled by 2048)
  VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>:
                                   dst = uchar(( ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)

    python tests/golden/make_cv2_handworked.py      -> tests/golden/cv2_resize_handworked.json
"""
import json
import math
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


def f32(x):
    """round a Python float (double) to float32 and back: OpenCV keeps fx in a `float`."""
    return struct.unpack('f', struct.pack('f', x))[0]


def cv_round(x):
    """cvRound: nearest integer, ties to even (what saturate_cast<short>(float) uses)."""
    r = math.floor(x)
    d = x - r
    if d > 0.5 or (d == 0.5 and r % 2 == 1):
        r += 1
    return int(r)


def taps_x(src, dst):
    scale = 1.0 / (float(dst) / float(src))
    out = []
    for d in range(dst):
        fx = f32((d + 0.5) * scale - 0.5)
        sx = math.floor(fx)
        fx = f32(fx - sx)
        if sx < 0:
            sx, fx = 0, 0.0
        if sx >= src - 1:
            sx, fx = src - 1, 0.0
        a0 = cv_round(f32(f32(1.0 - fx) * 2048.0))
        a1 = cv_round(f32(fx * 2048.0))
        out.append((int(sx), a0, a1))
    return out


def taps_y(src, dst):
    scale = 1.0 / (float(dst) / float(src))
    out = []
    for d in range(dst):
        fy = f32((d + 0.5) * scale - 0.5)
        sy = math.floor(fy)
        fy = f32(fy - sy)
        b0 = cv_round(f32(f32(1.0 - fy) * 2048.0))
        b1 = cv_round(f32(fy * 2048.0))
        r0 = min(max(int(sy), 0), src - 1)
        r1 = min(max(int(sy) + 1, 0), src - 1)
        out.append((r0, r1, b0, b1))
    return out


def resize(img, dw, dh):
    """img: list of rows of pixels (lists of channel ints)."""
    sh, sw, ch = len(img), len(img[0]), len(img[0][0])
    if (dw, dh) == (sw, sh):
        return [[list(p) for p in row] for row in img]
    if sw == 2 * dw and sh == 2 * dh:
        return [[[(img[2 * y][2 * x][c] + img[2 * y][2 * x + 1][c] + img[2 * y + 1][2 * x][c] + img[2 * y + 1][2 * x + 1][c] + 2) >> 2
                  for c in range(ch)] for x in range(dw)] for y in range(dh)]
    tx, ty = taps_x(sw, dw), taps_y(sh, dh)
    out = []
    for (r0, r1, b0, b1) in ty:
        row = []
        for (sx, a0, a1) in tx:
            sx1 = min(sx + 1, sw - 1)
            px = []
            for c in range(ch):
                s0 = img[r0][sx][c] * a0 + img[r0][sx1][c] * a1
                s1 = img[r1][sx][c] * a0 + img[r1][sx1][c] * a1
                v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
                px.append(min(max(v, 0), 255))
            row.append(px)
        out.append(row)
    return out


def image(h, w, seed):
    """small deterministic test image without numpy: a linear congruential sequence"""
    s = seed
    img = []
    for y in range(h):
        row = []
        for x in range(w):
            px = []
            for c in range(3):
                s = (s * 1103515245 + 12345) & 0x7FFFFFFF
                px.append((s >> 16) & 0xFF)
            row.append(px)
        img.append(row)
    return img


CASES = [
    ('area_2x2', 4, 6, 3, 2, 11),          # exact 2x2 decimation -> INTER_AREA fast path
    ('ratio_5_3', 3, 5, 3, 2, 12),         # 5 -> 3 columns (scale 1.6667), 3 -> 2 rows (1.5)
    ('ratio_8_3', 6, 16, 6, 4, 13),        # 16 -> 6 columns = the 1280 -> 480 ratio, 6 -> 4 rows = the 720 -> 480... 3:2 ratio
    ('upscale', 3, 4, 7, 5, 14),           # enlargement: clamped taps at both borders
    ('extremes', 4, 7, 3, 3, 15),          # 0 / 255 checkerboard (saturation, rounding at the ends of the range)
]

if __name__ == '__main__':
    out = {'provenance': __doc__, 'cases': {}}
    for name, sh, sw, dw, dh, seed in CASES:
        img = image(sh, sw, seed)
        if name == 'extremes':
            img = [[[255 if (x + y + c) % 2 else 0 for c in range(3)] for x in range(sw)] for y in range(sh)]
        out['cases'][name] = {'src': img, 'dsize': [dw, dh], 'dst': resize(img, dw, dh)}
    # the tap tables of the benchmark geometry (1280 -> 480, 720 -> 360 is the area path; 1080 -> 360 rows for 1080p input)
    out['taps_x_1280_480_first8'] = taps_x(1280, 480)[:8]
    out['taps_x_1280_480_last4'] = taps_x(1280, 480)[-4:]
    out['taps_y_1080_360_first4'] = taps_y(1080, 360)[:4]
    with open(os.path.join(HERE, 'cv2_resize_handworked.json'), 'w') as f:
        json.dump(out, f, indent=0)
    print('wrote', len(out['cases']), 'cases;', out['taps_x_1280_480_first8'][:3])
