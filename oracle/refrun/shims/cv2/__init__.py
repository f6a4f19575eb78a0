"""Stand-in for `cv2` (opencv-python ~=4.6, absent in every interpreter of the build
container) used ONLY by oracle/refrun/make_golden.py.

Exactly the three OpenCV functions the reference's hot path calls are restated in NumPy
from OpenCV's published definitions, plus ``imread`` via PIL:

* ``cvtColor(img, COLOR_BGR2GRAY)`` — 8-bit fixed point, 15-bit coefficients
  (BY15=3735, GY15=19235, RY15=9798, round-half-up, >>15) as in OpenCV 4.x imgproc.
* ``medianBlur(img, 3)``            — 3x3 median, BORDER_REPLICATE, u8/u16.
* ``Sobel(img, CV_32F, dx, dy, ksize=3, borderType=BORDER_REFLECT)`` — correlation with
  [-1 0 1] (x) [1 2 1]^T, no normalisation; BORDER_REFLECT == numpy pad mode "symmetric".

Pinned by the reference's own tests: Sobel (tests/.../utils/test_jacobian.py:49-73).
medianBlur / cvtColor values are NOT pinned by any reference fixture ("parity unpinned").
Test infrastructure — never imported by the product package.
"""
import numpy as np

COLOR_BGR2GRAY = 6
COLOR_BGR2RGB = 4
COLOR_RGB2BGR = 4
CV_32F = 5
CV_32FC1 = 5
BORDER_REPLICATE = 1
BORDER_REFLECT = 2
BORDER_REFLECT_101 = 4
IMREAD_UNCHANGED = -1
IMREAD_GRAYSCALE = 0
IMREAD_COLOR = 1
IMREAD_ANYDEPTH = 2
IMREAD_ANYCOLOR = 4


def cvtColor(src, code):
    src = np.asarray(src)
    if code == COLOR_BGR2GRAY:
        assert src.dtype == np.uint8 and src.ndim == 3 and src.shape[2] == 3
        b = src[..., 0].astype(np.int32)
        g = src[..., 1].astype(np.int32)
        r = src[..., 2].astype(np.int32)
        return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)
    if code == COLOR_BGR2RGB:
        return np.ascontiguousarray(src[..., ::-1])
    raise NotImplementedError(code)


def medianBlur(src, ksize):
    assert ksize == 3
    src = np.asarray(src)
    assert src.ndim == 2
    p = np.pad(src, 1, mode="edge")
    h, w = src.shape
    stack = np.stack([p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], axis=0)
    return np.sort(stack, axis=0)[4].astype(src.dtype)


def Sobel(src, ddepth, dx, dy, ksize=3, scale=1, delta=0, borderType=BORDER_REFLECT_101):
    assert ksize == 3 and ddepth in (CV_32F,) and (dx, dy) in ((1, 0), (0, 1))
    mode = {BORDER_REFLECT: "symmetric", BORDER_REFLECT_101: "reflect", BORDER_REPLICATE: "edge"}[borderType]
    a = np.pad(np.asarray(src).astype(np.float32), 1, mode=mode)
    h, w = np.asarray(src).shape
    def s(yy, xx):
        return a[yy:yy + h, xx:xx + w]
    if (dx, dy) == (1, 0):
        out = (s(0, 2) + 2 * s(1, 2) + s(2, 2)) - (s(0, 0) + 2 * s(1, 0) + s(2, 0))
    else:
        out = (s(2, 0) + 2 * s(2, 1) + s(2, 2)) - (s(0, 0) + 2 * s(0, 1) + s(0, 2))
    return out.astype(np.float32)


def imread(path, flags=IMREAD_COLOR):
    from PIL import Image
    im = Image.open(path)
    if flags in (IMREAD_ANYDEPTH, IMREAD_UNCHANGED) and im.mode in ("I;16", "I", "I;16B"):
        return np.array(im).astype(np.uint16)
    if flags == IMREAD_GRAYSCALE:
        raise NotImplementedError("grayscale imread not needed by the harness")
    arr = np.array(im.convert("RGB"))
    return np.ascontiguousarray(arr[..., ::-1])  # RGB -> BGR like OpenCV
