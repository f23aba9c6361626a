"""NumPy restatement of the keypoint and descriptor stages of OpenCV's SIFT.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: ``features.py:192-201`` calls
``cv2.xfeatures2d.SIFT_create().detectAndCompute(img, None)`` and then takes the
RootSIFT of the descriptors; the SIFT arithmetic is inside OpenCV, which the
reference does not pin and which is not installed here.  What is restated is the
published algorithm of OpenCV 3.4 / 4.x ``xfeatures2d/src/sift.cpp`` with the
defaults the reference uses (nfeatures 0, 3 octave layers, contrast threshold
0.04, edge threshold 10, sigma 1.6, first octave -1):

  findScaleSpaceExtrema  26-neighbour extrema of the DoG layers above the
                         pre-threshold floor(0.5 * 0.04 / 3 * 255)
  adjustLocalExtrema     up to 5 Newton steps on the 3-D quadratic fit (float
                         LU solve), contrast and edge-response tests
  calcOrientationHist    36-bin gradient histogram, radius round(4.5 s),
                         Gaussian 1.5 s, [1 4 6 4 1]/16 smoothing, peaks >= 80 %
                         with parabolic refinement -> one keypoint per peak
  removeDuplicatedSorted sort by (x, y, -size, angle, -response, -octave), drop
                         repeats; then x, y, size halved (first octave -1)
  calcSIFTDescriptor     4 x 4 x 8 histogram, trilinear interpolation, clip at
                         0.2, scale by 512, saturate to uint8 (kept as float)

``cv::fastAtan2`` is its degree-7 polynomial (0.3 degree accuracy); ``exp`` is the
libm one (OpenCV's table-driven ``exp32f`` agrees to ~1e-7).  Everything is
float32 where OpenCV computes in float.  It checks ``pano360_amd.features``
(HIP) in tests/; loops are plain Python, so keep images small.

Known answers (round 6; tests/test_oracle_golden.py::test_sift_oracle_known_answers, and the same
checks on the HIP kernels: tests/test_gpu_parity.py::test_sift_kernels_known_answers) - what can be
known about the keypoint stages without OpenCV: a Gaussian blob of standard deviation s is found at
its centre (+ 0.25 px on both axes: the doubled first octave's pixel-centre alignment) with
size / 2 = s / sqrt(k), k = 2^(1/3) - the closed-form maximum over scale of the difference of
Gaussians, labelled with the lower scale of the pair; a weak blob on a strong linear ramp gets the
ramp's direction as its angle (to 4 degrees: 36 bins + the parabola); a quarter turn of the image
turns the keypoints and leaves their descriptors alone (median distance 0 of norm 512).
"""
import numpy as np

F = np.float32
N_LAYERS = 3
CONTRAST_THR = 0.04
EDGE_THR = 10.0
SIGMA = 1.6
IMG_BORDER = 5
MAX_INTERP_STEPS = 5
ORI_HIST_BINS = 36
ORI_SIG_FCTR = 1.5
ORI_RADIUS = 3 * ORI_SIG_FCTR
ORI_PEAK_RATIO = 0.8
DESCR_WIDTH = 4
DESCR_HIST_BINS = 8
DESCR_SCL_FCTR = 3.0
DESCR_MAG_THR = 0.2
INT_DESCR_FCTR = 512.0
FLT_EPSILON = np.finfo(np.float32).eps

_P1 = F(0.9997878412794807 * (180 / np.pi))
_P3 = F(-0.3258083974640975 * (180 / np.pi))
_P5 = F(0.1555786518463281 * (180 / np.pi))
_P7 = F(-0.04432655554792128 * (180 / np.pi))


def fast_atan2(y, x):
    """cv::fastAtan2 (degrees in [0, 360)), elementwise on float32 arrays."""
    y, x = np.asarray(y, F), np.asarray(x, F)
    ax, ay = np.abs(x), np.abs(y)
    eps = F(2.220446049250313e-16)
    swap = ax < ay
    num = np.where(swap, ax, ay)
    den = np.where(swap, ay, ax) + eps
    c = (num / den).astype(F)
    c2 = (c * c).astype(F)
    a = ((((_P7 * c2 + _P5) * c2 + _P3) * c2 + _P1) * c).astype(F)
    a = np.where(swap, F(90.0) - a, a)
    a = np.where(x < 0, F(180.0) - a, a)
    a = np.where(y < 0, F(360.0) - a, a)
    return a.astype(F)


def cv_round(v):
    return int(np.rint(v))


def _solve3(h, b):
    """Matx33f::solve(b, DECOMP_LU): float32 Gaussian elimination with partial
    pivoting; None when singular."""
    a = np.array(h, dtype=F).copy()
    x = np.array(b, dtype=F).copy()
    for i in range(3):
        k = i + int(np.argmax(np.abs(a[i:, i])))
        if np.abs(a[k, i]) < FLT_EPSILON:
            return None
        if k != i:
            a[[i, k]] = a[[k, i]]
            x[[i, k]] = x[[k, i]]
        d = F(-1.0) / a[i, i]
        for j in range(i + 1, 3):
            alpha = F(a[j, i] * d)
            a[j, i:] = (a[j, i:] + alpha * a[i, i:]).astype(F)
            x[j] = F(x[j] + alpha * x[i])
    out = np.zeros(3, F)
    for i in range(2, -1, -1):
        s = x[i]
        for k in range(i + 1, 3):
            s = F(s - a[i, k] * out[k])
        out[i] = F(s / a[i, i])
    return out


def adjust_local_extrema(dog_oct, octv, layer, r, c):
    """sift.cpp adjustLocalExtrema.  dog_oct: list of the octave's DoG planes.
    Returns None or a dict with the refined keypoint (octave coordinates kept)."""
    img_scale = F(1.0 / 255.0)
    deriv_scale = F(img_scale * F(0.5))
    second_deriv_scale = img_scale
    cross_deriv_scale = F(img_scale * F(0.25))
    rows, cols = dog_oct[0].shape
    xi = xr = xc = F(0)
    for step in range(MAX_INTERP_STEPS + 1):
        if step == MAX_INTERP_STEPS:
            return None
        img, prv, nxt = dog_oct[layer], dog_oct[layer - 1], dog_oct[layer + 1]
        dd = np.array([(img[r, c + 1] - img[r, c - 1]) * deriv_scale,
                       (img[r + 1, c] - img[r - 1, c]) * deriv_scale,
                       (nxt[r, c] - prv[r, c]) * deriv_scale], F)
        v2 = F(img[r, c] * 2)
        dxx = F((img[r, c + 1] + img[r, c - 1] - v2) * second_deriv_scale)
        dyy = F((img[r + 1, c] + img[r - 1, c] - v2) * second_deriv_scale)
        dss = F((nxt[r, c] + prv[r, c] - v2) * second_deriv_scale)
        dxy = F((img[r + 1, c + 1] - img[r + 1, c - 1] - img[r - 1, c + 1] + img[r - 1, c - 1])
                * cross_deriv_scale)
        dxs = F((nxt[r, c + 1] - nxt[r, c - 1] - prv[r, c + 1] + prv[r, c - 1])
                * cross_deriv_scale)
        dys = F((nxt[r + 1, c] - nxt[r - 1, c] - prv[r + 1, c] + prv[r - 1, c])
                * cross_deriv_scale)
        hess = np.array([[dxx, dxy, dxs], [dxy, dyy, dys], [dxs, dys, dss]], F)
        sol = _solve3(hess, dd)
        if sol is None:
            sol = np.zeros(3, F)                        # Matx::solve leaves zeros when singular
        xi, xr, xc = F(-sol[2]), F(-sol[1]), F(-sol[0])
        if abs(xi) < 0.5 and abs(xr) < 0.5 and abs(xc) < 0.5:
            break
        big = F(2147483647 // 3)
        if abs(xi) > big or abs(xr) > big or abs(xc) > big:
            return None
        c += cv_round(xc)
        r += cv_round(xr)
        layer += cv_round(xi)
        if (layer < 1 or layer > N_LAYERS or c < IMG_BORDER or c >= cols - IMG_BORDER
                or r < IMG_BORDER or r >= rows - IMG_BORDER):
            return None
    img, prv, nxt = dog_oct[layer], dog_oct[layer - 1], dog_oct[layer + 1]
    dd = np.array([(img[r, c + 1] - img[r, c - 1]) * deriv_scale,
                   (img[r + 1, c] - img[r - 1, c]) * deriv_scale,
                   (nxt[r, c] - prv[r, c]) * deriv_scale], F)
    t = F(dd[0] * xc + dd[1] * xr + dd[2] * xi)
    contr = F(img[r, c] * img_scale + t * F(0.5))
    if abs(contr) * N_LAYERS < CONTRAST_THR:
        return None
    v2 = F(img[r, c] * 2)
    dxx = F((img[r, c + 1] + img[r, c - 1] - v2) * second_deriv_scale)
    dyy = F((img[r + 1, c] + img[r - 1, c] - v2) * second_deriv_scale)
    dxy = F((img[r + 1, c + 1] - img[r + 1, c - 1] - img[r - 1, c + 1] + img[r - 1, c - 1])
            * cross_deriv_scale)
    tr = F(dxx + dyy)
    det = F(dxx * dyy - dxy * dxy)
    if det <= 0 or tr * tr * F(EDGE_THR) >= F((EDGE_THR + 1) * (EDGE_THR + 1)) * det:
        return None
    scale = float(1 << octv)
    return dict(x=F((c + xc) * scale), y=F((r + xr) * scale),
                octave=octv + (layer << 8) + (cv_round((xi + F(0.5)) * 255) << 16),
                size=F(SIGMA * np.power(F(2.0), F((layer + xi) / N_LAYERS)) * scale * 2),
                response=F(abs(contr)), r=r, c=c, layer=layer, octv=octv)


def orientation_hist(img, c, r, radius, sigma):
    """calcOrientationHist -> (hist[36], max)."""
    n = ORI_HIST_BINS
    rows, cols = img.shape
    expf_scale = F(-1.0 / (2.0 * sigma * sigma))
    dxs, dys, ws = [], [], []
    for i in range(-radius, radius + 1):
        y = r + i
        if y <= 0 or y >= rows - 1:
            continue
        for j in range(-radius, radius + 1):
            x = c + j
            if x <= 0 or x >= cols - 1:
                continue
            dxs.append(F(img[y, x + 1] - img[y, x - 1]))
            dys.append(F(img[y - 1, x] - img[y + 1, x]))
            ws.append(F((i * i + j * j) * expf_scale))
    temp = np.zeros(n + 4, F)
    if dxs:
        dx, dy = np.array(dxs, F), np.array(dys, F)
        w = np.exp(np.array(ws, F)).astype(F)
        ori = fast_atan2(dy, dx)
        mag = np.sqrt(dx * dx + dy * dy).astype(F)
        for k in range(len(dx)):
            b = cv_round(F(n / 360.0) * ori[k])
            if b >= n:
                b -= n
            if b < 0:
                b += n
            temp[2 + b] = F(temp[2 + b] + w[k] * mag[k])
    temp[0], temp[1] = temp[n], temp[n + 1]
    temp[n + 2], temp[n + 3] = temp[2], temp[3]
    hist = np.zeros(n, F)
    for i in range(n):
        t = i + 2
        hist[i] = F(F((temp[t - 2] + temp[t + 2]) * F(1.0 / 16.0))
                    + F((temp[t - 1] + temp[t + 1]) * F(4.0 / 16.0))
                    + F(temp[t] * F(6.0 / 16.0)))
    return hist, F(hist.max())


def find_keypoints(gauss, dog):
    """findScaleSpaceExtrema over a pyramid (lists over octaves of lists of planes).
    Returns keypoints in detection order, coordinates of the doubled base image."""
    threshold = int(np.floor(0.5 * CONTRAST_THR / N_LAYERS * 255))
    n = ORI_HIST_BINS
    out = []
    for o, dog_oct in enumerate(dog):
        rows, cols = dog_oct[0].shape
        if rows <= 2 * IMG_BORDER or cols <= 2 * IMG_BORDER:
            continue
        stack = np.stack(dog_oct)                                   # [5][rows][cols]
        for i in range(1, N_LAYERS + 1):
            cur = stack[i, IMG_BORDER:rows - IMG_BORDER, IMG_BORDER:cols - IMG_BORDER]
            neigh = [stack[i + dl, IMG_BORDER + dy:rows - IMG_BORDER + dy,
                           IMG_BORDER + dx:cols - IMG_BORDER + dx]
                     for dl in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
            hi, lo = np.maximum.reduce(neigh), np.minimum.reduce(neigh)
            cand = (np.abs(cur) > threshold) & (((cur > 0) & (cur >= hi)) | ((cur < 0) & (cur <= lo)))
            for rr, cc in zip(*np.nonzero(cand)):
                kp = adjust_local_extrema(dog_oct, o, i, int(rr) + IMG_BORDER, int(cc) + IMG_BORDER)
                if kp is None:
                    continue
                scl = F(kp["size"] * F(0.5) / (1 << o))
                hist, omax = orientation_hist(gauss[o][kp["layer"]], kp["c"], kp["r"],
                                              cv_round(ORI_RADIUS * scl), F(ORI_SIG_FCTR * scl))
                mag_thr = F(omax * F(ORI_PEAK_RATIO))
                for j in range(n):
                    left, right = (j - 1) % n, (j + 1) % n
                    if hist[j] > hist[left] and hist[j] > hist[right] and hist[j] >= mag_thr:
                        b = F(j + F(0.5) * (hist[left] - hist[right])
                              / (hist[left] - 2 * hist[j] + hist[right]))
                        b = b + n if b < 0 else (b - n if b >= n else b)
                        angle = F(360.0 - F(360.0 / n) * b)
                        if abs(angle - 360.0) < FLT_EPSILON:
                            angle = F(0.0)
                        out.append(dict(kp, angle=angle))
    return out


def sort_unique(kps):
    """KeyPointsFilter::removeDuplicatedSorted."""
    key = lambda k: (k["x"], k["y"], -k["size"], k["angle"], -k["response"], -k["octave"])  # noqa: E731
    kps = sorted(kps, key=key)
    out = []
    for k in kps:
        if out and (out[-1]["x"], out[-1]["y"], out[-1]["size"], out[-1]["angle"]) == \
                (k["x"], k["y"], k["size"], k["angle"]):
            continue
        out.append(k)
    return out


def unpack_octave(packed):
    octave, layer = packed & 255, (packed >> 8) & 255
    if octave >= 128:
        octave |= -128
    scale = 1.0 / (1 << octave) if octave >= 0 else float(1 << -octave)
    return octave, layer, scale


def descriptor(img, ptx, pty, ori, scl):
    """calcSIFTDescriptor(img, ptf, ori, scl, d=4, n=8) -> float32[128]."""
    d, n = DESCR_WIDTH, DESCR_HIST_BINS
    px, py = cv_round(ptx), cv_round(pty)
    cos_t = F(np.cos(F(ori * F(np.pi / 180))))
    sin_t = F(np.sin(F(ori * F(np.pi / 180))))
    bins_per_rad = F(n / 360.0)
    exp_scale = F(-1.0 / (d * d * 0.5))
    hist_width = F(DESCR_SCL_FCTR * scl)
    radius = cv_round(hist_width * F(1.4142135623730951) * (d + 1) * F(0.5))
    rows, cols = img.shape
    radius = min(radius, int(np.sqrt(float(cols) * cols + float(rows) * rows)))
    cos_t, sin_t = F(cos_t / hist_width), F(sin_t / hist_width)
    hist = np.zeros((d + 2, d + 2, n + 2), F)
    xs, ys, rb, cb, ws = [], [], [], [], []
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            c_rot = F(F(j * cos_t) - F(i * sin_t))
            r_rot = F(F(j * sin_t) + F(i * cos_t))
            rbin = F(r_rot + d // 2 - F(0.5))
            cbin = F(c_rot + d // 2 - F(0.5))
            r, c = py + i, px + j
            if -1 < rbin < d and -1 < cbin < d and 0 < r < rows - 1 and 0 < c < cols - 1:
                xs.append(F(img[r, c + 1] - img[r, c - 1]))
                ys.append(F(img[r - 1, c] - img[r + 1, c]))
                rb.append(rbin)
                cb.append(cbin)
                ws.append(F(F(c_rot * c_rot + r_rot * r_rot) * exp_scale))
    if xs:
        x, y = np.array(xs, F), np.array(ys, F)
        oris = fast_atan2(y, x)
        mags = (np.sqrt(x * x + y * y).astype(F) * np.exp(np.array(ws, F)).astype(F)).astype(F)
        for k in range(len(xs)):
            rbin, cbin = rb[k], cb[k]
            obin = F(F(oris[k] - ori) * bins_per_rad)
            mag = mags[k]
            r0, c0, o0 = int(np.floor(rbin)), int(np.floor(cbin)), int(np.floor(obin))
            rbin, cbin, obin = F(rbin - r0), F(cbin - c0), F(obin - o0)
            if o0 < 0:
                o0 += n
            if o0 >= n:
                o0 -= n
            v_r1 = F(mag * rbin)
            v_r0 = F(mag - v_r1)
            v_rc11 = F(v_r1 * cbin)
            v_rc10 = F(v_r1 - v_rc11)
            v_rc01 = F(v_r0 * cbin)
            v_rc00 = F(v_r0 - v_rc01)
            v111 = F(v_rc11 * obin)
            v110 = F(v_rc11 - v111)
            v101 = F(v_rc10 * obin)
            v100 = F(v_rc10 - v101)
            v011 = F(v_rc01 * obin)
            v010 = F(v_rc01 - v011)
            v001 = F(v_rc00 * obin)
            v000 = F(v_rc00 - v001)
            a, b = r0 + 1, c0 + 1
            hist[a, b, o0] += v000
            hist[a, b, o0 + 1] += v001
            hist[a, b + 1, o0] += v010
            hist[a, b + 1, o0 + 1] += v011
            hist[a + 1, b, o0] += v100
            hist[a + 1, b, o0 + 1] += v101
            hist[a + 1, b + 1, o0] += v110
            hist[a + 1, b + 1, o0 + 1] += v111
    dst = np.zeros((d, d, n), F)
    for i in range(d):
        for j in range(d):
            cell = hist[i + 1, j + 1].copy()
            cell[0] = F(cell[0] + cell[n])
            cell[1] = F(cell[1] + cell[n + 1])
            dst[i, j] = cell[:n]
    dst = dst.reshape(-1)
    nrm2 = F(0)
    for v in dst:
        nrm2 = F(nrm2 + v * v)
    thr = F(np.sqrt(nrm2) * F(DESCR_MAG_THR))
    dst = np.minimum(dst, thr)
    nrm2 = F(0)
    for v in dst:
        nrm2 = F(nrm2 + v * v)
    scale = F(INT_DESCR_FCTR / max(float(np.sqrt(nrm2)), float(FLT_EPSILON)))
    return np.clip(np.rint(dst * scale), 0, 255).astype(F)      # saturate_cast<uchar>, kept as float


def detect_and_compute(gauss, dog):
    """SIFT::detectAndCompute on a prebuilt pyramid (first octave -1).  Returns
    (keypoints as dicts with x, y, size, angle, response, octave; descriptors [K][128])."""
    kps = sort_unique(find_keypoints(gauss, dog))
    first_octave = -1
    final = []
    for k in kps:
        k = dict(k)
        k["octave"] = (k["octave"] & ~255) | ((k["octave"] + first_octave) & 255)
        k["x"], k["y"], k["size"] = F(k["x"] * F(0.5)), F(k["y"] * F(0.5)), F(k["size"] * F(0.5))
        final.append(k)
    desc = np.zeros((len(final), DESCR_WIDTH * DESCR_WIDTH * DESCR_HIST_BINS), F)
    for idx, k in enumerate(final):
        octave, layer, scale = unpack_octave(k["octave"])
        size = F(k["size"] * scale)
        img = gauss[octave - first_octave][layer]
        angle = F(360.0 - k["angle"])
        if abs(angle - 360.0) < FLT_EPSILON:
            angle = F(0.0)
        desc[idx] = descriptor(img, F(k["x"] * scale), F(k["y"] * scale), angle, F(size * F(0.5)))
    return final, desc


def root_sift(des):
    """features.py:198."""
    return np.sqrt(des / (des.sum(axis=1, keepdims=True) + 1e-7))
