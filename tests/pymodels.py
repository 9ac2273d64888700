"""Independent, definition-level Python models used to cross-check the C oracle.

They are deliberately written from the *definitions* (brute force, python lists), not from the
oracle's C code, so that an error in the oracle's optimised formulation shows up as a mismatch.
Sized for small inputs only.
"""
import numpy as np

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def fast_is_corner(img, x, y, t):
    """FAST-9/16 by definition: >= 9 contiguous ring pixels all brighter than p+t or all darker
    than p-t (cv::FAST TYPE_9_16; called from src/ORBextractor.cc:811,816)."""
    v = int(img[y, x])
    q = [int(img[y + dy, x + dx]) for dx, dy in RING]
    for sign in (1, -1):
        flags = [(sign * (qq - v)) > t for qq in q]
        for s in range(16):
            if all(flags[(s + j) % 16] for j in range(9)):
                return True
    return False


def fast_score(img, x, y):
    """Score by definition: the largest t for which the pixel is still a corner (-1 if never)."""
    best = -1
    for t in range(0, 256):
        if fast_is_corner(img, x, y, t):
            best = t
        else:
            break
    return best


def fast9_16(img, th):
    """Whole cv::FAST(img, th, nonmax=True) on a small image, raster order [(x, y, score)]."""
    h, w = img.shape
    score = np.zeros((h, w), np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if fast_is_corner(img, x, y, th):
                score[y, x] = fast_score(img, x, y)
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s == 0 and not fast_is_corner(img, x, y, th):
                continue
            nb = [score[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx or dy)]
            if all(s > n for n in nb):
                out.append((x, y, int(s)))
    return out


def level_candidates(img, ini=20, mn=7):
    """Cell loop of ComputeKeyPointsOctTree (src/ORBextractor.cc:767-831), python ints."""
    import math
    h, w = img.shape
    minB = 16
    maxBX, maxBY = w - 16, h - 16
    width, height = np.float32(maxBX - minB), np.float32(maxBY - minB)
    nCols, nRows = int(width / np.float32(30)), int(height / np.float32(30))
    wCell = int(math.ceil(width / np.float32(nCols)))
    hCell = int(math.ceil(height / np.float32(nRows)))
    out = []
    for i in range(nRows):
        iniY = minB + i * hCell
        maxY = iniY + hCell + 6
        if iniY >= maxBY - 3:
            continue
        maxY = min(maxY, maxBY)
        for j in range(nCols):
            iniX = minB + j * wCell
            maxX = iniX + wCell + 6
            if iniX >= maxBX - 6:
                continue
            maxX = min(maxX, maxBX)
            sub = img[iniY:maxY, iniX:maxX]
            k = fast9_16(sub, ini)
            if not k:
                k = fast9_16(sub, mn)
            out += [(x + j * wCell, y + i * hCell, s) for x, y, s in k]
    return out


class _Node:
    __slots__ = ("ulx", "uly", "brx", "bry", "keys", "nomore", "seq")


def distribute_octtree(cands, width, height, N):
    """DistributeOctTree (src/ORBextractor.cc:541-765) with a python list as std::list.
    cands: list of (x, y, score).  Pointer tie-break -> creation sequence (later = greater).
    Returns indices into cands in result order."""
    import math
    f32 = np.float32
    nIni = int(math.floor(float(f32(width) / f32(height)) + 0.5))
    hX = f32(width) / f32(nIni)
    seq = [0]

    def new_node(ulx, uly, brx, bry):
        n = _Node()
        n.ulx, n.uly, n.brx, n.bry = ulx, uly, brx, bry
        n.keys = []
        n.nomore = False
        n.seq = seq[0]
        seq[0] += 1
        return n

    lst = []
    roots = []
    for i in range(nIni):
        n = new_node(int(hX * f32(i)), 0, int(hX * f32(i + 1)), height)
        lst.append(n)
        roots.append(n)
    for idx, (x, y, s) in enumerate(cands):
        roots[int(f32(x) / hX)].keys.append(idx)
    keep = []
    for n in lst:
        if len(n.keys) == 1:
            n.nomore = True
            keep.append(n)
        elif len(n.keys) > 1:
            keep.append(n)
    lst = keep

    def divide(n):
        halfX = int(math.ceil(float(f32(n.brx - n.ulx) / f32(2))))
        halfY = int(math.ceil(float(f32(n.bry - n.uly) / f32(2))))
        mx, my = n.ulx + halfX, n.uly + halfY
        ch = [new_node(n.ulx, n.uly, mx, my), new_node(mx, n.uly, n.brx, my),
              new_node(n.ulx, my, mx, n.bry), new_node(mx, my, n.brx, n.bry)]
        for idx in n.keys:
            x, y, _ = cands[idx]
            if x < mx:
                k = 0 if y < my else 2
            else:
                k = 1 if y < my else 3
            ch[k].keys.append(idx)
        for c in ch:
            if len(c.keys) == 1:
                c.nomore = True
        return ch

    finish = False
    while not finish:
        prev_size = len(lst)
        n_to_expand = 0
        vsz = []
        i = 0
        # iterate front->back over the nodes present at the start of the pass
        snapshot = list(lst)
        for n in snapshot:
            if n.nomore:
                continue
            for c in divide(n):
                if c.keys:
                    lst.insert(0, c)
                    if len(c.keys) > 1:
                        n_to_expand += 1
                        vsz.append(c)
            lst.remove(n)
        if len(lst) >= N or len(lst) == prev_size:
            finish = True
        elif len(lst) + n_to_expand * 3 > N:
            while not finish:
                prev_size = len(lst)
                prev = sorted(vsz, key=lambda c: (len(c.keys), c.seq))
                vsz = []
                for n in reversed(prev):
                    for c in divide(n):
                        if c.keys:
                            lst.insert(0, c)
                            if len(c.keys) > 1:
                                vsz.append(c)
                    lst.remove(n)
                    if len(lst) >= N:
                        break
                if len(lst) >= N or len(lst) == prev_size:
                    finish = True
    out = []
    for n in lst:
        best = n.keys[0]
        for k in n.keys[1:]:
            if cands[k][2] > cands[best][2]:
                best = k
        out.append(best)
    return out


def hamming(a, b):
    return sum(int(x ^ y).bit_count() for x, y in zip(bytes(a), bytes(b)))


def search_by_bow(desc1, valid1, angle1, fv1, desc2, valid2, angle2, fv2, th, strict, ratio,
                  check_ori):
    """SearchByBoW (src/ORBmatcher.cc:159-288 / :522-655) on dict FeatureVectors
    {node: [feature indices]} (std::map iteration = ascending node id)."""
    n1, n2 = len(desc1), len(desc2)
    m12 = [-1] * n1
    m21 = [-1] * n2
    hist = [[] for _ in range(30)]
    nm = 0
    for node in sorted(set(fv1) & set(fv2)):
        for i1 in fv1[node]:
            if not valid1[i1]:
                continue
            b1, b2, bi = 256, 256, -1
            for i2 in fv2[node]:
                if m21[i2] >= 0:
                    continue
                if valid2 is not None and not valid2[i2]:
                    continue
                d = hamming(desc1[i1], desc2[i2])
                if d < b1:
                    b2, b1, bi = b1, d, i2
                elif d < b2:
                    b2 = d
            ok = (b1 < th) if strict else (b1 <= th)
            if ok and np.float32(b1) < np.float32(ratio) * np.float32(b2):
                m12[i1] = bi
                m21[bi] = i1
                if check_ori:
                    rot = np.float32(angle1[i1]) - np.float32(angle2[bi])
                    if rot < 0:
                        rot = np.float32(rot + np.float32(360.0))
                    v = float(np.float32(rot * (np.float32(1.0) / np.float32(30))))
                    import math
                    b = int(math.floor(v + 0.5)) if v >= 0 else -int(math.floor(-v + 0.5))
                    if b == 30:
                        b = 0
                    hist[b].append(i1)
                nm += 1
    if check_ori:
        sizes = [len(h) for h in hist]
        m1 = m2 = m3 = 0
        i1 = i2 = i3 = -1
        for i, s in enumerate(sizes):
            if s > m1:
                m3, m2, m1 = m2, m1, s
                i3, i2, i1 = i2, i1, i
            elif s > m2:
                m3, m2 = m2, s
                i3, i2 = i2, i
            elif s > m3:
                m3, i3 = s, i
        if np.float32(m2) < np.float32(0.1) * np.float32(m1):
            i2 = i3 = -1
        elif np.float32(m3) < np.float32(0.1) * np.float32(m1):
            i3 = -1
        for i in range(30):
            if i in (i1, i2, i3):
                continue
            for a in hist[i]:
                m21[m12[a]] = -1
                m12[a] = -1
                nm -= 1
    return nm, m12, m21
