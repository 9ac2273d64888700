"""MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:283-349), batched over map points: the oracle against a
definition-level numpy model on CPU; the HIP path against the oracle on the GPU."""
import numpy as np
import pytest


def _lists(rng, P, nmax, entropy=None):
    """P descriptor lists of 0..nmax rows: noisy copies of one descriptor per point (as observations of one point are)."""
    n = rng.integers(0, nmax + 1, P)
    n[rng.integers(0, P, max(1, P // 20))] = 0
    n[rng.integers(0, P, max(1, P // 20))] = 1
    off = np.concatenate([[0], np.cumsum(n)]).astype(np.int32)
    desc = np.zeros((off[-1], 32), np.uint8)
    for p in range(P):
        base = rng.integers(0, 256, 32, dtype=np.uint8)
        for r in range(off[p], off[p + 1]):
            d = base.copy()
            for b in rng.integers(0, 256, rng.integers(0, 3 if entropy == "low" else 40)):
                d[b >> 3] ^= 1 << (b & 7)
            desc[r] = d
    return desc, off


def _model(desc, off):
    bits = np.unpackbits(desc, axis=1).astype(np.int32)
    best, med = [], []
    for p in range(len(off) - 1):
        b = bits[off[p]:off[p + 1]]
        N = len(b)
        if N == 0:
            best.append(-1)
            med.append(2**31 - 1)
            continue
        dist = np.abs(b[:, None, :] - b[None, :, :]).sum(2)
        m = np.sort(dist, axis=1)[:, int(0.5 * (N - 1))]
        best.append(int(np.argmin(m)))                        # argmin returns the first minimum
        med.append(int(m.min()))
    return np.array(best, np.int32), np.array(med, np.int32)


@pytest.mark.parametrize("entropy", [None, "low"])
def test_oracle_matches_numpy_model(oracle, entropy):
    desc, off = _lists(np.random.default_rng(3), 300, 30, entropy)
    best, med = oracle.distinctive_descriptors(desc, off)
    rb, rm = _model(desc, off)
    assert np.array_equal(best, rb) and np.array_equal(med, rm)
    assert (best >= 0).sum() > 250 and len(set(best.tolist())) > 10


def test_oracle_median_index_and_first_wins(oracle):
    """N = 4: the median is sorted_row[1] (0.5 * 3 truncated), i.e. the nearest other descriptor; equal medians keep the
    first row."""
    d = np.zeros((4, 32), np.uint8)
    d[1, 0], d[2, 0], d[3, 0] = 0x01, 0x03, 0xFF              # distances to row 0: 1, 2, 8
    best, med = oracle.distinctive_descriptors(d, [0, 4])
    assert best[0] == 0 and med[0] == 1                       # rows 0, 1 and 2 all have a neighbour at distance 1
    best, med = oracle.distinctive_descriptors(d[[3, 2, 1, 0]], [0, 4])
    assert best[0] == 1 and med[0] == 1
    best, med = oracle.distinctive_descriptors(d, [0, 0, 1, 4])
    assert best.tolist() == [-1, 0, 0] and med[1] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("P,nmax,entropy", [(2000, 24, None), (500, 150, None), (3000, 12, "low"), (40, 700, "low")])
def test_hip_distinctive_descriptors_match_oracle(oracle, P, nmax, entropy):
    from orbhip import guided
    from orbhip.extractor import ORBextractor
    ex = ORBextractor(500, max_w=320, max_h=240)
    desc, off = _lists(np.random.default_rng(P + nmax), P, nmax, entropy)
    best, med = guided.ComputeDistinctiveDescriptors(ex, desc, off)
    rb, rm = oracle.distinctive_descriptors(desc, off)
    assert np.array_equal(best, rb) and np.array_equal(med, rm)
    best, med = guided.ComputeDistinctiveDescriptors(ex, desc[:0], [0, 0, 0])
    assert best.tolist() == [-1, -1]
    from orbhip.capi import OrbHipError
    with pytest.raises(OrbHipError):
        guided.ComputeDistinctiveDescriptors(ex, desc, [0, 5, 3])
    with pytest.raises(OrbHipError):
        guided.ComputeDistinctiveDescriptors(ex, desc, [1, 5])
    ex.close()
