"""ORACLE tooling -- seeded prediction / label cases for the accuracy-harness fixtures (test infrastructure only).

Used by oracle/make_golden.py (which runs the REFERENCE's metrics.py / utils.py on them and stores the values) and by
the tests (which run deepcharuco_amd/metrics.py on the regenerated inputs and compare with those values)."""
import numpy as np


def dc_case(seed: int, n: int = 4, hc: int = 6, wc: int = 8, n_ids: int = 16):
    """-> loc_logits (n,65,hc,wc) f32, ids_logits (n,n_ids+1,hc,wc) f32, loc_target (n,hc,wc) i64, ids_target (n,hc,wc) i64.
    Targets: 3..8 corners per frame with unique ids (frame n-1 has none).  Predictions: mostly the target with a clear
    margin, some shifted by one pixel, some far off (> 3 px), some missed, some false positives (new id, or a second
    detection of a target id in another cell)."""
    rng = np.random.default_rng([seed, 17])
    loc = (rng.standard_normal((n, 65, hc, wc), dtype=np.float32) * np.float32(0.5)).astype(np.float32)
    ids = (rng.standard_normal((n, n_ids + 1, hc, wc), dtype=np.float32) * np.float32(0.5)).astype(np.float32)
    ids[:, n_ids] += np.float32(6.0)                       # default: dust-bin everywhere
    loc_t = rng.integers(0, 64, (n, hc, wc)).astype(np.int64)
    ids_t = np.full((n, hc, wc), n_ids, np.int64)
    for b in range(n - 1):
        k = int(rng.integers(3, 9))
        cells = rng.choice(hc * wc, k + 2, replace=False)
        tid = rng.choice(n_ids, k, replace=False)
        for j in range(k):
            cy, cx = divmod(int(cells[j]), wc)
            ids_t[b, cy, cx] = tid[j]
            t_loc = int(loc_t[b, cy, cx])
            u = rng.random()
            if u < 0.10:
                continue                                    # missed: stays dust-bin
            p_loc = t_loc
            if u < 0.35:
                p_loc = t_loc + 1 if t_loc % 8 < 7 else t_loc - 1       # one pixel off
            elif u < 0.45:
                p_loc = (t_loc + 36) % 64                   # far off inside the cell
            loc[b, p_loc, cy, cx] += np.float32(8.0)
            ids[b, tid[j], cy, cx] += np.float32(14.0)
        # false positives: one with an id that is not a target, one repeating a target id elsewhere
        others = [i for i in range(n_ids) if i not in set(tid.tolist())]
        for extra, eid in ((k, others[0] if others else int(tid[0])), (k + 1, int(tid[0]))):
            cy, cx = divmod(int(cells[extra]), wc)
            loc[b, int(rng.integers(0, 64)), cy, cx] += np.float32(8.0)
            ids[b, eid, cy, cx] += np.float32(14.0)
    return loc, ids, loc_t, ids_t


def refinenet_case(seed: int, bs: int = 6):
    """-> heat (bs,1,64,64) f32 predictions, target (bs,64,64) f32 labels (peaks a few 1/8-px apart)."""
    rng = np.random.default_rng([seed, 23])
    heat = rng.standard_normal((bs, 1, 64, 64), dtype=np.float32)
    target = np.zeros((bs, 64, 64), np.float32)
    for b in range(bs):
        ty, tx = rng.integers(8, 56, 2)
        target[b, ty, tx] = 1.0
        dy, dx = rng.integers(-4, 5, 2)
        heat[b, 0, ty + dy, tx + dx] += np.float32(9.0)
    return heat, target


def pixel_error_case(seed: int, k: int = 12):
    """-> kpts_raw, kpts_ref, kpts_target (K,3) float64 [x, y, id] (raw = integer detector output on a subset of the
    targets, ref = refined to 1/8 px) and a second raw array containing an id that is not a target (-> (None, None))."""
    rng = np.random.default_rng([seed, 29])
    ids = rng.permutation(16)[:k].astype(np.float64)
    tgt = np.concatenate([rng.uniform(20, 300, (k, 2)), ids[:, None]], axis=1)
    keep = np.sort(rng.choice(k, k - 3, replace=False))
    raw = tgt[keep].copy()
    raw[:, :2] = np.rint(raw[:, :2] + rng.uniform(-2, 2, (len(keep), 2)))
    ref = raw.copy()
    ref[:, :2] = raw[:, :2] + np.rint(rng.uniform(-3, 3, (len(keep), 2)) * 8) / 8
    bad = raw.copy()
    bad[0, 2] = 99.0
    return raw, ref, tgt, bad
