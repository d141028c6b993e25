"""numpy prototype: PLOC (radius 2) over references with optional pre-splitting; reports the BVH2 SAH sum (inner-node areas / root area)
and, through a C++ hook, nothing else.  Only a quality proxy to decide whether device-side pre-splitting is worth building."""
import sys, time, importlib
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
hrt = importlib.import_module("nvidia-optix-ray-tracer_amd")


def half_area(lo, hi):
    d = hi - lo
    return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]


def morton(c, lo, hi):
    q = np.clip(((c - lo) / (hi - lo) * 2097151.0), 0, 2097151).astype(np.uint64)
    def spread(x):
        x = x & np.uint64(0x1fffff)
        x = (x | (x << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        x = (x | (x << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        x = (x | (x << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        x = (x | (x << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        x = (x | (x << np.uint64(2))) & np.uint64(0x1249249249249249)
        return x
    return (spread(q[:, 0]) << np.uint64(2)) | (spread(q[:, 1]) << np.uint64(1)) | spread(q[:, 2])


def ploc(lo, hi, radius=2):
    """returns the sum of the half areas of all inner nodes created (excluding nothing), and the root box"""
    cen = 0.5 * (lo + hi)
    order = np.argsort(morton(cen, cen.min(0), cen.max(0)), kind="stable")
    lo, hi = lo[order].copy(), hi[order].copy()
    total = 0.0
    rounds = 0
    while len(lo) > 1:
        m = len(lo)
        best = np.full(m, np.inf); nn = np.full(m, -1, np.int64)
        idx = np.arange(m)
        for d in list(range(-radius, 0)) + list(range(1, radius + 1)):
            j = idx + d
            ok = (j >= 0) & (j < m)
            jj = np.clip(j, 0, m - 1)
            a = half_area(np.minimum(lo, lo[jj]), np.maximum(hi, hi[jj]))
            a = np.where(ok, a, np.inf)
            better = a < best
            best = np.where(better, a, best); nn = np.where(better, jj, nn)
        mutual = nn[nn] == idx
        starts = mutual & (idx < nn)
        ends = mutual & (idx > nn)
        keep = ~ends
        nlo, nhi = lo.copy(), hi.copy()
        s = np.nonzero(starts)[0]
        nlo[s] = np.minimum(lo[s], lo[nn[s]]); nhi[s] = np.maximum(hi[s], hi[nn[s]])
        total += half_area(nlo[s], nhi[s]).sum()
        lo, hi = nlo[keep], nhi[keep]
        rounds += 1
    return total, half_area(lo[0], hi[0]), rounds


def clip_boxes(v, blo, bhi, axis, pos):
    """triangles v (n,3,3) restricted to boxes [blo,bhi]; split by plane x[axis] = pos: conservative-tight bounds of both halves"""
    n = len(v)
    out = []
    for side in (0, 1):
        lo = np.full((n, 3), np.inf); hi = np.full((n, 3), -np.inf)
        x = v[:, :, axis]
        inside = (x <= pos[:, None]) if side == 0 else (x >= pos[:, None])
        for k in range(3):
            m = inside[:, k]
            lo[m] = np.minimum(lo[m], v[m, k]); hi[m] = np.maximum(hi[m], v[m, k])
        for a, b in ((0, 1), (1, 2), (2, 0)):
            xa, xb = x[:, a], x[:, b]
            cross = ((xa < pos) & (xb > pos)) | ((xa > pos) & (xb < pos))
            t = np.where(cross, (pos - xa) / np.where(cross, xb - xa, 1.0), 0.0)
            p = v[:, a] + t[:, None] * (v[:, b] - v[:, a])
            p[:, axis] = pos
            lo[cross] = np.minimum(lo[cross], p[cross]); hi[cross] = np.maximum(hi[cross], p[cross])
        lo = np.maximum(lo, blo); hi = np.minimum(hi, bhi)
        if side == 0:
            hi[:, axis] = np.minimum(hi[:, axis], pos)
        else:
            lo[:, axis] = np.maximum(lo[:, axis], pos)
        out.append((lo, hi))
    return out


def presplit(verts, beta, scene_lo, scene_hi):
    """Karras-Aila style: priority = (2^-level * (A_box - A_ideal))^(1/3), splits ~ priority, each split at the most important
    Morton plane crossing the reference's box.  Returns (lo, hi) of all references."""
    v = verts.astype(np.float64)
    lo, hi = v.min(1), v.max(1)
    ext = (scene_hi - scene_lo)

    def important_plane(lo, hi):
        # per axis: highest level (coarsest) grid plane k/2^L crossing (lo,hi): find smallest L such that floor(lo*2^L) != floor(hi*2^L)
        a = (lo - scene_lo) / ext; b = (hi - scene_lo) / ext
        level = np.full(lo.shape, 30, np.int64); pos = np.zeros(lo.shape)
        found = np.zeros(lo.shape, bool)
        for L in range(1, 22):
            fa = np.floor(a * (1 << L)); fb = np.floor(b * (1 << L))
            cross = (fa != fb) & ~found & (b > a)
            # plane = ceil-ish: the grid line just above a's cell
            p = (fa + 1) / (1 << L)
            level = np.where(cross, L, level); pos = np.where(cross, p, pos); found |= cross
        ax = np.argmin(level, axis=1)
        n = len(lo); r = np.arange(n)
        return ax, level[r, ax], scene_lo[ax] + pos[r, ax] * ext[ax]

    ax, lvl, pos = important_plane(lo, hi)
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    cr = np.cross(e1, e2)
    a_ideal = np.abs(cr).sum(1) * 0.5            # sum of projected areas = half-area of the "ideal" flat box
    a_box = half_area(lo, hi)
    pr = np.cbrt(np.maximum(a_box - a_ideal, 0) * 2.0 ** (-lvl))
    budget = beta * len(v)
    D = budget / pr.sum()
    s = np.floor(D * pr).astype(np.int64)
    tri = np.arange(len(v))
    out_lo, out_hi = [], []
    cur_tri, cur_lo, cur_hi, cur_s = tri, lo, hi, s
    while len(cur_tri):
        done = cur_s <= 0
        out_lo.append(cur_lo[done]); out_hi.append(cur_hi[done])
        t, l, h, ss = cur_tri[~done], cur_lo[~done], cur_hi[~done], cur_s[~done]
        if not len(t):
            break
        ax, lvl, pos = important_plane(l, h)
        (llo, lhi), (rlo, rhi) = clip_boxes(v[t], l, h, 0, pos) if False else (None, None), (None, None)
        # per-axis clipping needs grouping by axis
        nl_lo = np.empty_like(l); nl_hi = np.empty_like(l); nr_lo = np.empty_like(l); nr_hi = np.empty_like(l)
        for a in range(3):
            m = ax == a
            if m.any():
                (a0, a1), (b0, b1) = clip_boxes(v[t[m]], l[m], h[m], a, pos[m])
                nl_lo[m], nl_hi[m], nr_lo[m], nr_hi[m] = a0, a1, b0, b1
        rem = ss - 1
        wl = (nl_hi - nl_lo).max(1); wr = (nr_hi - nr_lo).max(1)
        sl = np.floor(rem * wl / np.maximum(wl + wr, 1e-30) + 0.5).astype(np.int64); sr = rem - sl
        okl = (nl_hi >= nl_lo).all(1); okr = (nr_hi >= nr_lo).all(1)
        cur_tri = np.concatenate([t[okl], t[okr]]); cur_lo = np.concatenate([nl_lo[okl], nr_lo[okr]]); cur_hi = np.concatenate([nl_hi[okl], nr_hi[okr]])
        cur_s = np.concatenate([sl[okl], sr[okr]])
    return np.concatenate(out_lo), np.concatenate(out_hi)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    scene = hrt.scenes.soup_1m(64, 64, 1) if n == 1_000_000 else hrt.scenes.random_soup(n, hrt.scenes.soup_law_edge(n), 1, 64, 64, 1)
    v = scene["instances"][0]["vertices"]
    lo, hi = v.min(1).astype(np.float64), v.max(1).astype(np.float64)
    slo, shi = lo.min(0), hi.max(0)
    t0 = time.time(); tot, root, rounds = ploc(lo, hi, 2); print("PLOC r=2 no split: refs %d inner-area sum / root %.2f leaf-area sum / root %.2f (%d rounds, %.1fs)" % (len(lo), tot / root, half_area(lo, hi).sum() / root, rounds, time.time() - t0))
    for beta in (0.2, 0.5, 1.0):
        t0 = time.time(); rlo, rhi = presplit(v, beta, slo, shi)
        tot, root, rounds = ploc(rlo, rhi, 2)
        print("PLOC r=2 presplit beta %.1f: refs %d inner-area sum / root %.2f leaf-area sum / root %.2f (%.1fs)" % (beta, len(rlo), tot / root, half_area(rlo, rhi).sum() / root, time.time() - t0))


def grid_cells_split(verts, G, slo, shi):
    """every triangle cut at the planes of a uniform G^3 grid over the scene box: references (clipped boxes) that lie in one cell each"""
    v = verts.astype(np.float64)
    lo, hi = v.min(1), v.max(1)
    tri = np.arange(len(v))
    ext = shi - slo
    for a in range(3):
        step = ext[a] / G
        for _ in range(3):                     # (a triangle of this soup spans at most two cells per axis)
            ca = np.floor((lo[:, a] - slo[a]) / step + 1e-9).clip(0, G - 1); cb = np.floor((hi[:, a] - slo[a]) / step - 1e-9).clip(0, G - 1)
            cross = cb > ca
            if not cross.any():
                break
            pos = slo[a] + (ca[cross] + 1) * step
            (l0, l1), (r0, r1) = clip_boxes(v[tri[cross]], lo[cross], hi[cross], a, pos)
            r0[:, a] = np.maximum(r0[:, a], pos + 1e-12)
            keep = ~cross
            tri = np.concatenate([tri[keep], tri[cross], tri[cross]]); lo = np.concatenate([lo[keep], l0, r0]); hi = np.concatenate([hi[keep], l1, r1])
            ok = (hi >= lo).all(1)
            tri, lo, hi = tri[ok], lo[ok], hi[ok]
    cen = 0.5 * (lo + hi)
    cell = np.floor((cen - slo) / (ext / G)).clip(0, G - 1).astype(np.int64)
    return lo, hi, cell[:, 0] * G * G + cell[:, 1] * G + cell[:, 2]


def ploc_in_cells(lo, hi, cell, radius=2):
    """PLOC inside every cell (a cluster only merges with clusters of its own cell), then PLOC over the cell roots"""
    cen = 0.5 * (lo + hi)
    key = morton(cen, cen.min(0), cen.max(0))
    order = np.lexsort((key, cell))
    lo, hi, cell = lo[order].copy(), hi[order].copy(), cell[order].copy()
    total = 0.0
    staged = True
    while len(lo) > 1:
        m = len(lo); idx = np.arange(m)
        best = np.full(m, np.inf); nn = np.full(m, -1, np.int64)
        for d in list(range(-radius, 0)) + list(range(1, radius + 1)):
            j = idx + d; ok = (j >= 0) & (j < m); jj = np.clip(j, 0, m - 1)
            if staged:
                ok &= cell[jj] == cell
            a = np.where(ok, half_area(np.minimum(lo, lo[jj]), np.maximum(hi, hi[jj])), np.inf)
            better = a < best; best = np.where(better, a, best); nn = np.where(better, jj, nn)
        valid = nn >= 0
        mutual = valid & (nn[np.clip(nn, 0, m - 1)] == idx)
        starts = mutual & (idx < nn); ends = mutual & (idx > nn)
        if not starts.any():
            if staged:
                staged = False; continue
            break
        s = np.nonzero(starts)[0]
        nlo, nhi = lo.copy(), hi.copy()
        nlo[s] = np.minimum(lo[s], lo[nn[s]]); nhi[s] = np.maximum(hi[s], hi[nn[s]])
        total += half_area(nlo[s], nhi[s]).sum()
        keep = ~ends
        lo, hi, cell = nlo[keep], nhi[keep], cell[keep]
    return total, half_area(lo[0], hi[0])


if __name__ == "__main__" and "--cells" in sys.argv:
    for G in (4, 8, 16, 32):
        t0 = time.time(); rlo, rhi, cell = grid_cells_split(v, G, slo, shi)
        tot, root = ploc_in_cells(rlo, rhi, cell, 2)
        print("grid %2d^3 cells + PLOC inside cells: refs %d inner-area sum / root %.2f leaf-area sum / root %.2f (%.1fs)" % (G, len(rlo), tot / root, half_area(rlo, rhi).sum() / root, time.time() - t0))
        tot2, root2 = ploc_in_cells(lo, hi, np.floor((0.5 * (lo + hi) - slo) / ((shi - slo) / G)).clip(0, G - 1).astype(np.int64) @ np.array([G * G, G, 1]), 2)
        print("          same cells by centroid, NO splitting: inner-area sum / root %.2f" % (tot2 / root2))
