"""numpy model of k_ssn_tree (laser_slam_amd/csrc/lsgpu_ssn_tree.hip.h): the box tree of SamplingSurfaceNormal built from
three presorted axes and stable partitions, step for step as the kernel does it -- lists, dense ranks, cur_pos, the
counting fix-up of tie runs -- so that the SCHEME can be checked on the CPU against the chain of stable sorts the
restatement defines (oracle/icp_oracle.c, lsgpu_host_filters.cpp).  Test infrastructure only."""
import numpy as np


def order_key(f):
    u = np.asarray(f, np.float32).view(np.uint32).copy()
    u[u == 0x80000000] = 0
    return np.where((u & 0x80000000) != 0, ~u, u | np.uint32(0x80000000)).astype(np.uint32)


def cut_axis(lo, hi):
    cut, ext = 0, np.float32(hi[0]) - np.float32(lo[0])
    if np.float32(hi[1]) - np.float32(lo[1]) > ext:
        ext, cut = np.float32(hi[1]) - np.float32(lo[1]), 1
    if np.float32(hi[2]) - np.float32(lo[2]) > ext:
        cut = 2
    return cut


def chain_of_stable_sorts(pts, knn, lo, hi):
    """The restatement: per segment a stable sort by the cut coordinate, split at count - count // 2."""
    leaves = []

    def rec(order, lo, hi):
        c = len(order)
        if c <= knn:
            leaves.append(order)
            return
        a = cut_axis(lo, hi)
        o2 = order[np.argsort(order_key(pts[order, a]), kind="stable")]
        left = c - c // 2
        cutval = pts[o2[left], a]
        hi2, lo2 = hi.copy(), lo.copy()
        hi2[a] = cutval
        lo2[a] = cutval
        rec(o2[:left], lo, hi2)
        rec(o2[left:], lo2, hi)

    rec(np.arange(len(pts)), lo.copy(), hi.copy())
    return leaves


def presorted_lists(pts, knn, lo, hi):
    """k_ssn_tree: returns the leaves (arrays of local ids, each in its current order)."""
    n = len(pts)
    keys = [order_key(pts[:, d]) for d in range(3)]
    lists = [np.argsort(keys[d], kind="stable") for d in range(3)]
    rank = []
    for d in range(3):
        ks = keys[d][lists[d]]
        r = np.empty(n, np.int64)
        r[lists[d]] = np.cumsum(np.concatenate([[0], (ks[1:] != ks[:-1]).astype(np.int64)]))
        rank.append(r)
    cur_pos = np.arange(n)
    segs = [dict(start=0, count=n, lo=lo.copy(), hi=hi.copy(), ord=-1)]
    while any(s["count"] > knn for s in segs):
        new = []
        for s in segs:
            st, c = s["start"], s["count"]
            if c <= knn:   # finished: carries over as child 2s, child 2s + 1 is empty
                new += [s, dict(start=st + c, count=0, lo=s["lo"], hi=s["hi"], ord=s["ord"])]
                continue
            a = cut_axis(s["lo"], s["hi"])
            la = lists[a]
            if s["ord"] != -1 and s["ord"] != a:   # step 1: tie runs of list[a] into the current order, by counting
                seg = la[st:st + c].copy()
                out = seg.copy()
                r = rank[a][seg]
                for i in range(c):
                    l, h = i, i + 1
                    while l > 0 and r[l - 1] == r[i]:
                        l -= 1
                    while h < c and r[h] == r[i]:
                        h += 1
                    if h - l > 1:
                        out[l + sum(1 for j in range(l, h) if cur_pos[seg[j]] < cur_pos[seg[i]])] = seg[i]
                la[st:st + c] = out
            cur_pos[la[st:st + c]] = np.arange(st, st + c)   # step 2
            left = c - c // 2
            for d in range(3):                               # step 3: stable partition of the other two lists
                if d != a:
                    seg = lists[d][st:st + c]
                    f = cur_pos[seg] - st >= left
                    lists[d][st:st + c] = np.concatenate([seg[~f], seg[f]])
            cutval = pts[la[st + left], a]
            hi2, lo2 = s["hi"].copy(), s["lo"].copy()
            hi2[a] = cutval
            lo2[a] = cutval
            new += [dict(start=st, count=left, lo=s["lo"], hi=hi2, ord=a), dict(start=st + left, count=c - left, lo=lo2, hi=s["hi"], ord=a)]
        segs = new
    leaves = []
    for s in segs:
        if s["count"]:
            st, c = s["start"], s["count"]
            leaves.append(np.arange(st, st + c) if s["ord"] == -1 else lists[s["ord"]][st:st + c].copy())
    return leaves


def select_then_tree(pts, knn, lo, hi, root_size):
    """Round 5, lsgpu_ssn_select.hip.h + k_ssn_tree: the UPPER levels keep no order at all.  A segment is a SET of points
    (kept in original-index order by stable partitions) with a signature -- the axes it was cut along, most recent first --
    and is halved at the median of the total order  (key on the cut axis, keys on the signature's other axes in order,
    original index),  which is exactly the order the chain of stable sorts would have put it in.  Once a segment holds
    <= root_size points it goes to the workgroup kernel: local ids in arrival (= original-index) order, three lists
    presorted by (key, local id); the list of the signature's first axis is put into the segment's current order by the
    same comparator (one fix-up of its tie runs), cur_pos is taken from it, and the levels run as in presorted_lists."""
    n = len(pts)
    keys = [order_key(pts[:, d]) for d in range(3)]
    leaves = []

    def tuple_of(e, a, sig):
        others = [x for x in sig if x != a]
        return (int(keys[a][e]),) + tuple(int(keys[x][e]) for x in others) + (int(e),)

    def tree(ids, lo, hi, sig):
        """k_ssn_tree on one root: ids in original-index order."""
        m = len(ids)
        k = [keys[d][ids] for d in range(3)]
        lists = [np.argsort(k[d], kind="stable") for d in range(3)]          # local ids, ties by local id
        rank = []
        for d in range(3):
            ks = k[d][lists[d]]
            r = np.empty(m, np.int64)
            r[lists[d]] = np.cumsum(np.concatenate([[0], (ks[1:] != ks[:-1]).astype(np.int64)]))
            rank.append(r)
        cur_pos = np.arange(m)
        root_ord = -1
        if sig:   # the initial fix-up: list[sig[0]] into the root's current order
            o1 = sig[0]
            others = [x for x in sig if x != o1]
            l1 = lists[o1]
            out = l1.copy()
            r = rank[o1][l1]
            ck = lambda e: tuple(int(rank[x][e]) for x in others) + (int(e),)
            for i in range(m):
                a0, b0 = i, i + 1
                while a0 > 0 and r[a0 - 1] == r[i]:
                    a0 -= 1
                while b0 < m and r[b0] == r[i]:
                    b0 += 1
                if b0 - a0 > 1:
                    out[a0 + sum(1 for j in range(a0, b0) if ck(l1[j]) < ck(l1[i]))] = l1[i]
            lists[o1] = out
            cur_pos[out] = np.arange(m)
            root_ord = o1
        segs = [dict(start=0, count=m, lo=lo.copy(), hi=hi.copy(), ord=root_ord)]
        while any(s["count"] > knn for s in segs):
            new = []
            for s in segs:
                st, c = s["start"], s["count"]
                if c <= knn:
                    new += [s, dict(start=st + c, count=0, lo=s["lo"], hi=s["hi"], ord=s["ord"])]
                    continue
                a = cut_axis(s["lo"], s["hi"])
                la = lists[a]
                if s["ord"] != -1 and s["ord"] != a:
                    seg = la[st:st + c].copy()
                    out = seg.copy()
                    r = rank[a][seg]
                    for i in range(c):
                        a0, b0 = i, i + 1
                        while a0 > 0 and r[a0 - 1] == r[i]:
                            a0 -= 1
                        while b0 < c and r[b0] == r[i]:
                            b0 += 1
                        if b0 - a0 > 1:
                            out[a0 + sum(1 for j in range(a0, b0) if cur_pos[seg[j]] < cur_pos[seg[i]])] = seg[i]
                    la[st:st + c] = out
                cur_pos[la[st:st + c]] = np.arange(st, st + c)
                left = c - c // 2
                for d in range(3):
                    if d != a:
                        seg = lists[d][st:st + c]
                        f = cur_pos[seg] - st >= left
                        lists[d][st:st + c] = np.concatenate([seg[~f], seg[f]])
                cutval = pts[ids[la[st + left]], a]
                hi2, lo2 = s["hi"].copy(), s["lo"].copy()
                hi2[a] = cutval
                lo2[a] = cutval
                new += [dict(start=st, count=left, lo=s["lo"], hi=hi2, ord=a), dict(start=st + left, count=c - left, lo=lo2, hi=s["hi"], ord=a)]
            segs = new
        for s in segs:
            if s["count"]:
                st, c = s["start"], s["count"]
                loc = np.arange(st, st + c) if s["ord"] == -1 else lists[s["ord"]][st:st + c]
                leaves.append(ids[loc])

    def upper(ids, lo, hi, sig):
        c = len(ids)
        if c <= root_size or c <= knn:
            tree(ids, lo, hi, sig)
            return
        a = cut_axis(lo, hi)
        left = c - c // 2
        order = sorted(ids.tolist(), key=lambda e: tuple_of(e, a, sig))   # (the device SELECTS the median of this order)
        median = tuple_of(order[left], a, sig)
        side = np.array([tuple_of(e, a, sig) >= median for e in ids])
        cutval = pts[order[left], a]
        hi2, lo2 = hi.copy(), lo.copy()
        hi2[a] = cutval
        lo2[a] = cutval
        nsig = [a] + [x for x in sig if x != a]
        upper(ids[~side], lo, hi2, nsig)       # stable: original-index order kept
        upper(ids[side], lo2, hi, nsig)

    upper(np.arange(n), lo.copy(), hi.copy(), [])
    return leaves


# ---- the upper levels' bookkeeping (csrc/lsgpu_ssn_select.hip.h) ------------------------------------------------------------
def block_tables_by_halving(n, levels, tile=2048):
    """What the host used to build and copy: per level the list of (first, count, segment, seg_start, seg_count, fb, nb)
    rows, from the recursion c -> (c - c // 2, c // 2), left child first."""
    out = []
    segs = [(0, n)]
    for _ in range(levels):
        rows, fb, nxt = [], 0, []
        for s, (st, c) in enumerate(segs):
            nb = (c + tile - 1) // tile
            for k in range(nb):
                rows.append((st + k * tile, min(tile, c - k * tile), s, st, c, fb, nb))
            fb += nb
            left = c - c // 2
            nxt += [(st, left), (st + left, c - left)]
        out.append(rows)
        segs = nxt
    return out


def block_tables_by_bit_path(n, levels, tile=2048):
    """k_gs_plan: every segment finds its (start, count) on its own from the bits of its index -- the path from the root,
    most significant bit first, 1 = right child -- then an exclusive scan of the block counts and, per row, a binary search
    for the segment that owns it."""
    out = []
    for L in range(levels):
        ns = 1 << L
        st_c = []
        for sg in range(ns):
            st, c = 0, n
            for bit in range(L - 1, -1, -1):
                left = c - c // 2
                if (sg >> bit) & 1:
                    st += left; c -= left
                else:
                    c = left
            st_c.append((st, c))
        nb = [(c + tile - 1) // tile for _, c in st_c]
        fbs = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
        rows = []
        for j in range(int(fbs[-1])):
            lo, hi = 0, ns
            while hi - lo > 1:
                mid = (lo + hi) >> 1
                if fbs[mid] <= j:
                    lo = mid
                else:
                    hi = mid
            k = j - int(fbs[lo]); st, c = st_c[lo]
            rows.append((st + k * tile, min(tile, c - k * tile), lo, st, c, int(fbs[lo]), int(fbs[lo + 1] - fbs[lo])))
        out.append(rows)
    return out


def radix_select_tuple(cands, target):
    """k_gs_select above 256 candidates: the tuple of rank `target` (0-based) among distinct 4-word tuples, by a radix select
    over their 16 bytes, most significant first, that stops as soon as one candidate has the chosen leading bytes.
    cands: (c, 4) uint32.  Returns the row index."""
    c = np.asarray(cands, np.uint64)
    chosen = [0, 0, 0, 0]
    rem = int(target)
    for p in range(16):
        field, shift = p >> 2, 24 - 8 * (p & 3)
        himask = 0 if shift == 24 else (0xFFFFFFFF << (shift + 8)) & 0xFFFFFFFF
        alive = np.ones(len(c), bool)
        for f in range(field):
            alive &= c[:, f] == chosen[f]
        alive &= ((c[:, field] ^ chosen[field]) & himask) == 0
        byte = ((c[:, field] >> shift) & 255).astype(np.int64)
        hist = np.bincount(byte[alive], minlength=256)
        excl = np.concatenate([[0], np.cumsum(hist)[:-1]])
        b = int(np.nonzero((hist > 0) & (excl <= rem) & (rem < excl + hist))[0][0])
        rem -= int(excl[b])
        chosen[field] |= b << shift
        if hist[b] == 1:
            lomask = (0xFFFFFFFF << shift) & 0xFFFFFFFF
            m = np.ones(len(c), bool)
            for f in range(field):
                m &= c[:, f] == chosen[f]
            m &= ((c[:, field] ^ chosen[field]) & lomask) == 0
            (i,) = np.nonzero(m)
            return int(i[0])
    raise AssertionError("equal tuples")
