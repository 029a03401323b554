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


def presorted_lists_signature(pts, knn, lo, hi):
    """lsgpu_ssn_levels.hip.h (upper levels, global memory): the same scheme WITHOUT cur_pos -- the members of a tie run
    of the cut axis are ordered by comparing their keys on the axes the segment was cut along before, most recent first,
    then their original index (the "signature" of the segment's current order); the partition's side flag comes from the
    position in the cut axis' list.  Returns the leaves like presorted_lists."""
    n = len(pts)
    keys = [order_key(pts[:, d]) for d in range(3)]
    lists = [np.argsort(keys[d], kind="stable") for d in range(3)]
    segs = [dict(start=0, count=n, lo=lo.copy(), hi=hi.copy(), sig=[])]
    while any(s["count"] > knn for s in segs):
        new = []
        side = np.zeros(n, bool)
        for s in segs:
            st, c = s["start"], s["count"]
            if c <= knn:
                new += [s, dict(start=st + c, count=0, lo=s["lo"], hi=s["hi"], sig=s["sig"])]
                continue
            a = cut_axis(s["lo"], s["hi"])
            la = lists[a]
            sig = s["sig"]
            if sig and sig[0] != a:
                others = [x for x in sig if x != a]
                seg = la[st:st + c].copy()
                out = seg.copy()
                k = keys[a][seg]
                cmpkey = lambda e: tuple(int(keys[x][e]) for x in others) + (int(e),)
                for i in range(c):
                    l, h = i, i + 1
                    while l > 0 and k[l - 1] == k[i]:
                        l -= 1
                    while h < c and k[h] == k[i]:
                        h += 1
                    if h - l > 1:
                        out[l + sum(1 for j in range(l, h) if cmpkey(seg[j]) < cmpkey(seg[i]))] = seg[i]
                la[st:st + c] = out
            left = c - c // 2
            side[la[st + left:st + c]] = True
            for d in range(3):
                if d != a:
                    seg = lists[d][st:st + c]
                    f = side[seg]
                    lists[d][st:st + c] = np.concatenate([seg[~f], seg[f]])
            cutval = pts[la[st + left], a]
            hi2, lo2 = s["hi"].copy(), s["lo"].copy()
            hi2[a] = cutval
            lo2[a] = cutval
            nsig = [a] + [x for x in sig if x != a]
            new += [dict(start=st, count=left, lo=s["lo"], hi=hi2, sig=nsig), dict(start=st + left, count=c - left, lo=lo2, hi=s["hi"], sig=nsig)]
        segs = new
    leaves = []
    for s in segs:
        if s["count"]:
            st, c = s["start"], s["count"]
            leaves.append(np.arange(st, st + c) if not s["sig"] else lists[s["sig"][0]][st:st + c].copy())
    return leaves
