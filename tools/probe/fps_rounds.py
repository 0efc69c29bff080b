"""How many samples does a round of the third form of the FPS kernel take?  CPU model of its rule (numpy): every group
(a wave's buckets) offers its two best points; a round takes the longest prefix of the merged descending order in which no point
is lowered by an earlier one, cut behind the first second-best it contains.  Checks the picks against plain FPS.
usage: python tools/probe/fps_rounds.py [groups=16] [scene=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sad_amd
from sad_amd import synth

def morton(p, bits=10):
    lo, hi = p.min(0), p.max(0)
    q = ((p - lo) / (hi - lo + 1e-9) * ((1 << bits) - 1)).astype(np.uint64)
    def spread(v):
        r = np.zeros_like(v)
        for b in range(bits):
            r |= ((v >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b)
        return r
    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))

def rounds(p, M, ng):
    N = len(p)
    mind = np.full(N, np.inf, np.float32)
    picks = [0]
    mind = np.minimum(mind, ((p - p[0]) ** 2).sum(1).astype(np.float32))
    hist = {}
    while len(picks) < M:
        cand = []
        for g in range(ng):
            a, b = g * N // ng, (g + 1) * N // ng
            v = mind[a:b]
            top = np.argpartition(-v, 1)[:2]
            top = top[np.argsort(-v[top], kind="stable")]
            cand += [(v[top[0]], a + top[0], 1), (v[top[1]], a + top[1], 2)]
        cand.sort(key=lambda c: -c[0])
        acc = []
        for val, i, which in cand:
            if len(picks) + len(acc) >= M or (val == 0 and acc) or any(((p[i] - p[j]) ** 2).sum() < val for j in acc):
                break
            acc.append(i)
            if which == 2:
                break
        for i in acc:
            mind = np.minimum(mind, ((p - p[i]) ** 2).sum(1).astype(np.float32))
        picks += acc
        hist[len(acc)] = hist.get(len(acc), 0) + 1
    return picks, hist

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 16
scene = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pts = synth.make_scene(scene, 16384)[:, :3].astype(np.float32)
order = np.argsort(morton(pts), kind="stable")
p = pts[order]
first = int(np.where(order == 0)[0][0])
p = np.concatenate([p[first:first + 1], p[:first], p[first + 1:]])      # the start point first
picks, hist = rounds(p, 4096, ng)
n = sum(hist.values())
print(f"16384 -> 4096, {ng} groups: {n} rounds, {4095 / n:.2f} samples per round; rounds by samples taken: {dict(sorted(hist.items()))}")
mind = np.full(len(p), np.inf, np.float32); ref = [0]
for _ in range(600):
    mind = np.minimum(mind, ((p - p[ref[-1]]) ** 2).sum(1).astype(np.float32)); ref.append(int(np.argmax(mind)))
print("first 600 samples equal plain FPS:", picks[:601] == ref)
