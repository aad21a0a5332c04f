"""Development aid (CPU, numpy): what would a PER-LANE walk of a small BVH over the clusters cost, in wave steps, against the
wave-uniform loop over all cluster boxes that nearest_hit_culled's step 1 runs today (128 steps for 1024 objects)?

A divergent per-lane loop costs a wave the MAXIMUM over its 64 lanes.  The sim builds bench.py's L* scene, the median-split
clusters of rt_cull.h, the binary tree those splits form and its 4-wide collapse; traces camera rays, two generations of diffuse
bounce rays and soft-shadow taps with plain numpy slab / sphere tests; groups rays into waves (64 consecutive rays in 8x8-block
order: the coherent end -- and 64 random rays of a generation: the incoherent end; the wavefront kernel's mix of eight pixels'
samples per wave lies between the two) and prints, per generation: clusters touched per ray, and for the binary tree (nodes
visited) and the 4-wide one (nodes visited, 4 box tests each) the mean per ray and the mean over waves of the MAX over lanes.
usage: bvh_walk_sim.py [objects [width height]]"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "tests")); sys.path.insert(0, os.path.join(HERE, "..", ".."))
from rtlibs import large_scene, LARGE_SCENE_CAMERA, scene_objects      # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (320, 184)
CL = 8
rng = np.random.default_rng(5)
objs, n = scene_objects(large_scene(N, seed=17))
objs = objs[:n]
is_sphere = objs["type"] == 1
g = objs["geom"].astype(np.float64)
lo = np.where(is_sphere[:, None], g[:, :3] - g[:, 3:4], g[:, :3])
hi = np.where(is_sphere[:, None], g[:, :3] + g[:, 3:4], g[:, :3] + g[:, 3:6])
centre = 0.5 * (lo + hi)

# ---- the median-split tree of rt_cull.h (ranges cut at a multiple of CL nearest the middle, along the widest axis of the centres)
ids = np.arange(n)
nodes = []          # (first, last, left, right) ; leaves have left = -1


def split(first, last):
    me = len(nodes)
    nodes.append([first, last, -1, -1])
    if last - first <= CL:
        return me
    c = centre[ids[first:last]]
    axis = int(np.argmax(c.max(0) - c.min(0)))
    clusters = (last - first + CL - 1) // CL
    mid = first + (clusters // 2) * CL
    order = np.lexsort((ids[first:last], c[:, axis]))
    ids[first:last] = ids[first:last][order]
    nodes[me][2] = split(first, mid)
    nodes[me][3] = split(mid, last)
    return me


split(0, n)
nodes = np.array(nodes)
NN = len(nodes)
node_lo = np.array([lo[ids[a:b]].min(0) for a, b, _, _ in nodes])
node_hi = np.array([hi[ids[a:b]].max(0) for a, b, _, _ in nodes])
leaf = nodes[:, 2] < 0
parent = np.full(NN, -1)
for i, (_, _, l, r) in enumerate(nodes):
    if l >= 0:
        parent[l] = i; parent[r] = i
depth = np.zeros(NN, int)
for i in range(1, NN):
    depth[i] = depth[parent[i]] + 1
print(f"{n} objects, {int(leaf.sum())} clusters, {NN} tree nodes, depth {depth.max()}")


def slab(o, d, blo, bhi):
    """rays (R,3) against boxes (B,3): (R,B) bool -- enter <= leave and leave >= 0"""
    inv = 1.0 / d
    a = (blo[None] - o[:, None]) * inv[:, None]; b = (bhi[None] - o[:, None]) * inv[:, None]
    enter = np.minimum(a, b).max(2); leave = np.maximum(a, b).min(2)
    return (enter <= leave) & (leave >= 0)


def nearest(o, d):
    """nearest object hit of each ray: (t, object or -1, normal)"""
    R = len(o)
    best_t = np.full(R, np.inf); best_i = np.full(R, -1)
    for c0 in range(0, n, 128):
        sl = slice(c0, min(c0 + 128, n))
        # boxes
        inv = 1.0 / d
        a = (lo[None, sl] - o[:, None]) * inv[:, None]; b = (hi[None, sl] - o[:, None]) * inv[:, None]
        enter = np.minimum(a, b).max(2); leave = np.maximum(a, b).min(2)
        tb = np.where((enter <= leave) & (enter >= 0) & ~is_sphere[None, sl], enter, np.inf)
        # spheres
        oc = g[None, sl, :3] - o[:, None]
        bq = (oc * d[:, None]).sum(2); cq = (oc * oc).sum(2) - g[None, sl, 3] ** 2
        disc = bq * bq - cq
        root = np.sqrt(np.maximum(disc, 0)); t0 = bq - root; t1 = bq + root
        ts = np.where(t0 >= 0, t0, t1)
        ts = np.where((disc > 0) & (ts >= 0) & is_sphere[None, sl], ts, np.inf)
        t = np.minimum(tb, ts)
        k = t.argmin(1); tk = t[np.arange(R), k]
        better = tk < best_t
        best_t = np.where(better, tk, best_t); best_i = np.where(better, k + c0, best_i)
    p = o + d * np.where(np.isfinite(best_t), best_t, 0)[:, None]
    nrm = np.zeros_like(p)
    hit = best_i >= 0
    s = hit & is_sphere[np.maximum(best_i, 0)]
    nrm[s] = (p[s] - g[best_i[s], :3]); nrm[s] /= np.linalg.norm(nrm[s], axis=1, keepdims=True)
    bx = hit & ~s
    if bx.any():
        q = p[bx]; bl = lo[best_i[bx]]; bh = hi[best_i[bx]]
        dist = np.concatenate([np.abs(q - bl), np.abs(q - bh)], 1)
        f = dist.argmin(1)
        nb = np.zeros_like(q); nb[np.arange(len(q)), f % 3] = np.where(f < 3, -1.0, 1.0)
        nrm[bx] = nb
    return best_t, best_i, p, nrm


def walk_costs(o, d, label, coherent_order):
    hit_box = np.concatenate([slab(o[i:i + 4096], d[i:i + 4096], node_lo, node_hi) for i in range(0, len(o), 4096)])
    # binary: a node is VISITED (its box tested) when its parent is visited and hit; the root is always visited
    visited = np.zeros_like(hit_box); visited[:, 0] = True
    for i in range(1, NN):
        visited[:, i] = visited[:, parent[i]] & hit_box[:, parent[i]]
    touched = (visited & hit_box & leaf[None]).sum(1)
    # what front-to-back order with early termination could save: the clusters a ray enters before (or at) its nearest hit, and the
    # same if only the NEAREST touched cluster were looked at first and the others then filtered by the hit found there
    leaves_ = np.flatnonzero(leaf)
    t_hit = nearest(o, d)[0]
    inv = 1.0 / d
    ent = np.full((len(o), len(leaves_)), np.inf)
    for i0 in range(0, len(o), 4096):
        sl = slice(i0, i0 + 4096)
        a = (node_lo[leaves_][None] - o[sl, None]) * inv[sl, None]; b = (node_hi[leaves_][None] - o[sl, None]) * inv[sl, None]
        en = np.minimum(a, b).max(2); lv = np.maximum(a, b).min(2)
        ent[sl] = np.where((en <= lv) & (lv >= 0), np.maximum(en, 0), np.inf)
    before_hit = ((ent <= t_hit[:, None]) & np.isfinite(ent)).sum(1)
    print(f"    front to back: clusters touched {touched.mean():.2f} per ray, of which entered no later than the nearest hit {before_hit.mean():.2f} (rays that hit nothing: {np.mean(~np.isfinite(t_hit)):.2f} of all)")
    vis2 = visited.sum(1)
    # 4-wide: nodes at even depth are the 4-wide nodes; visiting one tests its (up to) four grandchildren's boxes -- or its children's where those are leaves
    wide = (depth % 2 == 0)
    grand = np.where(parent >= 0, parent[np.maximum(parent, 0)], -1)
    v4 = np.zeros_like(hit_box); v4[:, 0] = True
    for i in range(1, NN):
        if wide[i] and not leaf[i]:
            v4[:, i] = v4[:, grand[i]] & hit_box[:, i]
    vis4 = (v4 & (wide & ~leaf)[None]).sum(1)
    R = len(o)
    # the hybrid that was built (round 6): GROUPS of G consecutive clusters (tree order) tested wave-uniformly, then the wave's (ray, group)
    # pairs dealt 64 at a time, each lane testing its pair's G cluster boxes: steps = clusters / G uniform + ceil(pairs / 64) dealt
    leaves = np.flatnonzero(leaf)
    hyb = []
    for G in (4, 8, 16):
        ng = (len(leaves) + G - 1) // G
        glo = np.array([node_lo[leaves[k * G:(k + 1) * G]].min(0) for k in range(ng)]); ghi = np.array([node_hi[leaves[k * G:(k + 1) * G]].max(0) for k in range(ng)])
        gh = np.concatenate([slab(o[i:i + 4096], d[i:i + 4096], glo, ghi) for i in range(0, len(o), 4096)]).sum(1)
        hyb.append((G, ng, gh))
    out = []
    for name, order in (("coherent", coherent_order), ("random", rng.permutation(R))):
        m = (R // 64) * 64
        idx = order[:m].reshape(-1, 64)
        out.append((name, vis2[idx].max(1).mean(), vis4[idx].max(1).mean(), touched[idx].max(1).mean()))
    print(f"{label:34s} rays {R:7d}  clusters touched/ray {touched.mean():5.2f}   binary: nodes/ray {vis2.mean():5.1f}   4-wide: nodes/ray {vis4.mean():5.2f} (x4 box tests = {4 * vis4.mean():5.1f})")
    for G, ng, gh in hyb:
        m = (R // 64) * 64
        pairs = gh[coherent_order[:m]].reshape(-1, 64).sum(1)
        print(f"    hybrid, groups of {G:2d}: {ng:3d} uniform box tests + groups hit/ray {gh.mean():4.2f} -> {np.ceil(pairs / 64).mean():4.1f} dealt steps of {G} box tests per wave of 64 rays "
              f"(= {ng + G * np.ceil(pairs / 64).mean():5.1f} box-test steps; lanes busy in a dealt step {(pairs / (64 * np.ceil(pairs / 64).clip(1))).mean():.2f})")
    for name, m2, m4, mt in out:
        print(f"    waves of 64 {name:9s}: max over lanes -- clusters {mt:5.1f}   binary nodes {m2:6.1f}   4-wide nodes {m4:5.1f} (x4 = {4 * m4:5.1f} box tests; today: {int(leaf.sum())} wave-uniform box tests)")
    return touched


cam = LARGE_SCENE_CAMERA
pos = np.array(cam["pos"], float); front = np.array(cam["front"], float); front /= np.linalg.norm(front)
up = np.array(cam["up"], float); right = np.cross(front, up); right /= np.linalg.norm(right); upv = np.cross(right, front)
half = np.tan(cam["fov"] / 2)
# pixels in 8x8-block order (rt_primary_pass)
ys, xs = np.mgrid[0:H, 0:W]
blk = (ys // 8) * ((W + 7) // 8) + xs // 8
order = np.lexsort(((xs % 8).ravel(), (ys % 8).ravel(), blk.ravel()))
px = xs.ravel()[order]; py = ys.ravel()[order]
u = (px / (W - 1) - 0.5) * 2 * half * W / H; v = (py / (H - 1) - 0.5) * 2 * half
d = front[None] + u[:, None] * right[None] + v[:, None] * upv[None]; d /= np.linalg.norm(d, axis=1, keepdims=True)
o = np.repeat(pos[None], len(d), 0)
light = g[n // 2, :3]
gen = 0
coh = np.arange(len(o))
while gen < 3 and len(o) >= 64:
    walk_costs(o, d, f"generation {gen} rays", coh)
    t, i, p, nrm = nearest(o, d)
    hit = i >= 0
    p, nrm = p[hit], nrm[hit]
    if len(p) < 64:
        break
    # soft-shadow taps: towards the emitter, jittered (main.c:191-207: normalize(rd * 0.5 + L))
    rd = rng.uniform(-1, 1, (len(p), 3)); rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    L = light[None] - p
    td = rd * 0.5 + L; td /= np.linalg.norm(td, axis=1, keepdims=True)
    walk_costs(p + td * 1e-3, td, f"  taps from generation {gen} hits", np.arange(len(p)))
    # diffuse bounce
    nd = rng.uniform(-1, 1, (len(p), 3)); nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    nd = np.where(((nd * nrm).sum(1) < 0)[:, None], -nd, nd)
    o, d = p + nd * 1e-3, nd
    d = np.where(np.abs(d) < 1e-9, 1e-9, d)
    coh = np.arange(len(o))
    gen += 1
