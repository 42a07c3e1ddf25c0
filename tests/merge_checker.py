"""A SECOND, independent statement of the merge stage (GlbHashMap::mergeNewObsv, glb_hash_map.cu:146-207).

Test infrastructure.  Written from the reference's kernels themselves — MarkLimitedObserve / obtainFrontiers / UpdateHashBatch
(src/kernel/par_wave/unify_helper.cuh:201-523), raise_outside / lower_outside / lower_inside (wave_core.cuh:103-393), the level
loop of parWave (wave_helper.h:8-93) — and SURVEY.md App. E, WITHOUT looking at oracle/gie_oracle.c or at the HIP kernels: a
different author's-eye view, different data structures (one dense numpy "world" instead of hashed blocks, pairs kept as plain
(distance, wave-range coordinate) arrays instead of packed 64-bit words), and a different SCHEDULE: every frontier entry is
expanded one after the other in queue order with its stores visible at once (the serial order a single-threaded run of the
reference's kernels would produce), where the oracle and the HIP path use the order-independent canonical schedule of DESIGN.md.

What the two statements must agree on (SURVEY.md §7 "Determinism"): everything up to the seeds of the three waves is
deterministic in the reference, and the distance every voxel INSIDE the local volume ends up with did not depend on the
schedule in any probe; closest obstacles — and, rarely, distances of hashed voxels outside the volume — do.  For those the
definition is checked instead: a stored (distance, obstacle) pair is a witness, |obstacle - voxel|^2 == distance.

Inputs per map update come from the stages in front of the merge (the fused `_glb_type` plane and the batch EDT), which have
their own pins (brute-force EDT, float64 statements of the OGM kernels).
"""
import numpy as np

UNKNOWN, FREE, OCCUPIED, FNT = 0, 1, 2, 3
EMPTY_KEY = 999999
DIRS = ((-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1))     # glb_hash_map.h:59-61
GRAY0, GRAY1, BLACK = 16677219, 16677220, 16677223                              # voxmap_utils.cuh:17-22


class MergeChecker:
    def __init__(self, size, cutoff_sq, fast_mode, world_lo, world_hi):
        self.size = np.array(size, dtype=np.int64)
        X, Y, Z = size
        self.max_width = X + Y + Z
        self.max_loc_dist_sq = X * X + Y * Y + Z * Z
        if self.max_width < 1022:                                  # local_batch.h:51-58, voxmap_utils.cuh:8,161-165
            self.wr = np.array((2046, 2046, 1022)); self.empty = 999999; self.invalid_dist = 900000
        else:                                                      # the build's documented extension
            self.wr = np.array((16382, 16382, 8190)); self.empty = 4194303; self.invalid_dist = 4000000
        self.none = tuple(int(v) + 1 for v in self.wr)             # "0xffffffff": decodes to a coordinate outside every wave range
        self.cutoff_sq, self.fast = int(cutoff_sq), bool(fast_mode)
        self.lo = np.array(world_lo, dtype=np.int64)
        dims = tuple(int(v) for v in (np.array(world_hi) - self.lo))[::-1]      # arrays are [z][y][x]
        # GlbVoxel fields (voxmap_utils.cuh:29-44), one dense array each
        self.g_type = np.zeros(dims, np.int8)
        self.g_dist = np.full(dims, self.empty, np.int64)
        self.g_coc = np.full(dims + (3,), EMPTY_KEY, np.int64)
        self.g_wl = np.full(dims, -1, np.int64)
        self.g_uct = np.zeros(dims, np.int64)
        self.g_pd = np.zeros(dims, np.int64)                       # dist_id_pair: distance ...
        self.g_pp = np.zeros(dims + (3,), np.int64)                # ... and parent as a wave-range coordinate of the frame that wrote it
        n = (Z, Y, X)
        self.pd = np.zeros(n, np.int64)                            # _dist_id_pair, carried over by local index
        self.pp = np.zeros(n + (3,), np.int64)
        self.map_ct = 0

    # ---- helpers
    def _gi(self, g):
        i = (int(g[2] - self.lo[2]), int(g[1] - self.lo[1]), int(g[0] - self.lo[0]))
        assert min(i) >= 0 and all(i[k] < self.g_type.shape[k] for k in range(3)), "world box too small"
        return i

    def _in_loc(self, v):
        return 0 <= v[0] < self.size[0] and 0 <= v[1] < self.size[1] and 0 <= v[2] < self.size[2]

    def _in_wr(self, w):
        return 0 <= w[0] < self.wr[0] and 0 <= w[1] < self.wr[1] and 0 <= w[2] < self.wr[2]

    @staticmethod
    def _d2(a, b):
        return int((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2 + (a[2] - b[2]) ** 2)

    def update(self, pvt, upvt, T, bdist, bcoc):
        """One mergeNewObsv.  pvt / upvt: local and wave-range pivots; T: `_glb_type` after the fuse [Z][Y][X]; bdist, bcoc: batch
        EDT (`_aux`, and the closest obstacle in local coordinates, -1 where the volume holds no obstacle).  Returns the
        post-merge pair distances of the local volume and the seed counts (A, B, C) of obtainFrontiers."""
        self.map_ct += 1
        ct = self.map_ct
        pvt = np.array(pvt, dtype=np.int64); upvt = np.array(upvt, dtype=np.int64)
        X, Y, Z = (int(v) for v in self.size)
        T = np.array(T, dtype=np.int8, copy=True)
        o = pvt - self.lo
        vol = (slice(o[2], o[2] + Z), slice(o[1], o[1] + Y), slice(o[0], o[0] + X))
        assert min(o) >= 1 and all(o[k] + self.size[k] < self.g_type.shape[2 - k] for k in range(3)), "world box too small"
        self.g_type[vol] = T          # what the fuse left: the stored type of every voxel under the volume
        zz, yy, xx = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
        loc = np.stack([xx, yy, zz], -1).astype(np.int64)
        known = T != UNKNOWN

        # ---------------- MarkLimitedObserve (unify_helper.cuh:201-273), all voxels at once
        aux = bdist.astype(np.int64).copy()
        coc_new = bcoc.astype(np.int64).copy()
        invalid = (coc_new < 0).any(-1) | (coc_new > self.max_width).any(-1)          # invalid_coc_buf
        pd, pp = self.pd, self.pp
        m = known & invalid
        pd[m] = self.empty; pp[m] = self.none; aux[m] = self.empty
        coc_new[invalid] = (2045, 2045, 1021)                                        # INVALID_LOC_COC: outside every volume / wave range
        old_d = self.g_dist[vol]
        old_loc = self.g_coc[vol] - pvt
        old_in = ((old_loc >= 0) & (old_loc < self.size)).all(-1)
        lim = known & (bdist > old_d) & ~old_in
        coc_new[lim] = old_loc[lim]; aux[lim] = old_d[lim]
        w = coc_new + pvt - upvt
        inwr = ((w >= 0) & (w < self.wr)).all(-1)
        m = known & ~inwr
        pd[m] = self.empty; aux[m] = self.empty                                       # parent id left as it is
        m = known & inwr
        pd[m] = aux[m]; pp[m] = w[m]
        g_bak = np.where(known, aux, 0)                                               # `_g` / `_coc_idx`: the read-only backups
        c_bak = pp.copy()

        # ---------------- obtainFrontiers (unify_helper.cuh:275-446)
        wl = np.full((Z, Y, X), self.empty, np.int64)                                 # _loc_wave_layer
        cl = c_bak + upvt - pvt
        cl_in = ((cl >= 0) & (cl < self.size)).all(-1)
        act = known & cl_in
        fA, fB, fC = [], [], []
        nbr_unknown = np.zeros((Z, Y, X), bool)
        on_face = (xx == 0) | (yy == 0) | (zz == 0) | (xx == X - 1) | (yy == Y - 1) | (zz == Z - 1)
        # voxels off the faces: six neighbours inside the volume, direction by direction (a later match overwrites the pair)
        inner = act & ~on_face
        seeded = np.zeros((Z, Y, X), bool)
        for d in DIRS:
            sh = lambda a: np.roll(a, shift=(-d[2], -d[1], -d[0]), axis=(0, 1, 2))    # value of the neighbour in direction d
            n_unknown = sh(T) == UNKNOWN
            nbr_unknown |= inner & n_unknown
            nw = sh(c_bak)
            nl = nw + upvt - pvt
            n_in = ((nl >= 0) & (nl < self.size)).all(-1)
            n_valid = ((nw >= 0) & (nw < self.wr)).all(-1)
            dd = ((nl - loc) ** 2).sum(-1)
            hit = inner & ~n_unknown & ~n_in & n_valid & (dd < g_bak)
            pd[hit] = dd[hit]; pp[hit] = nw[hit]
            seeded |= hit
        for z, y, x in zip(*np.nonzero(seeded)):
            wl[z, y, x] = 1; fC.append((int(x), int(y), int(z)))
        # voxels on a face: some neighbours are hashed voxels outside the volume
        for z, y, x in zip(*np.nonzero(act & on_face)):
            v = (int(x), int(y), int(z))
            cur_w = tuple(int(t) for t in c_bak[z, y, x]); cur_l = tuple(int(t) for t in cl[z, y, x])
            cd = int(g_bak[z, y, x])
            in_q = False
            for d in DIRS:
                n = (v[0] + d[0], v[1] + d[1], v[2] + d[2])
                if self._in_loc(n):
                    if T[n[2], n[1], n[0]] == UNKNOWN:
                        nbr_unknown[z, y, x] = True
                        continue
                    nw = tuple(int(t) for t in c_bak[n[2], n[1], n[0]])
                    nl = tuple(int(nw[k] + upvt[k] - pvt[k]) for k in range(3))
                    if not self._in_loc(nl) and self._in_wr(nw):
                        dd = self._d2(nl, v)
                        if dd < cd:
                            pd[z, y, x] = dd; pp[z, y, x] = nw
                            if not in_q:
                                in_q = True; wl[z, y, x] = 1; fC.append(v)
                    continue
                ng = tuple(int(n[k] + pvt[k]) for k in range(3))
                gi = self._gi(ng)
                if self.g_type[gi] == UNKNOWN:
                    nbr_unknown[z, y, x] = True
                    continue
                nd = int(self.g_dist[gi])
                if nd < 0 or nd >= self.invalid_dist:
                    continue
                nc = tuple(int(t) for t in self.g_coc[gi])
                if max(nc) > 900000:
                    continue
                nw = tuple(int(nc[k] - upvt[k]) for k in range(3)); nl = tuple(int(nc[k] - pvt[k]) for k in range(3))
                n_local = self._in_loc(nl)
                if not n_local and self._in_wr(nw):
                    dd = self._d2(nl, v)
                    if dd < cd:
                        pd[z, y, x] = dd; pp[z, y, x] = nw
                        if not in_q:
                            in_q = True; wl[z, y, x] = 1; fC.append(v)
                if self.fast:
                    continue
                c2n = self._d2(n, cur_l)
                if c2n < nd:                                                          # lower out
                    self.g_wl[gi] = 1; self.g_uct[gi] = ct
                    self.g_pd[gi] = c2n; self.g_pp[gi] = cur_w
                    fB.append(ng)
                elif c2n > nd and n_local:                                            # raise out
                    if T[nl[2], nl[1], nl[0]] != OCCUPIED:
                        self.g_dist[gi] = c2n; self.g_coc[gi] = tuple(int(cur_l[k] + pvt[k]) for k in range(3))
                        self.g_wl[gi] = -ct
                        self.g_pd[gi] = c2n; self.g_pp[gi] = cur_w
                        fA.append(ng)
        T[act & (T == FREE) & nbr_unknown] = FNT
        seeds = (len(fA), len(fB), len(fC))

        # ---------------- the waves (glb_hash_map.cu:174-201): A feeds B, B feeds C
        if not self.fast:
            self._wave_a(fA, fB, pvt, upvt, aux, ct)
            self._wave_b(fB, fC, pvt, upvt, aux, pd, pp, wl, ct)
        self._wave_c(fC, pvt, upvt, pd, pp, wl)

        # ---------------- UpdateHashBatch (unify_helper.cuh:448-523)
        m = known & (pd != self.empty)
        gd = self.g_dist[vol]; gc = self.g_coc[vol]; gpd = self.g_pd[vol]; gpp = self.g_pp[vol]; gt = self.g_type[vol]
        gd[m] = pd[m]; gc[m] = pp[m] + upvt; gpd[m] = pd[m]; gpp[m] = pp[m]
        gt[m & (T == FNT)] = FNT
        self.T = T
        return pd.copy(), seeds

    # raise_outside, wave_core.cuh:103-224, under parWave's level loop
    def _wave_a(self, front, fB, pvt, upvt, aux, ct):
        cur = list(front)
        while cur:
            nxt = []
            for g in cur:
                gi = self._gi(g)
                if self.g_dist[gi] > self.cutoff_sq:
                    continue
                in_q = False
                lc = tuple(int(t) for t in self.g_coc[gi])
                lw = tuple(int(lc[k] - upvt[k]) for k in range(3))
                for d in DIRS:
                    ng = (g[0] + d[0], g[1] + d[1], g[2] + d[2])
                    if self._in_loc(tuple(int(ng[k] - pvt[k]) for k in range(3))):
                        continue
                    ni = self._gi(ng)
                    nd = int(self.g_dist[ni]); nc = tuple(int(t) for t in self.g_coc[ni])
                    if self.g_type[ni] == UNKNOWN or max(nc) > 900000 or nd < 0 or nd >= self.invalid_dist:
                        continue
                    if self.g_wl[ni] == -ct or self.g_uct[ni] == -ct:
                        continue
                    if nc == lc:
                        continue
                    raised = False
                    nl = tuple(int(nc[k] - pvt[k]) for k in range(3))
                    if self._in_loc(nl) and aux[nl[2], nl[1], nl[0]] != 0:           # its obstacle has disappeared
                        c2n = self._d2(lc, ng)
                        self.g_dist[ni] = c2n; self.g_coc[ni] = lc
                        self.g_wl[ni] = -ct; self.g_uct[ni] = -ct
                        self.g_pd[ni] = c2n; self.g_pp[ni] = lw
                        nxt.append(ng)
                        raised = True
                    if not raised:
                        n2c = self._d2(nc, g)
                        if self.g_dist[gi] > n2c:
                            self.g_dist[gi] = n2c; self.g_coc[gi] = nc
                            self.g_wl[gi] = 1; self.g_uct[gi] = ct
                            nw = tuple(int(nc[k] - upvt[k]) for k in range(3))
                            if not self._in_wr(nw):
                                continue
                            self.g_pd[gi] = n2c; self.g_pp[gi] = nw
                            if not in_q:
                                in_q = True; fB.append(g)
            cur = nxt

    # lower_outside, wave_core.cuh:229-350
    def _wave_b(self, front, fC, pvt, upvt, aux, pd, pp, wl, ct):
        cur = list(front)
        level = 0
        while cur:
            gray = GRAY0 if level % 2 == 0 else GRAY1
            nxt = []
            for g in cur:
                gi = self._gi(g)
                if self.g_dist[gi] > self.cutoff_sq:
                    continue
                self.g_wl[gi] = BLACK
                cw = tuple(int(t) for t in self.g_pp[gi])
                cc = tuple(int(cw[k] + upvt[k]) for k in range(3))
                self.g_coc[gi] = cc; self.g_dist[gi] = self.g_pd[gi]
                for d in DIRS:
                    ng = (g[0] + d[0], g[1] + d[1], g[2] + d[2])
                    nb = tuple(int(ng[k] - pvt[k]) for k in range(3))
                    cand = self._d2(cc, ng)
                    if not self._in_loc(nb):
                        ni = self._gi(ng)
                        if self.g_type[ni] == UNKNOWN:
                            continue
                        if max(int(t) for t in self.g_coc[ni]) > 900000:
                            continue
                        old = int(self.g_pd[ni])
                        if old > cand:                                                # id_atomicMin, wave_core.cuh:9-22
                            self.g_pd[ni] = cand; self.g_pp[ni] = cw
                            old_color = int(self.g_wl[ni]); self.g_wl[ni] = gray
                            if old_color == gray and self.g_uct[ni] == ct:
                                continue
                            self.g_uct[ni] = ct
                            nxt.append(ng)
                    else:
                        if aux[nb[2], nb[1], nb[0]] > cand:
                            pd[nb[2], nb[1], nb[0]] = cand; pp[nb[2], nb[1], nb[0]] = cw
                            if wl[nb[2], nb[1], nb[0]] == 1:
                                continue
                            fC.append(nb)
            cur = nxt
            level += 1

    # lower_inside, wave_core.cuh:353-393
    def _wave_c(self, front, pvt, upvt, pd, pp, wl):
        cur = list(front)
        level = 0
        while cur:
            gray = GRAY0 if level % 2 == 0 else GRAY1
            nxt = []
            for v in cur:
                wl[v[2], v[1], v[0]] = BLACK
                cw = tuple(int(t) for t in pp[v[2], v[1], v[0]])
                cl = tuple(int(cw[k] + upvt[k] - pvt[k]) for k in range(3))
                for d in DIRS:
                    n = (v[0] + d[0], v[1] + d[1], v[2] + d[2])
                    if not self._in_loc(n):
                        continue
                    cand = self._d2(cl, n)
                    if pd[n[2], n[1], n[0]] > cand:
                        pd[n[2], n[1], n[0]] = cand; pp[n[2], n[1], n[0]] = cw
                        if wl[n[2], n[1], n[0]] == gray:
                            continue
                        wl[n[2], n[1], n[0]] = gray
                        nxt.append(n)
            cur = nxt
            level += 1

    # ---- what a stored record has to satisfy whatever the schedule
    def witness_violations(self, pvt):
        """Known voxels of the world whose stored distance is valid but is not |closest obstacle - voxel|^2."""
        d = self.g_dist; c = self.g_coc
        valid = (self.g_type != UNKNOWN) & (d >= 0) & (d < self.invalid_dist) & (c <= 900000).all(-1)
        zz, yy, xx = np.nonzero(valid)
        g = np.stack([xx, yy, zz], -1) + self.lo
        return int((((c[zz, yy, xx] - g) ** 2).sum(-1) != d[zz, yy, xx]).sum())
