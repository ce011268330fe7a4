"""Synthetic Hex8 box meshes in the reference's unit-test style (`generated:IxJxK`,
unit_tests/UnitTestUtils.C:77-90): node id = 1 + i + nxn*j + nxn*nyn*k, lattice edges
oriented low id -> high id, z-slab decomposition with the lower rank owning the shared plane.

This is harness code (mesh + partition for tests and bench.py); the hot path is in csrc/."""
import numpy as np


def split_elems(nez, nranks):
    """element z-layers [e0, e1) of each rank (Ioss-style contiguous slabs)"""
    base, rem = divmod(nez, nranks)
    out, e = [], 0
    for r in range(nranks):
        n = base + (1 if r < rem else 0)
        out.append((e, e + n))
        e += n
    return out


def box_edges(nx, ny, nz):
    """Edges of an nx x ny x nz node lattice, sorted by (L, R), as an (E, 2) int32 array and
    the direction (0, 1, 2) of every edge."""
    n = nx * ny * nz
    idx = np.arange(n, dtype=np.int64)
    i = idx % nx
    j = (idx // nx) % ny
    k = idx // (nx * ny)
    hx, hy, hz = i < nx - 1, j < ny - 1, k < nz - 1
    cnt = hx.astype(np.int64) + hy + hz
    off = np.zeros(n + 1, np.int64)
    np.cumsum(cnt, out=off[1:])
    ne = int(off[-1])
    en = np.empty((ne, 2), np.int32)
    ed = np.empty(ne, np.int8)
    px = off[:-1][hx]
    en[px, 0], en[px, 1], ed[px] = idx[hx], idx[hx] + 1, 0
    py = (off[:-1] + hx)[hy]
    en[py, 0], en[py, 1], ed[py] = idx[hy], idx[hy] + nx, 1
    pz = (off[:-1] + hx + hy)[hz]
    en[pz, 0], en[pz, 1], ed[pz] = idx[hz], idx[hz] + nx * ny, 2
    return en, ed


class BoxSlab:
    """Rank `rank` of `nranks` of a box with (nxn, nyn, nzn) nodes, in local numbering:
    owned nodes (sorted by gid), then the shared-not-owned plane (owner rank-1), then the
    ghost plane (owner rank+1) -- the TpetraLinearSystem order
    (src/TpetraLinearSystem.C:247-285, 1046-1074).  Local edges = every edge touching an
    owned node (cut edges are duplicated on both sides, see include/nalu_gpu.h)."""

    def __init__(self, nxn, nyn, nzn, rank=0, nranks=1, lengths=(1.0, 1.0, 1.0), edge_order="sorted"):
        assert nzn - 1 >= nranks, "need at least one element layer per rank"
        self.dims = (nxn, nyn, nzn)
        self.rank, self.nranks, self.lengths = rank, nranks, lengths
        e0, e1 = split_elems(nzn - 1, nranks)[rank]
        plane = nxn * nyn
        self.plane = plane
        k_own0 = 0 if rank == 0 else e0 + 1
        k_own1 = e1  # inclusive
        kA = 0 if rank == 0 else e0
        kB = e1 if rank == nranks - 1 else e1 + 1
        self.k_range = (kA, kB)
        nzl = kB - kA + 1
        self.n_owned = (k_own1 - k_own0 + 1) * plane
        self.n_shared = plane if rank > 0 else 0
        self.n_ghost = plane if rank < nranks - 1 else 0
        self.n_local = self.n_owned + self.n_shared + self.n_ghost
        # local box index t -> local id
        t = np.arange(nzl * plane, dtype=np.int64)
        kk = t // plane + kA
        lid = np.empty(nzl * plane, np.int64)
        own = (kk >= k_own0) & (kk <= k_own1)
        lid[own] = t[own] - (k_own0 - kA) * plane
        if rank > 0:
            sh = kk == e0
            lid[sh] = self.n_owned + (t[sh] % plane)
        if rank < nranks - 1:
            gh = kk == e1 + 1
            lid[gh] = self.n_owned + self.n_shared + (t[gh] % plane)
        # per local id: lattice indices, gid, owner
        order = np.argsort(lid)
        tt = t[order]
        self.i = (tt % nxn).astype(np.int32)
        self.j = ((tt // nxn) % nyn).astype(np.int32)
        self.k = (tt // plane + kA).astype(np.int32)
        self.gids = 1 + self.i.astype(np.int64) + nxn * self.j.astype(np.int64) + \
            plane * self.k.astype(np.int64)
        self.owners = np.full(self.n_local, rank, np.int32)
        self.owners[self.n_owned:self.n_owned + self.n_shared] = rank - 1
        self.owners[self.n_owned + self.n_shared:] = rank + 1
        # edges of the local box that touch an owned node
        en, ed = box_edges(nxn, nyn, nzl)
        keep = own[en[:, 0]] | own[en[:, 1]]
        en, ed = en[keep], ed[keep]
        if edge_order == "slot":
            # slot-major: all first "upper" edges of the rows, then the second, ... (here:
            # x-, y-, z-edges).  The k-th entries of consecutive rows then reference
            # consecutive edges, so the edge-field gathers of the row kernels coalesce.
            o = np.argsort(ed, kind="stable")
            en, ed = en[o], ed[o]
        self._lid, self._nzl = lid, nzl
        self.edge_nodes = np.ascontiguousarray(lid[en].astype(np.int32))
        self.edge_dir = ed
        self.n_edges = len(ed)
        # global edge key for comparisons with a global mesh: (gidL-1, gidR-1)
        # halo plan
        nb, so, sl, ro = [], [0], [], [0]
        if rank > 0:  # rank-1 needs my first owned plane as its ghost plane
            nb.append(rank - 1)
            sl.append(np.arange(0, plane, dtype=np.int32))
            so.append(so[-1] + plane)
            ro.append(ro[-1] + plane)
        if rank < nranks - 1:  # rank+1 needs my last owned plane (its shared plane)
            nb.append(rank + 1)
            sl.append(np.arange(self.n_owned - plane, self.n_owned, dtype=np.int32))
            so.append(so[-1] + plane)
            ro.append(ro[-1] + plane)
        self.neighbors = np.array(nb, np.int32)
        self.send_offsets = np.array(so, np.int32)
        self.send_lids = np.concatenate(sl) if sl else np.zeros(0, np.int32)
        self.recv_offsets = np.array(ro, np.int32)

    def hex_nodes(self):
        """(n_elem, 8) int32 local node ids of the Hex8 elements of the local box, node order
        of the reference's generated meshes (bottom face counter-clockwise, then the top)."""
        nxn, nyn, _ = self.dims
        i, j, k = np.meshgrid(np.arange(nxn - 1), np.arange(nyn - 1), np.arange(self._nzl - 1),
                              indexing="ij")
        n0 = (i + nxn * (j + nyn * k)).transpose(2, 1, 0).reshape(-1)  # i fastest
        sx, sy, sz = 1, nxn, nxn * nyn
        h = np.stack([n0, n0 + sx, n0 + sx + sy, n0 + sy, n0 + sz, n0 + sx + sz,
                      n0 + sx + sy + sz, n0 + sy + sz], axis=1)
        return np.ascontiguousarray(self._lid[h].astype(np.int32))

    # uniform lattice geometry (exact for the unperturbed box)
    def spacing(self):
        nxn, nyn, nzn = self.dims
        return tuple(self.lengths[d] / max(n - 1, 1) for d, n in enumerate((nxn, nyn, nzn)))

    def coords(self):
        """(3, n_local) SoA coordinates"""
        hx, hy, hz = self.spacing()
        return np.stack([self.i * hx, self.j * hy, self.k * hz]).astype(np.float64)

    def _frac(self):
        nxn, nyn, nzn = self.dims
        fx = np.where((self.i == 0) | (self.i == nxn - 1), 0.5, 1.0)
        fy = np.where((self.j == 0) | (self.j == nyn - 1), 0.5, 1.0)
        fz = np.where((self.k == 0) | (self.k == nzn - 1), 0.5, 1.0)
        return fx, fy, fz

    def uniform_geometry(self):
        """dual_nodal_volume (n_local) and edge_area_vector (3, n_edges) of the uniform box
        (what the Hex8 SCV/SCS sums give on an undistorted lattice)."""
        hx, hy, hz = self.spacing()
        fx, fy, fz = self._frac()
        vol = hx * hy * hz * fx * fy * fz
        L = self.edge_nodes[:, 0]
        area = np.zeros((3, self.n_edges))
        d = self.edge_dir
        area[0, d == 0] = (hy * hz * fy[L] * fz[L])[d == 0]
        area[1, d == 1] = (hx * hz * fx[L] * fz[L])[d == 1]
        area[2, d == 2] = (hx * hy * fx[L] * fy[L])[d == 2]
        return vol, area


class BoxBlock:
    """Rank `rank` of a (px, py, pz) block partition of a box with (nxn, nyn, nzn) nodes: every
    rank gets a brick of elements, a node on a cut plane is owned by the lowest rank that holds it
    (the reference's STK rule), so a rank has up to 7 neighbours on a 2 x 2 x 2 partition -- one
    per xGMI link of an 8-GPU node.  Same attributes as BoxSlab; the
    shared-not-owned nodes (sorted by global id, src/TpetraLinearSystem.C:247-285) have several
    owners here, so the halo plan carries a receive index list (`recv_lids`,
    nalu_gpu_set_halo_plan_indexed)."""

    def __init__(self, nxn, nyn, nzn, rank, parts, lengths=(1.0, 1.0, 1.0), edge_order="sorted"):
        self.dims = (nxn, nyn, nzn)
        self.parts = parts
        self.rank, self.nranks, self.lengths = rank, parts[0] * parts[1] * parts[2], lengths
        info = self._rank_nodes(rank)
        own, sh, gh = info["owned"], info["shared"], info["ghost"]
        self.n_owned, self.n_shared, self.n_ghost = len(own[0]), len(sh[0]), len(gh[0])
        self.n_local = self.n_owned + self.n_shared + self.n_ghost
        self.gids = np.concatenate([own[0], sh[0], gh[0]])
        self.owners = np.concatenate([own[1], sh[1], gh[1]]).astype(np.int32)
        g0 = self.gids - 1
        self.i = (g0 % nxn).astype(np.int32)
        self.j = ((g0 // nxn) % nyn).astype(np.int32)
        self.k = (g0 // (nxn * nyn)).astype(np.int32)
        # gid -> local id (all in numpy: a rank of a 512^3 box holds 17 M nodes)
        order = np.argsort(self.gids, kind="stable")
        sg = self.gids[order]

        def lid_of(g):
            pos = np.searchsorted(sg, g)
            assert np.array_equal(sg[pos], g), "node is not held by this rank"
            return order[pos]
        # local edges: every lattice edge with an owned end, oriented low gid -> high gid, sorted (L, R)
        step = (1, nxn, nxn * nyn)
        lim = (nxn, nyn, nzn)
        idx = (self.i, self.j, self.k)
        no = self.n_owned
        og = self.gids[:no]
        on = np.arange(no, dtype=np.int64)
        ens, eds = [], []
        for d in range(3):
            up = idx[d][:no] + 1 < lim[d]
            ens.append(np.stack([on[up], lid_of(og[up] + step[d])], axis=1))
            eds.append(np.full(int(up.sum()), d, np.int8))
            dn = idx[d][:no] > 0
            m = lid_of(og[dn] - step[d])
            sel = m >= no  # (owned-owned edges are listed from their lower end)
            ens.append(np.stack([m[sel], on[dn][sel]], axis=1))
            eds.append(np.full(int(sel.sum()), d, np.int8))
        en, ed = np.concatenate(ens).astype(np.int64).reshape(-1, 2), np.concatenate(eds)
        gl, gr = self.gids[en[:, 0]], self.gids[en[:, 1]]
        o = np.lexsort((gr, gl))
        if edge_order == "slot":
            o = o[np.argsort(ed[o], kind="stable")]
        self.edge_nodes = np.ascontiguousarray(en[o].astype(np.int32))
        self.edge_dir = ed[o]
        self.n_edges = len(ed)
        # halo plan: what each neighbour holds of mine, in ITS local order; what I hold of its
        nbrs = sorted(set(int(r) for r in np.unique(self.owners[self.n_owned:])) |
                      set(q for q in self._adjacent_ranks() if self._needs_from(q, rank)))
        so, sl, ro, rl = [0], [], [0], []
        for q in nbrs:
            theirs = self._rank_nodes(q)
            tg = np.concatenate([theirs["shared"][0], theirs["ghost"][0]])
            to = np.concatenate([theirs["shared"][1], theirs["ghost"][1]])
            send = lid_of(tg[to == rank])
            sl.append(send)
            so.append(so[-1] + len(send))
            mine = np.nonzero(self.owners[self.n_owned:] == q)[0] + self.n_owned
            rl.append(mine)
            ro.append(ro[-1] + len(mine))
        keep = [n for n in range(len(nbrs)) if so[n + 1] > so[n] or ro[n + 1] > ro[n]]
        self.neighbors = np.array([nbrs[n] for n in keep], np.int32)
        self.send_offsets = np.array([0] + [so[n + 1] for n in keep], np.int32)
        self.recv_offsets = np.array([0] + [ro[n + 1] for n in keep], np.int32)
        self.send_lids = (np.concatenate(sl) if sl else np.zeros(0)).astype(np.int32)
        self.recv_lids = (np.concatenate(rl) if rl else np.zeros(0)).astype(np.int32)

    # ---- partition geometry
    def _coords_of(self, rank):
        px, py, _ = self.parts
        return rank % px, (rank // px) % py, rank // (px * py)

    def _elem_ranges(self, rank):
        return [split_elems(self.dims[d] - 1, self.parts[d])[c] for d, c in enumerate(self._coords_of(rank))]

    def _adjacent_ranks(self):
        px, py, pz = self.parts
        a, b, c = self._coords_of(self.rank)
        out = []
        for dc in (-1, 0, 1):
            for db in (-1, 0, 1):
                for da in (-1, 0, 1):
                    x, y, z = a + da, b + db, c + dc
                    if (da or db or dc) and 0 <= x < px and 0 <= y < py and 0 <= z < pz:
                        out.append(x + px * (y + py * z))
        return out

    def _needs_from(self, q, owner):
        t = self._rank_nodes(q)
        return bool((np.concatenate([t["shared"][1], t["ghost"][1]]) == owner).any())

    def _dim_owner(self, d):
        """per node index of dimension d: the lowest part whose element range holds it"""
        cache = self.__dict__.setdefault("_dim_owner_cache", {})
        if d not in cache:
            rng = split_elems(self.dims[d] - 1, self.parts[d])
            ow = np.empty(self.dims[d], np.int64)
            for p in reversed(range(self.parts[d])):  # (lower parts overwrite: the lowest holder owns)
                ow[rng[p][0]:rng[p][1] + 1] = p
            cache[d] = ow
        return cache[d]

    def _owner_of(self, i, j, k):
        """lowest rank whose element brick holds node (i, j, k) (arrays or scalars)"""
        c = [self._dim_owner(d)[v] for d, v in enumerate((i, j, k))]
        return c[0] + self.parts[0] * (c[1] + self.parts[1] * c[2])

    def _rank_nodes(self, rank):
        """gids and owners of a rank's owned nodes (sorted by gid), shared-not-owned nodes (sorted by gid) and
        ghosts (lattice neighbours of owned nodes outside the brick, sorted by (owner, gid))"""
        cache = self.__dict__.setdefault("_cache", {})
        if rank in cache:
            return cache[rank]
        nxn, nyn, nzn = self.dims
        er = self._elem_ranges(rank)
        ax = [np.arange(er[d][0], er[d][1] + 1, dtype=np.int64) for d in range(3)]
        k, j, i = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")  # (i fastest: ascending gid)
        i, j, k = i.reshape(-1), j.reshape(-1), k.reshape(-1)
        gid = 1 + i + nxn * (j + nyn * k)
        own = self._owner_of(i, j, k)
        mine = own == rank
        owned = (gid[mine], own[mine])
        shared = (gid[~mine], own[~mine])
        # ghosts: a neighbour that differs in one coordinate leaves the brick iff that coordinate leaves its range
        oi = (i[mine], j[mine], k[mine])
        gh = []
        for d, lim in enumerate((nxn, nyn, nzn)):
            for s in (-1, 1):
                v = oi[d] + s
                out = (v >= 0) & (v < lim) & ((v < er[d][0]) | (v > er[d][1]))
                c = [oi[0][out], oi[1][out], oi[2][out]]
                c[d] = v[out]
                gh.append(np.stack([1 + c[0] + nxn * (c[1] + nyn * c[2]), self._owner_of(*c)], axis=1))
        gh = np.unique(np.concatenate(gh), axis=0) if gh else np.zeros((0, 2), np.int64)
        o = np.lexsort((gh[:, 0], gh[:, 1]))
        out = dict(owned=owned, shared=shared, ghost=(gh[o, 0], gh[o, 1]))
        cache[rank] = out
        return out

    # ---- geometry (as BoxSlab)
    spacing = BoxSlab.spacing
    coords = BoxSlab.coords
    _frac = BoxSlab._frac
    uniform_geometry = BoxSlab.uniform_geometry


# ---------------------------------------------------------------------------------------------
# Reference-style ("export mode") bookkeeping of a partition: the elements a rank owns, the edges of
# those elements, who owns and who shares an edge -- what STK's comm lists give the reference
# (src/ComputeGeometryAlgorithmDriver.C:126-133 sums edge_area_vector over the sharers).
HEX_EDGES = ((0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7))


def _parts(S):
    return S.parts if isinstance(S, BoxBlock) else (1, 1, S.nranks)


def _dim_ranges(S):
    """per dimension: the inclusive node range of every part's bricks"""
    return [split_elems(S.dims[d] - 1, _parts(S)[d]) for d in range(3)]


def _lid_lookup(S):
    order = np.argsort(S.gids, kind="stable")
    sg = S.gids[order]

    def look(g):
        pos = np.searchsorted(sg, g)
        assert np.array_equal(sg[pos], g), "node is not held by this rank"
        return order[pos]
    return look


def own_hex_nodes(S):
    """(n_elem, 8) int32 local node ids of the Hex8 elements of the rank's OWN brick (its elements in the
    reference's sense; BoxSlab.hex_nodes() also returns the neighbour's first layer)"""
    px, py, pz = _parts(S)
    c = (S.rank % px, (S.rank // px) % py, S.rank // (px * py))
    r = [_dim_ranges(S)[d][c[d]] for d in range(3)]
    nxn, nyn, _ = S.dims
    i, j, k = np.meshgrid(np.arange(r[0][0], r[0][1]), np.arange(r[1][0], r[1][1]), np.arange(r[2][0], r[2][1]),
                          indexing="ij")
    n0 = (1 + i + nxn * (j + nyn * k)).transpose(2, 1, 0).reshape(-1).astype(np.int64)
    sx, sy, sz = 1, nxn, nxn * nyn
    h = np.stack([n0, n0 + sx, n0 + sx + sy, n0 + sy, n0 + sz, n0 + sx + sz, n0 + sx + sy + sz, n0 + sy + sz], axis=1)
    return np.ascontiguousarray(_lid_lookup(S)(h.reshape(-1)).reshape(-1, 8).astype(np.int32))


def reference_edges(S, hexes=None):
    """The rank's edge list in the reference's sense: S.edge_nodes (every edge that touches an owned node:
    the graph needs them) followed by the edges of the rank's own elements between two shared-not-owned
    nodes.  Returns (edge_nodes, owner, send, recv): owner[e] = the rank that owns edge e (the lowest
    rank whose elements hold it; -1: no element of this rank holds it, it only shapes the graph), and the
    share lists of nalu_gpu_share_plan_create as {neighbour rank: local edge indices}, ascending global
    (L, R) on both sides."""
    hexes = own_hex_nodes(S) if hexes is None else hexes
    g = S.gids
    M = int(g.max()) + 1
    have = g[S.edge_nodes[:, 0]].astype(np.int64) * M + g[S.edge_nodes[:, 1]]
    pairs = np.concatenate([hexes[:, [a, b]] for a, b in HEX_EDGES]).astype(np.int64)
    flip = g[pairs[:, 0]] > g[pairs[:, 1]]
    pairs[flip] = pairs[flip][:, ::-1]
    key = g[pairs[:, 0]] * M + g[pairs[:, 1]]
    ukey, first = np.unique(key, return_index=True)
    extra = ~np.isin(ukey, have)
    en = np.concatenate([S.edge_nodes, pairs[first[extra]].astype(np.int32)])
    # sharers of an edge: the parts whose node range holds both ends, per dimension
    rng = _dim_ranges(S)
    px, py, pz = _parts(S)
    idx = (S.i.astype(np.int64), S.j.astype(np.int64), S.k.astype(np.int64))
    holds = []  # per dimension: (n_edges, parts) bool
    for d in range(3):
        a, b = idx[d][en[:, 0]], idx[d][en[:, 1]]
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        holds.append(np.stack([(lo >= r0) & (hi <= r1) for r0, r1 in rng[d]], axis=1))
    rank_holds = np.zeros((len(en), px * py * pz), bool)
    for q in range(px * py * pz):
        rank_holds[:, q] = holds[0][:, q % px] & holds[1][:, (q // px) % py] & holds[2][:, q // (px * py)]
    owner = np.where(rank_holds[:, S.rank], rank_holds.argmax(axis=1), -1).astype(np.int32)
    ekey = g[en[:, 0]].astype(np.int64) * M + g[en[:, 1]]
    send, recv = {}, {}
    for q in range(px * py * pz):
        if q == S.rank:
            continue
        mine = np.nonzero((owner == S.rank) & rank_holds[:, q])[0]
        theirs = np.nonzero(owner == q)[0]
        if len(mine):
            send[q] = mine[np.argsort(ekey[mine])].astype(np.int32)
        if len(theirs):
            recv[q] = theirs[np.argsort(ekey[theirs])].astype(np.int32)
    return np.ascontiguousarray(en), owner, send, recv


def share_plan_arrays(S, send, recv):
    """{rank: ids} dictionaries -> the (offsets, ids) arrays of nalu_gpu_share_plan_create, one list per
    halo neighbour of S in its order"""
    so, si, ro, ri = [0], [], [0], []
    for q in [int(v) for v in S.neighbors]:
        si.append(send.get(q, np.zeros(0, np.int32)))
        ri.append(recv.get(q, np.zeros(0, np.int32)))
        so.append(so[-1] + len(si[-1]))
        ro.append(ro[-1] + len(ri[-1]))
    assert set(send) | set(recv) <= set(int(v) for v in S.neighbors), "an edge is shared with a rank that is no halo neighbour"
    cat = lambda l: np.concatenate(l).astype(np.int32) if l else np.zeros(0, np.int32)
    return np.array(so, np.int32), cat(si), np.array(ro, np.int32), cat(ri)
