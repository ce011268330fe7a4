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
