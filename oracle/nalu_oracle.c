/*
 * nalu_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY (see nalu_oracle.h).
 * Plain C restatement of the reference algorithm; each function cites the
 * reference file:line (relative to /root/reference) that it follows.
 */
#include "nalu_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP /* only the multi-core CPU-baseline build (liboracle_omp.so) */
#include <omp.h>
#endif

/* ================================================================== mesh */

void orc_box_coords(int nxn, int nyn, int nzn, double lx, double ly, double lz, double *xyz)
{
  const double hx = nxn > 1 ? lx / (nxn - 1) : 0.0;
  const double hy = nyn > 1 ? ly / (nyn - 1) : 0.0;
  const double hz = nzn > 1 ? lz / (nzn - 1) : 0.0;
  int64_t n = 0;
  for (int k = 0; k < nzn; ++k)
    for (int j = 0; j < nyn; ++j)
      for (int i = 0; i < nxn; ++i, ++n) {
        xyz[3 * n + 0] = i * hx;
        xyz[3 * n + 1] = j * hy;
        xyz[3 * n + 2] = k * hz;
      }
}

/* unit_tests/UnitTestUtils.C:41-75: Lcg a=1103515245 c=12345 m=2147483647,
 * value = 2*(prev/m)-1, coords += perturbSize*lcg().  The reference walks owned nodes in
 * STK bucket order (not reproducible here); the build walks interior nodes in id order. */
void orc_box_perturb(int nxn, int nyn, int nzn, double *xyz, uint32_t seed, double amp)
{
  const uint32_t a = 1103515245u, c = 12345u, m = 2147483647u;
  uint32_t prev = seed;
  for (int k = 1; k < nzn - 1; ++k)
    for (int j = 1; j < nyn - 1; ++j)
      for (int i = 1; i < nxn - 1; ++i) {
        const int64_t n = i + (int64_t)nxn * (j + (int64_t)nyn * k);
        for (int d = 0; d < 3; ++d) {
          prev = (uint32_t)(((uint64_t)a * (uint64_t)prev + c) % m);
          xyz[3 * n + d] += amp * (2. * (prev / (double)m) - 1.);
        }
      }
}

void orc_box_hexes(int nxn, int nyn, int nzn, int *hex_nodes)
{
  int64_t e = 0;
  for (int k = 0; k < nzn - 1; ++k)
    for (int j = 0; j < nyn - 1; ++j)
      for (int i = 0; i < nxn - 1; ++i, ++e) {
        const int n0 = i + nxn * (j + nyn * k);
        const int sx = 1, sy = nxn, sz = nxn * nyn;
        int *h = hex_nodes + 8 * e;
        h[0] = n0;
        h[1] = n0 + sx;
        h[2] = n0 + sx + sy;
        h[3] = n0 + sy;
        h[4] = n0 + sz;
        h[5] = n0 + sx + sz;
        h[6] = n0 + sx + sy + sz;
        h[7] = n0 + sy + sz;
      }
}

int64_t orc_box_edges(int nxn, int nyn, int nzn, int *edge_nodes)
{
  int64_t e = 0;
  for (int k = 0; k < nzn; ++k)
    for (int j = 0; j < nyn; ++j)
      for (int i = 0; i < nxn; ++i) {
        const int n0 = i + nxn * (j + nyn * k);
        if (i < nxn - 1) { edge_nodes[2 * e] = n0; edge_nodes[2 * e + 1] = n0 + 1; ++e; }
        if (j < nyn - 1) { edge_nodes[2 * e] = n0; edge_nodes[2 * e + 1] = n0 + nxn; ++e; }
        if (k < nzn - 1) { edge_nodes[2 * e] = n0; edge_nodes[2 * e + 1] = n0 + nxn * nyn; ++e; }
      }
  return e;
}

/* ============================================================== geometry */

/* src/master_element/CalculateQuadArea.F:33-70 */
void orc_quad_area_by_triangle_facets(const double ac[4][3], double area[3])
{
  /* triangularFacetTable(3,*) third entries: 2,3,4,1 (1-based) */
  static const int third[4] = {1, 2, 3, 0};
  double xmid[3], r1[3], r2[3];
  for (int k = 0; k < 3; ++k) xmid[k] = 0.25 * (ac[0][k] + ac[1][k] + ac[2][k] + ac[3][k]);
  for (int k = 0; k < 3; ++k) { area[k] = 0.0; r2[k] = ac[0][k] - xmid[k]; }
  for (int t = 0; t < 4; ++t) {
    const int iq = third[t];
    for (int k = 0; k < 3; ++k) r1[k] = r2[k];
    for (int k = 0; k < 3; ++k) r2[k] = ac[iq][k] - xmid[k];
    area[0] = area[0] + r1[1] * r2[2] - r2[1] * r1[2];
    area[1] = area[1] + r1[2] * r2[0] - r2[2] * r1[0];
    area[2] = area[2] + r1[0] * r2[1] - r2[0] * r1[1];
  }
  for (int k = 0; k < 3; ++k) area[k] = 0.5 * area[k];
}

/* src/master_element/CalculateHexVolume.F:27-98 + CalculatePolyhedralVolume.F:41-73 */
double orc_hex_volume_by_triangle_facets(const double vc[8][3])
{
  static const int tri[24][3] = {
      {0, 8, 1},  {8, 2, 1},  {3, 2, 8},  {3, 8, 0},  {6, 9, 5},  {7, 9, 6},
      {4, 9, 7},  {4, 5, 9},  {10, 0, 1}, {5, 10, 1}, {4, 10, 5}, {4, 0, 10},
      {7, 6, 11}, {6, 2, 11}, {2, 3, 11}, {3, 7, 11}, {6, 12, 2}, {5, 12, 6},
      {5, 1, 12}, {1, 2, 12}, {0, 4, 13}, {4, 7, 13}, {7, 3, 13}, {3, 0, 13}};
  double c[14][3];
  for (int j = 0; j < 8; ++j)
    for (int k = 0; k < 3; ++k) c[j][k] = vc[j][k];
  for (int k = 0; k < 3; ++k) {
    c[8][k] = 0.25 * (vc[0][k] + vc[1][k] + vc[2][k] + vc[3][k]);
    c[9][k] = 0.25 * (vc[4][k] + vc[5][k] + vc[6][k] + vc[7][k]);
    c[10][k] = 0.25 * (vc[0][k] + vc[1][k] + vc[5][k] + vc[4][k]);
    c[11][k] = 0.25 * (vc[3][k] + vc[2][k] + vc[6][k] + vc[7][k]);
    c[12][k] = 0.25 * (vc[1][k] + vc[2][k] + vc[6][k] + vc[5][k]);
    c[13][k] = 0.25 * (vc[0][k] + vc[3][k] + vc[7][k] + vc[4][k]);
  }
  double volume = 0.0;
  for (int t = 0; t < 24; ++t) {
    const int ip = tri[t][0], iq = tri[t][1], ir = tri[t][2];
    double xf[3];
    for (int k = 0; k < 3; ++k) xf[k] = c[ip][k] + c[iq][k] + c[ir][k];
    volume = volume +
             xf[0] * ((c[iq][1] - c[ip][1]) * (c[ir][2] - c[ip][2]) -
                      (c[ir][1] - c[ip][1]) * (c[iq][2] - c[ip][2])) -
             xf[1] * ((c[iq][0] - c[ip][0]) * (c[ir][2] - c[ip][2]) -
                      (c[ir][0] - c[ip][0]) * (c[iq][2] - c[ip][2])) +
             xf[2] * ((c[iq][0] - c[ip][0]) * (c[ir][1] - c[ip][1]) -
                      (c[ir][0] - c[ip][0]) * (c[iq][1] - c[ip][1]));
  }
  return volume / 18.0;
}

/* 27-point subdivision shared by hex_scs_det / hex_scv_det
 * (src/master_element/MasterElementWork.F:61-210 and :283-432); index = F77 index - 1 */
static void subdivide_hex(const double c[8][3], double p[27][3])
{
  for (int j = 0; j < 8; ++j)
    for (int k = 0; k < 3; ++k) p[j][k] = c[j][k];
  for (int k = 0; k < 3; ++k) {
    p[8][k] = 0.5 * (c[0][k] + c[1][k]);
    p[9][k] = 0.5 * (c[1][k] + c[2][k]);
    p[10][k] = 0.5 * (c[2][k] + c[3][k]);
    p[11][k] = 0.5 * (c[3][k] + c[0][k]);
    p[12][k] = 0.25 * (c[0][k] + c[1][k] + c[2][k] + c[3][k]);
    p[13][k] = 0.5 * (c[4][k] + c[5][k]);
    p[14][k] = 0.5 * (c[5][k] + c[6][k]);
    p[15][k] = 0.5 * (c[6][k] + c[7][k]);
    p[16][k] = 0.5 * (c[7][k] + c[4][k]);
    p[17][k] = 0.25 * (c[4][k] + c[5][k] + c[6][k] + c[7][k]);
    p[18][k] = 0.5 * (c[1][k] + c[5][k]);
    p[19][k] = 0.5 * (c[0][k] + c[4][k]);
    p[20][k] = 0.25 * (c[0][k] + c[1][k] + c[4][k] + c[5][k]);
    p[21][k] = 0.5 * (c[3][k] + c[7][k]);
    p[22][k] = 0.5 * (c[2][k] + c[6][k]);
    p[23][k] = 0.25 * (c[2][k] + c[3][k] + c[6][k] + c[7][k]);
    p[24][k] = 0.25 * (c[1][k] + c[2][k] + c[5][k] + c[6][k]);
    p[25][k] = 0.25 * (c[0][k] + c[3][k] + c[4][k] + c[7][k]);
    p[26][k] = 0.125 * (c[0][k] + c[1][k] + c[2][k] + c[3][k] + c[4][k] + c[5][k] + c[6][k] +
                        c[7][k]);
  }
}

/* src/master_element/MasterElementWork.F:41-54,212-226 (table is 1-based there) */
void orc_hex_scs_det(const double cordel[8][3], double area_vec[12][3])
{
  static const int tab[12][4] = {{21, 9, 13, 27},  {25, 10, 13, 27}, {11, 13, 27, 24},
                                 {12, 26, 27, 13}, {14, 21, 27, 18}, {18, 15, 25, 27},
                                 {18, 16, 24, 27}, {17, 18, 27, 26}, {20, 21, 27, 26},
                                 {21, 19, 25, 27}, {23, 24, 27, 25}, {22, 26, 27, 24}};
  double p[27][3], s[4][3];
  subdivide_hex(cordel, p);
  for (int ics = 0; ics < 12; ++ics) {
    for (int n = 0; n < 4; ++n)
      for (int k = 0; k < 3; ++k) s[n][k] = p[tab[ics][n] - 1][k];
    orc_quad_area_by_triangle_facets(s, area_vec[ics]);
  }
}

/* src/master_element/MasterElementWork.F:271-280,434-452 (table is 0-based there) */
void orc_hex_scv_det(const double cordel[8][3], double vol[8])
{
  static const int tab[8][8] = {
      {0, 8, 12, 11, 19, 20, 26, 25},  {8, 1, 9, 12, 20, 18, 24, 26},
      {12, 9, 2, 10, 26, 24, 22, 23},  {11, 12, 10, 3, 25, 26, 23, 21},
      {19, 20, 26, 25, 4, 13, 17, 16}, {20, 18, 24, 26, 13, 5, 14, 17},
      {26, 24, 22, 23, 17, 14, 6, 15}, {25, 26, 23, 21, 16, 17, 15, 7}};
  double p[27][3], s[8][3];
  subdivide_hex(cordel, p);
  for (int icv = 0; icv < 8; ++icv) {
    for (int n = 0; n < 8; ++n)
      for (int k = 0; k < 3; ++k) s[n][k] = p[tab[icv][n]][k];
    vol[icv] = orc_hex_volume_by_triangle_facets(s);
  }
}

/* binary search of (L,R) in the sorted edge list */
static int64_t find_edge(int64_t nedges, const int *edge_nodes, int L, int R)
{
  int64_t lo = 0, hi = nedges - 1;
  while (lo <= hi) {
    const int64_t mid = (lo + hi) / 2;
    const int a = edge_nodes[2 * mid], b = edge_nodes[2 * mid + 1];
    if (a == L && b == R) return mid;
    if (a < L || (a == L && b < R)) lo = mid + 1;
    else hi = mid - 1;
  }
  return -1;
}

/* src/ComputeGeometryInteriorAlgorithm.C:78-129 (dual volume, ipNodeMap = identity,
 * src/master_element/Hex8CVFEM.C:84-85) and :134-216 (edge area with the L-node sign
 * convention; lrscv_ from src/master_element/Hex8CVFEM.C:263-276). */
int orc_compute_geometry(int64_t nelem, const int *hex_nodes, const double *xyz,
                         int64_t nnodes, int64_t nedges, const int *edge_nodes,
                         double *dual_volume, double *edge_area)
{
  static const int lrscv[24] = {0, 1, 1, 2, 2, 3, 0, 3, 4, 5, 5, 6,
                                6, 7, 4, 7, 0, 4, 1, 5, 2, 6, 3, 7};
  for (int64_t n = 0; n < nnodes; ++n) dual_volume[n] = 0.0;
  for (int64_t e = 0; e < 3 * nedges; ++e) edge_area[e] = 0.0;
  double c[8][3], vol[8], av[12][3];
  for (int64_t el = 0; el < nelem; ++el) {
    const int *h = hex_nodes + 8 * el;
    for (int ni = 0; ni < 8; ++ni)
      for (int j = 0; j < 3; ++j) c[ni][j] = xyz[3 * (int64_t)h[ni] + j];
    orc_hex_scv_det(c, vol);
    for (int ip = 0; ip < 8; ++ip) dual_volume[h[ip]] += vol[ip];
  }
  for (int64_t el = 0; el < nelem; ++el) {
    const int *h = hex_nodes + 8 * el;
    for (int ni = 0; ni < 8; ++ni)
      for (int j = 0; j < 3; ++j) c[ni][j] = xyz[3 * (int64_t)h[ni] + j];
    orc_hex_scs_det(c, av);
    for (int ne = 0; ne < 12; ++ne) {
      const int a = h[lrscv[2 * ne]], b = h[lrscv[2 * ne + 1]];
      const int L = a < b ? a : b, R = a < b ? b : a;
      const int64_t e = find_edge(nedges, edge_nodes, L, R);
      if (e < 0) return 1;
      /* sign: +1 if the element's L node is the edge's node 0 (:199-207) */
      const double sign = (a == L) ? 1.0 : -1.0;
      for (int j = 0; j < 3; ++j) edge_area[3 * e + j] += av[ne][j] * sign;
    }
  }
  return 0;
}

/* ===================================================== LocalGraphArrays */

/* include/LinearSolver.h:69-77,95-107 */
orc_graph *orc_graph_create(int64_t nrows, const size_t *rowLengths)
{
  orc_graph *g = (orc_graph *)calloc(1, sizeof(orc_graph));
  g->nrows = nrows;
  g->rowPointers = (size_t *)malloc(sizeof(size_t) * (size_t)(nrows + 1));
  size_t nnz = 0;
  for (int64_t i = 0; i < nrows; ++i) {
    g->rowPointers[i] = nnz;
    nnz += rowLengths[i];
  }
  g->rowPointers[nrows] = nnz;
  g->colIndices = (int *)malloc(sizeof(int) * (nnz ? nnz : 1));
  for (size_t i = 0; i < nnz; ++i) g->colIndices[i] = ORC_INVALID;
  return g;
}

void orc_graph_destroy(orc_graph *g)
{
  if (!g) return;
  free(g->rowPointers);
  free(g->colIndices);
  free(g);
}

size_t orc_graph_row_length(const orc_graph *g, int64_t row)
{
  return g->rowPointers[row + 1] - g->rowPointers[row];
}

/* include/LinearSolver.h:81-93 (insertIndices) and :115-123 (insert) */
void orc_graph_insert_indices(orc_graph *g, int64_t localRow, size_t numInds, const int *inds,
                              int numDof)
{
  int *row = g->colIndices + g->rowPointers[localRow];
  const size_t rowLen = orc_graph_row_length(g, localRow);
  int *rowEnd = row;
  while (rowEnd < row + rowLen && *rowEnd != ORC_INVALID) ++rowEnd; /* std::find */
  for (size_t i = 0; i < numInds; ++i) {
    /* std::lower_bound(row, rowEnd, inds[i]) */
    int *lo = row, *hi = rowEnd;
    while (lo < hi) {
      int *mid = lo + (hi - lo) / 2;
      if (*mid < inds[i]) lo = mid + 1;
      else hi = mid;
    }
    int *insertPoint = lo;
    if (insertPoint <= rowEnd && (insertPoint == row + rowLen || *insertPoint != inds[i])) {
      /* insert(ind, numDof, insertPoint, rowEnd+numDof), :115-123: shift the tail up by
       * numDof, then write ind..ind+numDof-1.  (The reference's shift loop also reads
       * ptr-numDof below insertPoint for the first numDof-1 slots, values that the fill
       * overwrites; those reads are skipped here.) */
      int *newEnd = rowEnd + numDof;
      for (int *ptr = newEnd - 1; ptr != insertPoint; --ptr)
        if (ptr - numDof >= insertPoint) *ptr = *(ptr - numDof);
      int *ip = insertPoint;
      for (int d = 0; d < numDof; ++d) *ip++ = inds[i] + d;
      rowEnd += numDof;
    }
  }
}

/* src/TpetraLinearSystem.C:1169-1201 */
void orc_graph_remove_invalid(orc_graph *g)
{
  const int64_t n = g->nrows;
  size_t *newLen = (size_t *)malloc(sizeof(size_t) * (size_t)(n ? n : 1));
  size_t newNnz = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int *row = g->colIndices + g->rowPointers[i];
    const int rowLen = (int)orc_graph_row_length(g, i);
    newLen[i] = 0;
    for (int j = rowLen - 1; j >= 0; --j)
      if (row[j] != ORC_INVALID) { newLen[i] = (size_t)j + 1; break; }
    newNnz += newLen[i];
  }
  if (newNnz < g->rowPointers[n]) {
    int *newCols = (int *)malloc(sizeof(int) * (newNnz ? newNnz : 1));
    size_t idx = 0;
    for (int64_t i = 0; i < n; ++i) {
      const int *row = g->colIndices + g->rowPointers[i];
      for (size_t j = 0; j < newLen[i]; ++j) newCols[idx++] = row[j];
    }
    free(g->colIndices);
    g->colIndices = newCols;
    size_t nnz = 0;
    for (int64_t i = 0; i < n; ++i) { g->rowPointers[i] = nnz; nnz += newLen[i]; }
    g->rowPointers[n] = nnz;
  }
  free(newLen);
}

/* ======================================================== linear system */

orc_linsys *orc_linsys_create(int numDof, int nOwnedNodes, int nSharedNodes)
{
  orc_linsys *ls = (orc_linsys *)calloc(1, sizeof(orc_linsys));
  ls->numDof = numDof;
  ls->nOwnedNodes = nOwnedNodes;
  ls->nSharedNodes = nSharedNodes;
  /* src/TpetraLinearSystem.C:217-218 */
  ls->maxOwnedRowId = nOwnedNodes * numDof;
  ls->maxSharedNotOwnedRowId = (nOwnedNodes + nSharedNodes) * numDof;
  const int nn = nOwnedNodes + nSharedNodes;
  ls->conn = (int **)calloc((size_t)(nn ? nn : 1), sizeof(int *));
  ls->connLen = (int *)calloc((size_t)(nn ? nn : 1), sizeof(int));
  ls->connCap = (int *)calloc((size_t)(nn ? nn : 1), sizeof(int));
  return ls;
}

void orc_linsys_destroy(orc_linsys *ls)
{
  if (!ls) return;
  const int nn = ls->nOwnedNodes + ls->nSharedNodes;
  if (ls->conn)
    for (int i = 0; i < nn; ++i) free(ls->conn[i]);
  free(ls->conn);
  free(ls->connLen);
  free(ls->connCap);
  orc_graph_destroy(ls->ownedGraph);
  orc_graph_destroy(ls->sharedGraph);
  free(ls->ownedVals);
  free(ls->sharedVals);
  free(ls->ownedRhs);
  free(ls->sharedRhs);
  free(ls->sln);
  free(ls);
}

/* src/TpetraLinearSystem.C:333-353 */
static void insert_connection(orc_linsys *ls, int a, int b)
{
  const int nn = ls->nOwnedNodes + ls->nSharedNodes;
  if (a >= nn) return; /* ghost row: entityToLID_ is "2000000000" there; never a row */
  int *v = ls->conn[a];
  for (int i = 0; i < ls->connLen[a]; ++i)
    if (v[i] == b) return;
  if (ls->connLen[a] == ls->connCap[a]) {
    ls->connCap[a] = ls->connCap[a] ? 2 * ls->connCap[a] : 8; /* reserve(8), :301 */
    ls->conn[a] = (int *)realloc(ls->conn[a], sizeof(int) * (size_t)ls->connCap[a]);
  }
  ls->conn[a][ls->connLen[a]++] = b;
}

/* src/TpetraLinearSystem.C:355-371; local ids are assigned in ascending-gid order inside
 * each of the owned / shared groups, and the comparison there is on global ids.  The
 * caller passes gid-ordered local ids only within a group, so compare through gid order
 * = local-id order is NOT valid across groups; the oracle therefore takes an optional
 * gid array (NULL => local id order == gid order, the single-rank case). */
static const int64_t *g_gids = NULL;
void orc_set_gids(const int64_t *gids) { g_gids = gids; }

void orc_add_connections(orc_linsys *ls, int64_t n_ent, int npe, const int *ent_nodes)
{
  for (int64_t e = 0; e < n_ent; ++e) {
    const int *ent = ent_nodes + (int64_t)npe * e;
    for (int a = 0; a < npe; ++a) {
      const int ea = ent[a];
      insert_connection(ls, ea, ea);
      for (int b = a + 1; b < npe; ++b) {
        const int eb = ent[b];
        const int a_then_b = g_gids ? (g_gids[ea] < g_gids[eb]) : (ea < eb);
        const int emin = a_then_b ? ea : eb;
        const int emax = a_then_b ? eb : ea;
        insert_connection(ls, emin, emax);
      }
    }
  }
}

static int cmp_int(const void *a, const void *b)
{
  const int x = *(const int *)a, y = *(const int *)b;
  return (x > y) - (x < y);
}

/* finalizeLinearSystem, src/TpetraLinearSystem.C:1232-1329, single-rank part:
 * sort_connections (:658-664), compute_graph_row_lengths (:802-865, upper bounds),
 * insert_graph_connections (:877-917), fill_in_extra_dof_rows_per_node (:1203-1222),
 * remove_invalid_indices (:1169-1201, applied to the owned graph only, :1287). */
void orc_finalize(orc_linsys *ls)
{
  const int nd = ls->numDof;
  const int nO = ls->nOwnedNodes, nS = ls->nSharedNodes, nn = nO + nS;
  for (int i = 0; i < nn; ++i) qsort(ls->conn[i], (size_t)ls->connLen[i], sizeof(int), cmp_int);

  size_t *lenO = (size_t *)calloc((size_t)(nO * nd + 1), sizeof(size_t));
  size_t *lenS = (size_t *)calloc((size_t)(nS * nd + 1), sizeof(size_t));
  /* add_to_length (:666-676) */
  for (int a = 0; a < nn; ++a) {
    const int nc = ls->connLen[a];
    for (int d = 0; d < nd; ++d) {
      if (a < nO) lenO[a * nd + d] += (size_t)nd * nc;
      else lenS[(a - nO) * nd + d] += (size_t)nd * nc;
    }
    for (int ii = 0; ii < nc; ++ii) {
      const int b = ls->conn[a][ii];
      if (b == a || b >= nn) continue;
      for (int d = 0; d < nd; ++d) {
        if (b < nO) lenO[b * nd + d] += (size_t)nd;
        else lenS[(b - nO) * nd + d] += (size_t)nd;
      }
    }
  }
  ls->ownedGraph = orc_graph_create((int64_t)nO * nd, lenO);
  ls->sharedGraph = orc_graph_create((int64_t)nS * nd, lenS);

  int *colsB = (int *)malloc(sizeof(int) * 4096);
  int capB = 4096;
  for (int a = 0; a < nn; ++a) {
    const int nc = ls->connLen[a];
    if (nc > capB) { capB = nc; colsB = (int *)realloc(colsB, sizeof(int) * (size_t)capB); }
    for (int j = 0; j < nc; ++j) colsB[j] = ls->conn[a][j] * nd; /* entityToColLID_ */
    const int colA = a * nd;
    {
      orc_graph *g = a < nO ? ls->ownedGraph : ls->sharedGraph;
      const int row = a < nO ? a * nd : (a - nO) * nd;
      orc_graph_insert_indices(g, row, (size_t)nc, colsB, nd);
    }
    for (int j = 0; j < nc; ++j) {
      const int b = ls->conn[a][j];
      if (b == a || b >= nn) continue;
      orc_graph *g = b < nO ? ls->ownedGraph : ls->sharedGraph;
      const int row = b < nO ? b * nd : (b - nO) * nd;
      orc_graph_insert_indices(g, row, 1, &colA, nd);
    }
  }
  free(colsB);
  /* fill_in_extra_dof_rows_per_node (:1203-1222) */
  if (nd > 1) {
    orc_graph *gs[2] = {ls->ownedGraph, ls->sharedGraph};
    for (int q = 0; q < 2; ++q) {
      orc_graph *g = gs[q];
      for (int64_t i = 0; i < g->nrows; i += nd) {
        const int *row = g->colIndices + g->rowPointers[i];
        const int rowLen = (int)orc_graph_row_length(g, i);
        for (int d = 1; d < nd; ++d) {
          int *row_d = g->colIndices + g->rowPointers[i] + (size_t)rowLen * d;
          for (int j = 0; j < rowLen; ++j) row_d[j] = row[j];
        }
      }
    }
  }
  orc_graph_remove_invalid(ls->ownedGraph);
  /* Tpetra's fillComplete on a StaticProfile graph keeps only valid entries; trim the
   * shared graph too so that row views never see INVALID (no effect on matched sums). */
  orc_graph_remove_invalid(ls->sharedGraph);
  free(lenO);
  free(lenS);

  const size_t nnzO = ls->ownedGraph->rowPointers[ls->ownedGraph->nrows];
  const size_t nnzS = ls->sharedGraph->rowPointers[ls->sharedGraph->nrows];
  ls->ownedVals = (double *)calloc(nnzO ? nnzO : 1, sizeof(double));
  ls->sharedVals = (double *)calloc(nnzS ? nnzS : 1, sizeof(double));
  ls->ownedRhs = (double *)calloc((size_t)(nO * nd + 1), sizeof(double));
  ls->sharedRhs = (double *)calloc((size_t)(nS * nd + 1), sizeof(double));
  ls->sln = (double *)calloc((size_t)(nO * nd + 1), sizeof(double));
}

/* src/TpetraLinearSystem.C:1331-1348 */
void orc_zero_system(orc_linsys *ls)
{
  const size_t nnzO = ls->ownedGraph->rowPointers[ls->ownedGraph->nrows];
  const size_t nnzS = ls->sharedGraph->rowPointers[ls->sharedGraph->nrows];
  memset(ls->sharedVals, 0, sizeof(double) * nnzS);
  memset(ls->ownedVals, 0, sizeof(double) * nnzO);
  memset(ls->sharedRhs, 0, sizeof(double) * (size_t)(ls->nSharedNodes * ls->numDof));
  memset(ls->ownedRhs, 0, sizeof(double) * (size_t)(ls->nOwnedNodes * ls->numDof));
  memset(ls->sln, 0, sizeof(double) * (size_t)(ls->nOwnedNodes * ls->numDof));
}

/* src/TpetraLinearSystem.C:1353-1384 */
static void sum_into_row(const int *colidx, double *values, int length, int numCols,
                         const int *localIds, const int *sort_permutation,
                         const double *input_values)
{
  int offset = 0;
  for (int j = 0; j < numCols; ++j) {
    const int perm_index = sort_permutation[j];
    const int cur = localIds[j];
    while (offset < length && colidx[offset] != cur) ++offset;
    if (offset < length) values[offset] += input_values[perm_index];
  }
}

/* Tpetra::Details::shellSortKeysAndValues is in Trilinos (not in tree); any sort of the
 * keys that carries the values gives the same result for distinct keys. */
static void shell_sort_keys_and_values(int *keys, int *vals, int n)
{
  for (int gap = n / 2; gap > 0; gap /= 2)
    for (int i = gap; i < n; ++i) {
      const int k = keys[i], v = vals[i];
      int j = i;
      for (; j >= gap && keys[j - gap] > k; j -= gap) {
        keys[j] = keys[j - gap];
        vals[j] = vals[j - gap];
      }
      keys[j] = k;
      vals[j] = v;
    }
}

/* src/TpetraLinearSystem.C:1454-1507.  [rowLo, rowHi) restricts the rows that are summed:
 * the whole range for the reference behaviour, one thread's owned rows in the multi-core
 * CPU-baseline build (one owner-computes partition per core, like one MPI rank per core). */
static void sum_into_rows(orc_linsys *ls, int n_obj, const int *node_lids, const double *rhs,
                          const double *lhs, int rowLo, int rowHi)
{
  const int nd = ls->numDof;
  const int numRows = n_obj * nd;
  int scratchIds[64], sortPermutation[64];
  int *ids = scratchIds, *perm = sortPermutation;
  if (numRows > 64) {
    ids = (int *)malloc(sizeof(int) * (size_t)numRows);
    perm = (int *)malloc(sizeof(int) * (size_t)numRows);
  }
  for (int i = 0; i < n_obj; ++i)
    for (int d = 0; d < nd; ++d) ids[i * nd + d] = node_lids[i] * nd + d;
  for (int i = 0; i < numRows; ++i) perm[i] = i;
  shell_sort_keys_and_values(ids, perm, numRows);

  for (int r = 0; r < numRows; ++r) {
    const int i = perm[r] / nd;
    int rowLid = node_lids[i] * nd; /* entityToLID_ == entityToColLID_ for owned+shared */
    rowLid += perm[r] % nd;
    const int cur = perm[r];
    const double *cur_lhs = lhs + (size_t)cur * numRows;
    const double cur_rhs = rhs[cur];
    if (rowLid < rowLo || rowLid >= rowHi) continue;
    if (rowLid < ls->maxOwnedRowId) {
      const orc_graph *g = ls->ownedGraph;
      const size_t b = g->rowPointers[rowLid];
      sum_into_row(g->colIndices + b, ls->ownedVals + b, (int)orc_graph_row_length(g, rowLid),
                   numRows, ids, perm, cur_lhs);
      ls->ownedRhs[rowLid] += cur_rhs;
    } else if (rowLid < ls->maxSharedNotOwnedRowId) {
      const int actual = rowLid - ls->maxOwnedRowId;
      const orc_graph *g = ls->sharedGraph;
      const size_t b = g->rowPointers[actual];
      sum_into_row(g->colIndices + b, ls->sharedVals + b, (int)orc_graph_row_length(g, actual),
                   numRows, ids, perm, cur_lhs);
      ls->sharedRhs[actual] += cur_rhs;
    }
  }
  if (numRows > 64) { free(ids); free(perm); }
}

void orc_sum_into(orc_linsys *ls, int n_obj, const int *node_lids, const double *rhs,
                  const double *lhs)
{
  sum_into_rows(ls, n_obj, node_lids, rhs, lhs, 0, INT_MAX);
}

/* src/TpetraLinearSystem.C:1546-1583 */
void orc_apply_dirichlet(orc_linsys *ls, int64_t n, const int *node_lids,
                         const double *solution, const double *bcValues, unsigned beginPos,
                         unsigned endPos)
{
  const int nd = ls->numDof;
  for (int64_t k = 0; k < n; ++k) {
    const int localIdOffset = node_lids[k] * nd;
    for (unsigned d = beginPos; d < endPos; ++d) {
      const int localId = localIdOffset + (int)d;
      const int useOwned = localId < ls->maxOwnedRowId;
      const int actual = useOwned ? localId : localId - ls->maxOwnedRowId;
      const orc_graph *g = useOwned ? ls->ownedGraph : ls->sharedGraph;
      double *vals = useOwned ? ls->ownedVals : ls->sharedVals;
      const double diagonal_value = useOwned ? 1.0 : 0.0;
      const size_t b = g->rowPointers[actual];
      const size_t len = orc_graph_row_length(g, actual);
      for (size_t i = 0; i < len; ++i)
        vals[b + i] = (g->colIndices[b + i] == localId) ? diagonal_value : 0.0;
      double *rhs = useOwned ? ls->ownedRhs : ls->sharedRhs;
      rhs[actual] = useOwned ? (bcValues[(size_t)node_lids[k] * nd + d] -
                                solution[(size_t)node_lids[k] * nd + d])
                             : 0.0;
    }
  }
}

/* src/TpetraLinearSystem.C:1650-1683 */
void orc_reset_rows(orc_linsys *ls, int64_t n, const int *node_lids, unsigned beginPos,
                    unsigned endPos)
{
  const int nd = ls->numDof;
  for (int64_t k = 0; k < n; ++k) {
    for (unsigned d = beginPos; d < endPos; ++d) {
      const int localId = node_lids[k] * nd + (int)d;
      const int useOwned = localId < ls->maxOwnedRowId;
      const int actual = useOwned ? localId : localId - ls->maxOwnedRowId;
      const orc_graph *g = useOwned ? ls->ownedGraph : ls->sharedGraph;
      double *vals = useOwned ? ls->ownedVals : ls->sharedVals;
      const size_t b = g->rowPointers[actual];
      const size_t len = orc_graph_row_length(g, actual);
      for (size_t i = 0; i < len; ++i) vals[b + i] = 0.0;
      (useOwned ? ls->ownedRhs : ls->sharedRhs)[actual] = 0.0;
    }
  }
}

/* ========================================================= edge kernels */

double orc_peclet_classic(double A, double hf, double pe)
{
  (void)A; /* src/PecletFunction.C:74 hard-codes 5.0 and never reads A_ */
  const double modPeclet = hf * pe;
  return modPeclet * modPeclet / (5.0 + modPeclet * modPeclet);
}

double orc_peclet_tanh(double c1, double c2, double x)
{
  return 0.50 * (1.0 + tanh((x - c1) / c2));
}

double orc_van_leer(double dqm, double dqp, double small)
{
  return (2.0 * (dqm * dqp + fabs(dqm * dqp))) / ((dqm + dqp) * (dqm + dqp) + small);
}

/* src/AssembleContinuityEdgeSolverAlgorithm.C:141-203, one edge */
static void continuity_edge(const int *en, const double *av, const double *xyz,
                            const double *dpdx, const double *vel, const double *pressure,
                            const double *density, double projTimeScale, double nocFac,
                            double interpTogether, double *p_lhs, double *p_rhs,
                            double *tmdot_out)
{
  const int nodeL = en[0], nodeR = en[1];
  const double om_interpTogether = 1.0 - interpTogether;
  const double *coordL = xyz + 3 * (size_t)nodeL, *coordR = xyz + 3 * (size_t)nodeR;
  const double *GpdxL = dpdx + 3 * (size_t)nodeL, *GpdxR = dpdx + 3 * (size_t)nodeR;
  const double *vrtmL = vel + 3 * (size_t)nodeL, *vrtmR = vel + 3 * (size_t)nodeR;
  const double pressureL = pressure[nodeL], pressureR = pressure[nodeR];
  const double densityL = density[nodeL], densityR = density[nodeR];

  double axdx = 0.0, asq = 0.0;
  for (int j = 0; j < 3; ++j) {
    const double axj = av[j];
    const double dxj = coordR[j] - coordL[j];
    asq += axj * axj;
    axdx += axj * dxj;
  }
  const double inv_axdx = 1.0 / axdx;
  const double rhoIp = 0.5 * (densityR + densityL);

  double tmdot = -projTimeScale * (pressureR - pressureL) * asq * inv_axdx;
  for (int j = 0; j < 3; ++j) {
    const double axj = av[j];
    const double dxj = coordR[j] - coordL[j];
    const double kxj = axj - asq * inv_axdx * dxj;
    const double rhoUjIp = 0.5 * (densityR * vrtmR[j] + densityL * vrtmL[j]);
    const double ujIp = 0.5 * (vrtmR[j] + vrtmL[j]);
    const double GjIp = 0.5 * (GpdxR[j] + GpdxL[j]);
    tmdot += (interpTogether * rhoUjIp + om_interpTogether * rhoIp * ujIp +
              projTimeScale * GjIp) * axj -
             projTimeScale * kxj * GjIp * nocFac;
  }
  const double lhsfac = -asq * inv_axdx;
  if (p_lhs) {
    p_lhs[0] = -lhsfac;
    p_lhs[1] = +lhsfac;
    p_rhs[0] = -tmdot / projTimeScale;
    p_lhs[2] = +lhsfac;
    p_lhs[3] = -lhsfac;
    p_rhs[1] = tmdot / projTimeScale;
  }
  if (tmdot_out) *tmdot_out = tmdot;
}

#ifdef _OPENMP
/* CPU-baseline build only (-fopenmp).  One MPI rank of the reference owns its edges: it does not
 * scan the other ranks' edges.  So every thread gets, once per edge list, the list of the edges
 * that touch its contiguous block of rows ("owner computes": cut edges appear in two lists), in
 * ascending edge order; building the lists is set-up (the mesh partition), not assembly. */
static struct {
  const int *edges; int64_t nedges; int nrows, nt;
  int64_t *off;   /* nt + 1 */
  int64_t *list;  /* edge ids, thread after thread */
} g_buckets;

void orc_omp_prepare_edge_buckets(const orc_linsys *ls, int64_t nedges, const int *edge_nodes)
{
  const int nt = omp_get_max_threads(), nrows = ls->maxOwnedRowId;
  if (g_buckets.edges == edge_nodes && g_buckets.nedges == nedges && g_buckets.nrows == nrows &&
      g_buckets.nt == nt)
    return;
  free(g_buckets.off); free(g_buckets.list);
  g_buckets.edges = edge_nodes; g_buckets.nedges = nedges; g_buckets.nrows = nrows; g_buckets.nt = nt;
  g_buckets.off = (int64_t *)calloc((size_t)nt + 1, sizeof(int64_t));
  int64_t *cnt = (int64_t *)calloc((size_t)nt, sizeof(int64_t));
#pragma omp parallel
  {
    const int t = omp_get_thread_num();
    const int lo = (int)((int64_t)nrows * t / nt), hi = (int)((int64_t)nrows * (t + 1) / nt);
    int64_t c = 0;
    for (int64_t k = 0; k < nedges; ++k) {
      const int L = edge_nodes[2 * k], R = edge_nodes[2 * k + 1];
      c += (L >= lo && L < hi) || (R >= lo && R < hi);
    }
    cnt[t] = c;
  }
  for (int t = 0; t < nt; ++t) g_buckets.off[t + 1] = g_buckets.off[t] + cnt[t];
  g_buckets.list = (int64_t *)malloc(sizeof(int64_t) * (size_t)(g_buckets.off[nt] > 0 ? g_buckets.off[nt] : 1));
#pragma omp parallel
  {
    const int t = omp_get_thread_num();
    const int lo = (int)((int64_t)nrows * t / nt), hi = (int)((int64_t)nrows * (t + 1) / nt);
    int64_t *out = g_buckets.list + g_buckets.off[t];
    for (int64_t k = 0; k < nedges; ++k) {
      const int L = edge_nodes[2 * k], R = edge_nodes[2 * k + 1];
      if ((L >= lo && L < hi) || (R >= lo && R < hi)) *out++ = k;
    }
  }
  free(cnt);
}
#else
void orc_omp_prepare_edge_buckets(const orc_linsys *ls, int64_t nedges, const int *edge_nodes)
{
  (void)ls; (void)nedges; (void)edge_nodes; /* single-thread builds walk the edge list itself */
}
#endif

void orc_assemble_continuity_edges(orc_linsys *ls, int64_t nedges, const int *edge_nodes,
                                   const double *edge_area, const double *xyz,
                                   const double *dpdx, const double *vel,
                                   const double *pressure, const double *density, double dt,
                                   double gamma1, double nocFac, double interpTogether)
{
  const double projTimeScale = dt / gamma1;
#ifdef _OPENMP
  /* CPU-baseline build only (-fopenmp): every thread owns a contiguous block of rows and
   * assembles the edges that touch it, summing only its own rows. */
  orc_omp_prepare_edge_buckets(ls, nedges, edge_nodes); /* no-op when bench.py prepared them */
#pragma omp parallel
  {
    const int nt = g_buckets.nt, t = omp_get_thread_num();
    const int nrows = ls->maxOwnedRowId;
    const int lo = (int)((int64_t)nrows * t / nt), hi = (int)((int64_t)nrows * (t + 1) / nt);
    double lhs[4], rhs[2];
    if (t < nt)
      for (int64_t i = g_buckets.off[t]; i < g_buckets.off[t + 1]; ++i) {
        const int64_t k = g_buckets.list[i];
        continuity_edge(edge_nodes + 2 * k, edge_area + 3 * k, xyz, dpdx, vel, pressure, density,
                        projTimeScale, nocFac, interpTogether, lhs, rhs, NULL);
        sum_into_rows(ls, 2, edge_nodes + 2 * k, rhs, lhs, lo, hi);
      }
  }
#else
  double lhs[4], rhs[2];
  for (int64_t k = 0; k < nedges; ++k) {
    continuity_edge(edge_nodes + 2 * k, edge_area + 3 * k, xyz, dpdx, vel, pressure, density,
                    projTimeScale, nocFac, interpTogether, lhs, rhs, NULL);
    orc_sum_into(ls, 2, edge_nodes + 2 * k, rhs, lhs); /* apply_coeff, :205 */
  }
#endif
}

void orc_continuity_edge_coeffs(int64_t nedges, const int *edge_nodes,
                                const double *edge_area, const double *xyz,
                                const double *dpdx, const double *vel,
                                const double *pressure, const double *density, double dt,
                                double gamma1, double nocFac, double interpTogether,
                                double *lhs, double *rhs)
{
  const double projTimeScale = dt / gamma1;
  for (int64_t k = 0; k < nedges; ++k)
    continuity_edge(edge_nodes + 2 * k, edge_area + 3 * k, xyz, dpdx, vel, pressure, density,
                    projTimeScale, nocFac, interpTogether, lhs + 4 * k, rhs + 2 * k, NULL);
}

void orc_compute_mdot_edges(int64_t nedges, const int *edge_nodes, const double *edge_area,
                            const double *xyz, const double *dpdx, const double *vel,
                            const double *pressure, const double *density, double dt,
                            double gamma1, double nocFac, double interpTogether,
                            double *mdot)
{
  /* src/ComputeMdotEdgeAlgorithm.C:145-171 is the same flux expression */
  const double projTimeScale = dt / gamma1;
  for (int64_t k = 0; k < nedges; ++k)
    continuity_edge(edge_nodes + 2 * k, edge_area + 3 * k, xyz, dpdx, vel, pressure, density,
                    projTimeScale, nocFac, interpTogether, NULL, NULL, mdot + k);
}

/* src/AssembleScalarEdgeSolverAlgorithm.C:150-309, one edge */
static void scalar_edge(const int *en, const double *av, double tmdot, const double *xyz,
                        const double *dqdx, const double *vel, const double *q,
                        const double *density, const double *diffFluxCoeff,
                        const orc_scalar_opts *o, double *p_lhs, double *p_rhs)
{
  const double small = 1.0e-16;
  const double alpha = o->alpha, alphaUpw = o->alphaUpw, hoUpwind = o->hoUpwind;
  const double om_alpha = 1.0 - alpha, om_alphaUpw = 1.0 - alphaUpw;
  for (int i = 0; i < 4; ++i) p_lhs[i] = 0.0;
  for (int i = 0; i < 2; ++i) p_rhs[i] = 0.0;
  const int nodeL = en[0], nodeR = en[1];
  const double *coordL = xyz + 3 * (size_t)nodeL, *coordR = xyz + 3 * (size_t)nodeR;
  const double *dqdxL = dqdx + 3 * (size_t)nodeL, *dqdxR = dqdx + 3 * (size_t)nodeR;
  const double *vrtmL = vel + 3 * (size_t)nodeL, *vrtmR = vel + 3 * (size_t)nodeR;
  const double qNp1L = q[nodeL], qNp1R = q[nodeR];
  const double densityL = density[nodeL], densityR = density[nodeR];
  const double diffFluxCoeffL = diffFluxCoeff[nodeL], diffFluxCoeffR = diffFluxCoeff[nodeR];

  double axdx = 0.0, asq = 0.0, udotx = 0.0;
  for (int j = 0; j < 3; ++j) {
    const double axj = av[j];
    const double dxj = coordR[j] - coordL[j];
    asq += axj * axj;
    axdx += axj * dxj;
    udotx += 0.5 * dxj * (vrtmL[j] + vrtmR[j]);
  }
  const double inv_axdx = 1.0 / axdx;
  const double viscIp = 0.5 * (diffFluxCoeffL + diffFluxCoeffR);
  const double diffIp = 0.5 * (diffFluxCoeffL / densityL + diffFluxCoeffR / densityR);
  const double pe = fabs(udotx) / (diffIp + small);
  const double pecfac = o->pecletKind == 0 ? orc_peclet_classic(o->pecA, o->pecHf, pe)
                                           : orc_peclet_tanh(o->pecC1, o->pecC2, pe);
  const double om_pecfac = 1.0 - pecfac;

  double dqL = 0.0, dqR = 0.0, nonOrth = 0.0;
  for (int j = 0; j < 3; ++j) {
    const double dxj = coordR[j] - coordL[j];
    dqL += 0.5 * dxj * dqdxL[j];
    dqR += 0.5 * dxj * dqdxR[j];
    const double axj = av[j];
    const double kxj = axj - asq * inv_axdx * dxj;
    const double GjIp = 0.5 * (dqdxL[j] + dqdxR[j]);
    nonOrth += -viscIp * kxj * GjIp;
  }
  double limitL = 1.0, limitR = 1.0;
  const double dq = qNp1R - qNp1L;
  if (o->useLimiter) {
    const double dqMl = 2.0 * 2.0 * dqL - dq;
    const double dqMr = 2.0 * 2.0 * dqR - dq;
    limitL = orc_van_leer(dqMl, dq, small);
    limitR = orc_van_leer(dqMr, dq, small);
  }
  const double qIpL = qNp1L + dqL * hoUpwind * limitL;
  const double qIpR = qNp1R - dqR * hoUpwind * limitR;

  double lhsfac = -viscIp * asq * inv_axdx;
  double diffFlux = lhsfac * (qNp1R - qNp1L) + nonOrth;
  p_lhs[0] = -lhsfac;
  p_lhs[1] = +lhsfac;
  p_rhs[0] = -diffFlux;
  p_lhs[2] = +lhsfac;
  p_lhs[3] = -lhsfac;
  p_rhs[1] = diffFlux;

  const double qIp = 0.5 * (qNp1L + qNp1R);
  const double qUpwind = (tmdot > 0) ? alphaUpw * qIpL + om_alphaUpw * qIp
                                     : alphaUpw * qIpR + om_alphaUpw * qIp;
  const double qHatL = alpha * qIpL + om_alpha * qIp;
  const double qHatR = alpha * qIpR + om_alpha * qIp;
  const double qCds = 0.5 * (qHatL + qHatR);
  const double aflux = tmdot * (pecfac * qUpwind + om_pecfac * qCds);

  double alhsfac = 0.5 * (tmdot + fabs(tmdot)) * pecfac * alphaUpw + 0.5 * alpha * om_pecfac * tmdot;
  p_lhs[0] += alhsfac;
  p_lhs[2] -= alhsfac;
  alhsfac = 0.5 * (tmdot - fabs(tmdot)) * pecfac * alphaUpw + 0.5 * alpha * om_pecfac * tmdot;
  p_lhs[3] -= alhsfac;
  p_lhs[1] += alhsfac;
  alhsfac = 0.5 * tmdot * (pecfac * om_alphaUpw + om_pecfac * om_alpha);
  p_lhs[0] += alhsfac;
  p_lhs[1] += alhsfac;
  p_lhs[2] -= alhsfac;
  p_lhs[3] -= alhsfac;
  p_rhs[0] -= aflux;
  p_rhs[1] += aflux;
}

void orc_assemble_scalar_edges(orc_linsys *ls, int64_t nedges, const int *edge_nodes,
                               const double *edge_area, const double *mdot,
                               const double *xyz, const double *dqdx, const double *vel,
                               const double *q, const double *density,
                               const double *diffFluxCoeff, const orc_scalar_opts *o)
{
#ifdef _OPENMP
  /* CPU-baseline build only: owner-computes edge lists per thread, as for the continuity loop */
  orc_omp_prepare_edge_buckets(ls, nedges, edge_nodes);
#pragma omp parallel
  {
    const int nt = g_buckets.nt, t = omp_get_thread_num();
    const int nrows = ls->maxOwnedRowId;
    const int lo = (int)((int64_t)nrows * t / nt), hi = (int)((int64_t)nrows * (t + 1) / nt);
    double lhs[4], rhs[2];
    if (t < nt)
      for (int64_t i = g_buckets.off[t]; i < g_buckets.off[t + 1]; ++i) {
        const int64_t k = g_buckets.list[i];
        scalar_edge(edge_nodes + 2 * k, edge_area + 3 * k, mdot[k], xyz, dqdx, vel, q, density,
                    diffFluxCoeff, o, lhs, rhs);
        sum_into_rows(ls, 2, edge_nodes + 2 * k, rhs, lhs, lo, hi);
      }
  }
#else
  double lhs[4], rhs[2];
  for (int64_t k = 0; k < nedges; ++k) {
    scalar_edge(edge_nodes + 2 * k, edge_area + 3 * k, mdot[k], xyz, dqdx, vel, q, density,
                diffFluxCoeff, o, lhs, rhs);
    orc_sum_into(ls, 2, edge_nodes + 2 * k, rhs, lhs); /* apply_coeff, :311 */
  }
#endif
}

void orc_scalar_edge_coeffs(int64_t nedges, const int *edge_nodes, const double *edge_area,
                            const double *mdot, const double *xyz, const double *dqdx,
                            const double *vel, const double *q, const double *density,
                            const double *diffFluxCoeff, const orc_scalar_opts *o,
                            double *lhs, double *rhs)
{
  for (int64_t k = 0; k < nedges; ++k)
    scalar_edge(edge_nodes + 2 * k, edge_area + 3 * k, mdot[k], xyz, dqdx, vel, q, density,
                diffFluxCoeff, o, lhs + 4 * k, rhs + 2 * k);
}

/* src/AssembleNodalGradEdgeAlgorithm.C:87-114 */
void orc_nodal_grad_edges(int64_t nedges, const int *edge_nodes, const double *edge_area,
                          const double *dual_volume, const double *q, double *dqdx)
{
  for (int64_t k = 0; k < nedges; ++k) {
    const int nodeL = edge_nodes[2 * k], nodeR = edge_nodes[2 * k + 1];
    double *gradQL = dqdx + 3 * (size_t)nodeL, *gradQR = dqdx + 3 * (size_t)nodeR;
    const double volL = dual_volume[nodeL], volR = dual_volume[nodeR];
    const double qL = q[nodeL], qR = q[nodeR];
    const double qip = 0.5 * (qL + qR);
    const double invVolL = 1.0 / volL, invVolR = 1.0 / volR;
    for (int j = 0; j < 3; ++j) {
      const double aj = edge_area[3 * k + j];
      const double ajQip = aj * qip;
      gradQL[j] += ajQip * invVolL;
      gradQR[j] -= ajQip * invVolR;
    }
  }
}

/* src/AssembleNodeSolverAlgorithm.C:84-110 (1x1 sumInto per owned node) with
 * src/ScalarMassBackwardEulerNodeSuppAlg.C:72-79 as the supplemental algorithm */
void orc_assemble_scalar_mass_be(orc_linsys *ls, int64_t nnodes, const double *qN,
                                 const double *qNp1, const double *rhoN,
                                 const double *rhoNp1, const double *dualVolume, double dt)
{
  for (int64_t n = 0; n < nnodes; ++n) {
    double lhs = 0.0, rhs = 0.0;
    const double lhsTime = rhoNp1[n] * dualVolume[n] / dt;
    rhs -= (rhoNp1[n] * qNp1[n] - qN[n] * rhoN[n]) * dualVolume[n] / dt;
    lhs += lhsTime;
    const int lid = (int)n;
    orc_sum_into(ls, 1, &lid, &rhs, &lhs);
  }
}

/* src/FieldFunctions.C:24-58: y = alpha x + beta y */
void orc_field_axpby(int64_t n, double alpha, const double *x, double beta, double *y)
{
  for (int64_t k = 0; k < n; ++k) y[k] = alpha * x[k] + beta * y[k];
}

/* src/LowMachEquationSystem.C:842-853: u^{n+1} -= (dt/gamma1)/rho (Gp_new - Gp_old), AoS
 * [nnodes][3] like the STK fields; projected == NULL projects every node */
void orc_project_nodal_velocity(int64_t nnodes, double dt, double gamma1, const double *dpdx_new,
                                const double *dpdx_old, const double *density,
                                const unsigned char *projected, double *velocity)
{
  const double projTimeScale = dt / gamma1;
  for (int64_t k = 0; k < nnodes; ++k) {
    if (projected && !projected[k]) continue;
    const double fac = projTimeScale / density[k];
    const size_t offSet = (size_t)k * 3;
    for (int j = 0; j < 3; ++j) {
      const double gdpx = dpdx_new[offSet + j] - dpdx_old[offSet + j];
      velocity[offSet + j] -= fac * gdpx;
    }
  }
}

/* src/MixtureFractionEquationSystem.C:942-1013 (update_and_clip): z += zTmp, the un-filtered value
 * kept in zUF, z clipped to [0 - deltaZ, 1 + deltaZ]; info = {numClip[0], numClip[1], minZ, maxZ} of
 * this rank (the all_reduce of :993-1006 is the caller's) */
void orc_update_and_clip(int64_t nnodes, const double *zTmp, double *mixFrac, double *mixFracUF,
                         double deltaZ, double *info)
{
  const double lowBound = 0.0 - deltaZ;
  const double highBound = 1.0 + deltaZ;
  double numClip[2] = {0.0, 0.0};
  double minZ = +1.0e16;
  double maxZ = -1.0e16;
  for (int64_t k = 0; k < nnodes; ++k) {
    double mixFracNp1 = mixFrac[k] + zTmp[k];
    if (mixFracUF) mixFracUF[k] = mixFracNp1;
    if (mixFracNp1 < lowBound) {
      minZ = mixFracNp1 < minZ ? mixFracNp1 : minZ;
      mixFracNp1 = lowBound;
      numClip[0] += 1.0;
    } else if (mixFracNp1 > highBound) {
      maxZ = mixFracNp1 > maxZ ? mixFracNp1 : maxZ;
      mixFracNp1 = highBound;
      numClip[1] += 1.0;
    }
    mixFrac[k] = mixFracNp1;
  }
  if (info) { info[0] = numClip[0]; info[1] = numClip[1]; info[2] = minZ; info[3] = maxZ; }
}

/* =============================================================== solver */

void orc_spmv(const orc_linsys *ls, const double *x, double *y)
{
  const orc_graph *g = ls->ownedGraph;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < g->nrows; ++i) {
    double s = 0.0;
    for (size_t k = g->rowPointers[i]; k < g->rowPointers[i + 1]; ++k)
      s += ls->ownedVals[k] * x[g->colIndices[k]];
    y[i] = s;
  }
}

/* Dot products.  Tpetra's dot / norm2 (src/LinearSolver.C:172, src/TpetraLinearSystem.C:1756; Kokkos
 * parallel_reduce underneath) fix no summation order; the default here is the plain sequential sum.
 * orc_set_dot_tree(1) selects another deterministic order: the fixed reduction tree of the device
 * library's dot products (nalu_amd/csrc/solve.hip: grid-stride per-thread sums over ceil(n / 1024)
 * workgroups of 256 threads, a shuffle tree over the 64 lanes of a wave, the four wave sums in order, then
 * one 1024-thread pass over the workgroup partials with the same lane tree and a tree over its 16 wave
 * sums).  The tests of BiCGStab and TFQMR use it: their short recurrences amplify the last-bit
 * differences between two summation orders by ~1e6 over a solve, so the two sides have to agree on ONE
 * order before their residual histories can be compared to 1e-10 (cg and gmres hold 1e-10 with either). */
static int g_dot_tree = 0;
void orc_set_dot_tree(int on) { g_dot_tree = on; }

#define ORC_TREE_MAX_BLOCKS 16384
static double lane_tree(double *v, int width)
{ /* v[l] += v[l + o] for o = width / 2 ... 1: __shfl_down trees (lanes beyond the width hold zero) */
  for (int o = width / 2; o > 0; o >>= 1)
    for (int l = 0; l < o; ++l) v[l] += v[l + o];
  return v[0];
}

static double dot_tree(const double *a, const double *b, int64_t n)
{
  int64_t vb = (n + 1023) / 1024;
  if (vb < 1) vb = 1;
  if (vb > ORC_TREE_MAX_BLOCKS) vb = ORC_TREE_MAX_BLOCKS;
  const int64_t T = vb * 256;
  double *part = (double *)malloc(sizeof(double) * (size_t)vb);
  for (int64_t B = 0; B < vb; ++B) {
    double s4[4];
    for (int w = 0; w < 4; ++w) {
      double v[64];
      for (int l = 0; l < 64; ++l) {
        double acc = 0.0;
        for (int64_t i = B * 256 + w * 64 + l; i < n; i += T) acc += a[i] * b[i];
        v[l] = acc;
      }
      s4[w] = lane_tree(v, 64);
    }
    part[B] = ((s4[0] + s4[1]) + s4[2]) + s4[3];
  }
  double s16[64];
  for (int w = 0; w < 64; ++w) s16[w] = 0.0;
  for (int w = 0; w < 16; ++w) {
    double v[64];
    for (int l = 0; l < 64; ++l) {
      const int64_t tid = w * 64 + l;
      double acc = 0.0;
      for (int64_t i0 = tid; i0 < vb; i0 += 8192)
        for (int u = 0; u < 8; ++u) acc += (i0 + u * 1024 < vb) ? part[i0 + u * 1024] : 0.0;
      v[l] = acc;
    }
    s16[w] = lane_tree(v, 64);
  }
  free(part);
  return lane_tree(s16, 16);
}

double orc_norm2(const double *v, int64_t n)
{
  if (g_dot_tree) return sqrt(dot_tree(v, v, n));
  double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
  for (int64_t i = 0; i < n; ++i) s += v[i] * v[i];
  return sqrt(s);
}

double orc_dot(const double *a, const double *b, int64_t n);
static double dot(const double *a, const double *b, int64_t n) { return orc_dot(a, b, n); }
double orc_dot(const double *a, const double *b, int64_t n)
{
  if (g_dot_tree) return dot_tree(a, b, n);
  double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
  for (int64_t i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* Ifpack2 RELAXATION applied to a zero starting solution, 1 sweep, damping 1
 * (src/LinearSolverConfig.C:71-85; Ifpack2 itself is not in the tree):
 *   Jacobi: z = D^-1 r ;  SGS: z = (D+U)^-1 D (D+L)^-1 r. */
/* Chebyshev polynomial preconditioner.  The reference itself offers none at the Ifpack2
 * level (src/LinearSolverConfig.C:71-99); BASELINE's north star asks for one and SURVEY.md
 * fixes its parameters from the MueLu smoother block reg_tests/xml/milestone.xml:5-12:
 * degree 2, eigenvalue ratio 20, 15 power iterations, zero starting solution.  Algorithm:
 * Ifpack2::Details::Chebyshev (ifpackApplyImpl + powerMethod; Trilinos, not in the tree,
 * restated from its published form: PARITY UNPINNED), on D^-1 A, boost factor 1.1.  Ifpack2
 * starts the power method from a random vector; here the start vector is a fixed hash of
 * the global node id so that runs are reproducible on any rank count. */
#define ORC_CHEB_DEGREE 2
#define ORC_CHEB_RATIO 20.0
#define ORC_CHEB_POWER_ITS 15
#define ORC_CHEB_BOOST 1.1
static double g_cheb_lambda_max = 0.0;

double orc_cheb_start_value(int64_t gid)
{
  uint32_t h = (uint32_t)gid * 2654435761u;
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
  return (double)h / 4294967296.0 * 2.0 - 1.0;
}

static double cheb_power_method(const orc_linsys *ls, const double *dinv)
{
  const int64_t n = ls->ownedGraph->nrows;
  double *x = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
  double *y = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
  for (int64_t i = 0; i < n; ++i) x[i] = orc_cheb_start_value(g_gids ? g_gids[i] : i + 1);
  double lambda = 0.0;
  double norm = orc_norm2(x, n);
  if (norm != 0.0) {
    for (int64_t i = 0; i < n; ++i) x[i] = x[i] / norm;
    for (int it = 0; it < ORC_CHEB_POWER_ITS; ++it) {
      orc_spmv(ls, x, y);
      for (int64_t i = 0; i < n; ++i) y[i] = dinv[i] * y[i];
      const double top = dot(y, x, n), bot = dot(x, x, n);
      lambda = top / bot;
      norm = orc_norm2(y, n);
      if (norm == 0.0) { lambda = 0.0; break; }
      for (int64_t i = 0; i < n; ++i) x[i] = y[i] / norm;
    }
  }
  free(x); free(y);
  return lambda;
}

static void cheb_apply(const orc_linsys *ls, const double *dinv, const double *r, double *x)
{
  const int64_t n = ls->ownedGraph->nrows;
  const double lambdaMax = g_cheb_lambda_max;
  const double alpha = lambdaMax / ORC_CHEB_RATIO, beta = ORC_CHEB_BOOST * lambdaMax;
  const double delta = 2.0 / (beta - alpha), theta = (beta + alpha) / 2.0, s1 = theta * delta;
  const double oneOverTheta = 1.0 / theta;
  double *w = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
  double *v = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
  for (int64_t i = 0; i < n; ++i) { w[i] = oneOverTheta * (dinv[i] * r[i]); x[i] = w[i]; }
  double rhok = 1.0 / s1;
  for (int deg = 1; deg < ORC_CHEB_DEGREE; ++deg) {
    orc_spmv(ls, x, v);
    const double rhokp1 = 1.0 / (2.0 * s1 - rhok);
    const double dtemp1 = rhokp1 * rhok, dtemp2 = 2.0 * rhokp1 * delta;
    rhok = rhokp1;
    for (int64_t i = 0; i < n; ++i) {
      w[i] = dtemp1 * w[i] + dtemp2 * (dinv[i] * (r[i] - v[i]));
      x[i] = x[i] + w[i];
    }
  }
  free(w); free(v);
}

/* Multicolour symmetric Gauss-Seidel ("mt_sgs", src/LinearSolverConfig.C:76-80: Ifpack2's
 * "MT Symmetric Gauss-Seidel", the KokkosKernels coloured sweep; Trilinos code, not in the
 * tree: PARITY UNPINNED, and KokkosKernels' colouring is its own).  Here: greedy colouring in
 * row order (colour of row i = smallest colour none of its lower-numbered owned neighbours
 * has), then one SGS sweep, zero starting solution, in (colour, row) order:
 *   forward, colours ascending:   t_i = (r_i - sum_{colour(c) < colour(i)} a_ic t_c) / d_i
 *   backward, colours descending: z_i = t_i - (sum_{colour(c) > colour(i)} a_ic z_c) / d_i
 * (zero starting solution: only the colours already visited contribute; rows of one colour are
 * not coupled; the sums in the row's column order).  Off-rank columns are ignored as in the
 * processor-local SGS. */
static int *g_mc_color = NULL, *g_mc_rows = NULL, *g_mc_off = NULL;
static int g_mc_ncolors = 0;
static int64_t g_mc_n = -1;

int orc_greedy_colors(const orc_linsys *ls, int *color)
{
  const orc_graph *g = ls->ownedGraph;
  const int64_t n = g->nrows;
  int ncolors = 0;
  for (int64_t i = 0; i < n; ++i) {
    uint64_t used = 0; /* up to 64 colours */
    for (size_t k = g->rowPointers[i]; k < g->rowPointers[i + 1]; ++k) {
      const int c = g->colIndices[k];
      if (c < i) used |= 1ull << color[c];
    }
    int col = 0;
    while (col < 63 && ((used >> col) & 1ull)) ++col;
    color[i] = col;
    if (col + 1 > ncolors) ncolors = col + 1;
  }
  return ncolors;
}

static void mc_sgs_setup(const orc_linsys *ls)
{
  const int64_t n = ls->ownedGraph->nrows;
  free(g_mc_color); free(g_mc_rows); free(g_mc_off);
  g_mc_color = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
  g_mc_rows = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
  g_mc_ncolors = orc_greedy_colors(ls, g_mc_color);
  g_mc_off = (int *)calloc((size_t)g_mc_ncolors + 2, sizeof(int));
  for (int64_t i = 0; i < n; ++i) ++g_mc_off[g_mc_color[i] + 1];
  for (int c = 0; c < g_mc_ncolors; ++c) g_mc_off[c + 1] += g_mc_off[c];
  int *cur = (int *)malloc(sizeof(int) * (size_t)(g_mc_ncolors + 1));
  memcpy(cur, g_mc_off, sizeof(int) * (size_t)(g_mc_ncolors + 1));
  for (int64_t i = 0; i < n; ++i) g_mc_rows[cur[g_mc_color[i]]++] = (int)i;
  free(cur);
  g_mc_n = n;
}

static void mc_sgs_apply(const orc_linsys *ls, const double *dinv, const double *r, double *z, double *t)
{
  const orc_graph *g = ls->ownedGraph;
  const int64_t n = g->nrows;
  for (int64_t i = 0; i < n; ++i) { t[i] = 0.0; z[i] = 0.0; }
  for (int col = 0; col < g_mc_ncolors; ++col)
    for (int q = g_mc_off[col]; q < g_mc_off[col + 1]; ++q) {
      const int i = g_mc_rows[q];
      double s = r[i];
      for (size_t k = g->rowPointers[i]; k < g->rowPointers[i + 1]; ++k) {
        const int c = g->colIndices[k];
        if (c != i && c < n && g_mc_color[c] < col) s -= ls->ownedVals[k] * t[c];
      }
      t[i] = s * dinv[i];
    }
  for (int col = g_mc_ncolors - 1; col >= 0; --col)
    for (int q = g_mc_off[col]; q < g_mc_off[col + 1]; ++q) {
      const int i = g_mc_rows[q];
      double s = 0.0;
      for (size_t k = g->rowPointers[i]; k < g->rowPointers[i + 1]; ++k) {
        const int c = g->colIndices[k];
        if (c != i && c < n && g_mc_color[c] > col) s += ls->ownedVals[k] * z[c];
      }
      z[i] = t[i] - s * dinv[i];
    }
}

static void apply_precond(const orc_linsys *ls, int precond, const double *dinv,
                          const double *r, double *z, double *tmp)
{
  const orc_graph *g = ls->ownedGraph;
  const int64_t n = g->nrows;
  if (precond == 3) {
    cheb_apply(ls, dinv, r, z);
  } else if (precond == 4) {
    mc_sgs_apply(ls, dinv, r, z, tmp);
  } else if (precond == 0) {
    memcpy(z, r, sizeof(double) * (size_t)n);
  } else if (precond == 1) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) z[i] = dinv[i] * r[i];
  } else {
    for (int64_t i = 0; i < n; ++i) { /* forward sweep */
      double s = r[i];
      for (size_t k = g->rowPointers[i]; k < g->rowPointers[i + 1]; ++k) {
        const int c = g->colIndices[k];
        if (c < i) s -= ls->ownedVals[k] * tmp[c];
      }
      tmp[i] = s * dinv[i];
    }
    for (int64_t i = n - 1; i >= 0; --i) { /* backward sweep */
      double s = 0.0;
      for (size_t k = g->rowPointers[i]; k < g->rowPointers[i + 1]; ++k) {
        const int c = g->colIndices[k];
        if (c > i && c < n) s += ls->ownedVals[k] * z[c];
      }
      z[i] = tmp[i] - s * dinv[i];
    }
  }
}

/* src/LinearSolver.C:181-230. Belos/Ifpack2 are third-party (Trilinos, not in tree);
 * algorithms restated from their published forms: PARITY UNPINNED. */
int orc_solve(orc_linsys *ls, const orc_solver_config *cfg, double *finalResidNrm,
              double *history, int history_cap, int *history_len)
{
  const orc_graph *g = ls->ownedGraph;
  const int64_t n = g->nrows;
  const double *b = ls->ownedRhs;
  double *x = ls->sln;
  int hl = 0, iters = 0;

  /* preconditioner_->compute() (:205): diagonal extraction */
  double *dinv = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    double d = 0.0;
    for (size_t k = g->rowPointers[i]; k < g->rowPointers[i + 1]; ++k)
      if (g->colIndices[k] == i) d = ls->ownedVals[k];
    dinv[i] = 1.0 / d;
  }
  double *tmp = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
  if (cfg->precond == 3) g_cheb_lambda_max = cheb_power_method(ls, dinv); /* compute() */
  if (cfg->precond == 4) mc_sgs_setup(ls);                               /* compute() */
  for (int64_t i = 0; i < n; ++i) x[i] = 0.0; /* zeroSystem: x0 = 0 */

  const double r0 = orc_norm2(b, n);
  if (history && hl < history_cap) history[hl++] = 1.0;

  if (r0 == 0.0) {
    /* nothing to do */
  } else if (cfg->method == 0) {
    /* preconditioned CG (Belos PseudoBlockCG): r = b, z = M^-1 r, p = z */
    double *r = (double *)malloc(sizeof(double) * (size_t)n);
    double *z = (double *)malloc(sizeof(double) * (size_t)n);
    double *p = (double *)malloc(sizeof(double) * (size_t)n);
    double *q = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(r, b, sizeof(double) * (size_t)n);
    apply_precond(ls, cfg->precond, dinv, r, z, tmp);
    memcpy(p, z, sizeof(double) * (size_t)n);
    double rz = dot(r, z, n);
    while (iters < cfg->max_iterations) {
      orc_spmv(ls, p, q);
      const double pAp = dot(p, q, n);
      const double alpha = rz / pAp;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) x[i] += alpha * p[i];
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) r[i] -= alpha * q[i];
      ++iters;
      const double rel = orc_norm2(r, n) / r0;
      if (history && hl < history_cap) history[hl++] = rel;
      if (rel <= cfg->tolerance) break;
      apply_precond(ls, cfg->precond, dinv, r, z, tmp);
      const double rz_new = dot(r, z, n);
      const double beta = rz_new / rz;
      rz = rz_new;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    }
    free(r); free(z); free(p); free(q);
  } else if (cfg->method == 2) {
    /* right-preconditioned BiCGStab (Belos "BICGSTAB", the method name biCgStab of 1 regression
     * input, e.g. reg_tests/test_files/dgNonConformal; Belos::BiCGStabIter is Trilinos code, not
     * in the tree: van der Vorst's algorithm restated, PARITY UNPINNED).  One iteration = two
     * operator applications; the status test sees the recurrence residual r. */
    double *r = (double *)malloc(sizeof(double) * (size_t)n), *rh = (double *)malloc(sizeof(double) * (size_t)n);
    double *p = (double *)calloc((size_t)n, sizeof(double)), *v = (double *)calloc((size_t)n, sizeof(double));
    double *sv = (double *)malloc(sizeof(double) * (size_t)n), *t = (double *)malloc(sizeof(double) * (size_t)n);
    double *y = (double *)malloc(sizeof(double) * (size_t)n), *z = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(r, b, sizeof(double) * (size_t)n);
    memcpy(rh, b, sizeof(double) * (size_t)n);
    double rho_old = 1.0, alpha = 1.0, omega = 1.0;
    while (iters < cfg->max_iterations) {
      const double rho_new = dot(rh, r, n);
      const double beta = (rho_new / rho_old) * (alpha / omega);
      for (int64_t i = 0; i < n; ++i) p[i] = r[i] + beta * (p[i] - omega * v[i]);
      apply_precond(ls, cfg->precond, dinv, p, y, tmp);
      orc_spmv(ls, y, v);
      alpha = rho_new / dot(rh, v, n);
      for (int64_t i = 0; i < n; ++i) sv[i] = r[i] - alpha * v[i];
      apply_precond(ls, cfg->precond, dinv, sv, z, tmp);
      orc_spmv(ls, z, t);
      omega = dot(t, sv, n) / dot(t, t, n);
      for (int64_t i = 0; i < n; ++i) x[i] = x[i] + alpha * y[i] + omega * z[i];
      for (int64_t i = 0; i < n; ++i) r[i] = sv[i] - omega * t[i];
      rho_old = rho_new;
      ++iters;
      const double rel = orc_norm2(r, n) / r0;
      if (history && hl < history_cap) history[hl++] = rel;
      if (rel <= cfg->tolerance) break;
    }
    free(r); free(rh); free(p); free(v); free(sv); free(t); free(y); free(z);
  } else if (cfg->method == 3) {
    /* right-preconditioned TFQMR (Belos "TFQMR", method name tfqmr of 3 regression inputs, e.g.
     * reg_tests/test_files/edgeHybridFluids/edgeHybridFluids.i:10; Belos::TFQMRIter is Trilinos
     * code: Freund's algorithm as in Saad, Iterative Methods, Alg. 7.8, on the operator A M^-1
     * with x = M^-1 (sum eta d) applied once at the end; one iteration = the even and the odd
     * half step; the status test sees the bound sqrt(2 iter + 1) tau.  PARITY UNPINNED). */
    double *w = (double *)malloc(sizeof(double) * (size_t)n), *u = (double *)malloc(sizeof(double) * (size_t)n);
    double *v = (double *)malloc(sizeof(double) * (size_t)n), *d = (double *)calloc((size_t)n, sizeof(double));
    double *au = (double *)malloc(sizeof(double) * (size_t)n), *rs = (double *)malloc(sizeof(double) * (size_t)n);
    double *ys = (double *)calloc((size_t)n, sizeof(double)), *z = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(w, b, sizeof(double) * (size_t)n);
    memcpy(u, b, sizeof(double) * (size_t)n);
    memcpy(rs, b, sizeof(double) * (size_t)n);
    apply_precond(ls, cfg->precond, dinv, u, z, tmp);
    orc_spmv(ls, z, v);                 /* v = A M^-1 u */
    memcpy(au, v, sizeof(double) * (size_t)n);
    double tau = r0, theta = 0.0, eta = 0.0, rho = dot(rs, w, n), alpha = 0.0;
    while (iters < cfg->max_iterations) {
      for (int half = 0; half < 2; ++half) {
        if (half == 0) {
          alpha = rho / dot(v, rs, n);
        } else {
          for (int64_t i = 0; i < n; ++i) u[i] = u[i] - alpha * v[i];
          apply_precond(ls, cfg->precond, dinv, u, z, tmp);
          orc_spmv(ls, z, au);          /* au = A M^-1 u */
        }
        for (int64_t i = 0; i < n; ++i) w[i] = w[i] - alpha * au[i];
        const double dfac = theta * theta * eta / alpha;
        for (int64_t i = 0; i < n; ++i) d[i] = u[i] + dfac * d[i];
        theta = orc_norm2(w, n) / tau;
        const double c = 1.0 / sqrt(1.0 + theta * theta);
        tau = tau * theta * c;
        eta = c * c * alpha;
        for (int64_t i = 0; i < n; ++i) ys[i] = ys[i] + eta * d[i];
        if (half == 1) {
          const double rho_new = dot(rs, w, n);
          const double beta = rho_new / rho;
          rho = rho_new;
          /* u = w + beta u ; v = A M^-1 u + beta (au + beta v) */
          for (int64_t i = 0; i < n; ++i) u[i] = w[i] + beta * u[i];
          for (int64_t i = 0; i < n; ++i) v[i] = beta * (au[i] + beta * v[i]);
          apply_precond(ls, cfg->precond, dinv, u, z, tmp);
          orc_spmv(ls, z, au);
          for (int64_t i = 0; i < n; ++i) v[i] = au[i] + v[i];
        }
      }
      ++iters;
      const double rel = sqrt(2.0 * iters + 1.0) * tau / r0;
      if (history && hl < history_cap) history[hl++] = rel;
      if (rel <= cfg->tolerance) break;
    }
    apply_precond(ls, cfg->precond, dinv, ys, z, tmp); /* x = M^-1 (sum eta d) */
    for (int64_t i = 0; i < n; ++i) x[i] = z[i];
    free(w); free(u); free(v); free(d); free(au); free(rs); free(ys); free(z);
  } else {
    /* restarted right-preconditioned GMRES(m) with ICGS (Belos BlockGmres, block size 1) */
    const int m = cfg->kspace;
    const int maxRestarts = cfg->max_iterations / m > 1 ? cfg->max_iterations / m : 1;
    double *V = (double *)malloc(sizeof(double) * (size_t)n * (size_t)(m + 1));
    double *H = (double *)calloc((size_t)(m + 1) * (size_t)m, sizeof(double));
    double *cs = (double *)calloc((size_t)m, sizeof(double));
    double *sn = (double *)calloc((size_t)m, sizeof(double));
    double *gv = (double *)calloc((size_t)(m + 1), sizeof(double));
    double *y = (double *)calloc((size_t)m, sizeof(double));
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    double *zv = (double *)malloc(sizeof(double) * (size_t)n);
    double *hc = (double *)calloc((size_t)(m + 1), sizeof(double));
    int converged = 0;
    for (int cycle = 0; cycle <= maxRestarts && !converged && iters < cfg->max_iterations;
         ++cycle) {
      /* r = b - A x */
      orc_spmv(ls, x, w);
      for (int64_t i = 0; i < n; ++i) w[i] = b[i] - w[i];
      const double beta = orc_norm2(w, n);
      if (beta / r0 <= cfg->tolerance) { converged = 1; break; }
      for (int64_t i = 0; i < n; ++i) V[i] = w[i] / beta;
      for (int i = 0; i <= m; ++i) gv[i] = 0.0;
      gv[0] = beta;
      int j = 0;
      for (; j < m && iters < cfg->max_iterations; ++j) {
        apply_precond(ls, cfg->precond, dinv, V + (size_t)j * n, zv, tmp);
        orc_spmv(ls, zv, w);
        for (int i = 0; i <= j; ++i) H[i + (size_t)j * (m + 1)] = 0.0;
        for (int pass = 0; pass < 2; ++pass) { /* ICGS: 2 classical Gram-Schmidt passes */
          for (int i = 0; i <= j; ++i) hc[i] = dot(V + (size_t)i * n, w, n);
          for (int i = 0; i <= j; ++i) {
            const double *vi = V + (size_t)i * n;
            const double h = hc[i];
            for (int64_t k = 0; k < n; ++k) w[k] -= h * vi[k];
            H[i + (size_t)j * (m + 1)] += h;
          }
        }
        const double hn = orc_norm2(w, n);
        H[(j + 1) + (size_t)j * (m + 1)] = hn;
        double *vn = V + (size_t)(j + 1) * n;
        for (int64_t k = 0; k < n; ++k) vn[k] = w[k] / hn;
        /* apply previous Givens rotations to column j, then form the new one */
        double *hj = H + (size_t)j * (m + 1);
        for (int i = 0; i < j; ++i) {
          const double t = cs[i] * hj[i] + sn[i] * hj[i + 1];
          hj[i + 1] = -sn[i] * hj[i] + cs[i] * hj[i + 1];
          hj[i] = t;
        }
        const double denom = sqrt(hj[j] * hj[j] + hj[j + 1] * hj[j + 1]);
        cs[j] = hj[j] / denom;
        sn[j] = hj[j + 1] / denom;
        hj[j] = denom;
        hj[j + 1] = 0.0;
        gv[j + 1] = -sn[j] * gv[j];
        gv[j] = cs[j] * gv[j];
        ++iters;
        const double rel = fabs(gv[j + 1]) / r0;
        if (history && hl < history_cap) history[hl++] = rel;
        if (rel <= cfg->tolerance) { converged = 1; ++j; break; }
      }
      /* solve the j x j upper-triangular system, x += M^-1 V y */
      for (int i = j - 1; i >= 0; --i) {
        double s = gv[i];
        for (int k = i + 1; k < j; ++k) s -= H[i + (size_t)k * (m + 1)] * y[k];
        y[i] = s / H[i + (size_t)i * (m + 1)];
      }
      for (int64_t k = 0; k < n; ++k) w[k] = 0.0;
      for (int i = 0; i < j; ++i) {
        const double *vi = V + (size_t)i * n;
        for (int64_t k = 0; k < n; ++k) w[k] += y[i] * vi[k];
      }
      apply_precond(ls, cfg->precond, dinv, w, zv, tmp);
      for (int64_t k = 0; k < n; ++k) x[k] += zv[k];
    }
    free(V); free(H); free(cs); free(sn); free(gv); free(y); free(w); free(zv); free(hc);
  }

  /* residual_norm (src/LinearSolver.C:153-178): ||A x - b||_2 */
  {
    double *res = (double *)malloc(sizeof(double) * (size_t)(n ? n : 1));
    orc_spmv(ls, x, res);
    for (int64_t i = 0; i < n; ++i) res[i] = res[i] - b[i];
    *finalResidNrm = orc_norm2(res, n);
    free(res);
  }
  if (history_len) *history_len = hl;
  free(dinv);
  free(tmp);
  return iters;
}
