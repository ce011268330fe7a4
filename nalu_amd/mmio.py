"""MatrixMarket dumps in the layout of TpetraLinearSystem::writeToFile /
writeSolutionToFile (src/TpetraLinearSystem.C:1901-1924, 2002-2019):

    <base>-O-<count>.mm.<p_size>    coordinate real general, 1-based GLOBAL row/column ids
    <base>-O-<count>.rhs.<p_size>   array real general, n x 1
    <base>-O-<count>.sln.<p_size>   array real general, n x 1

(Tpetra::MatrixMarket::Writer::writeSparseFile / writeDenseFile are Trilinos code, not in the
tree; the files follow the published MatrixMarket format with 17 significant digits, which is
what any reader -- Tpetra's, scipy.io.mmread -- needs.)  This is the only wire format
through which a real-Trilinos run can be compared with this library.
The reference's second, per-rank set of files is a Teuchos::describe() text dump and is not
reproduced; with more than one rank every rank writes its owned rows to
`<name>.<p_rank>` in the same coordinate format."""
import numpy as np


def file_names(base, count, p_size, p_rank, use_owned=True):
    tag = "O-" if use_owned else "G-"
    sfx = f".{p_size}" + (f".{p_rank}" if p_size > 1 else "")
    return {k: f"{base}-{tag}{count}.{k}{sfx}" for k in ("mm", "rhs", "sln")}


def write_sparse(path, n_global_rows, n_global_cols, row_gids, rowptr, col_gids, vals,
                 name="", description=""):
    """rows `row_gids[i]` (1-based global ids) with entries col_gids[rowptr[i]:rowptr[i+1]]"""
    rowptr = np.asarray(rowptr, np.int64)
    nnz = int(rowptr[-1])
    rows = np.repeat(np.asarray(row_gids, np.int64), np.diff(rowptr))
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real general\n")
        if name:
            f.write(f"% {name}\n")
        for line in description.splitlines():
            f.write(f"% {line}\n")
        f.write(f"{n_global_rows} {n_global_cols} {nnz}\n")
        np.savetxt(f, np.column_stack([rows, np.asarray(col_gids, np.int64)[:nnz],
                                       np.asarray(vals, np.float64)[:nnz]]),
                   fmt=("%d", "%d", "%.17e"))


def write_dense(path, vec):
    v = np.asarray(vec, np.float64).ravel()
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix array real general\n")
        f.write(f"{len(v)} 1\n")
        np.savetxt(f, v, fmt="%.17e")


def read_sparse(path):
    """-> (n_rows, n_cols, rows, cols, vals) of a `coordinate real general` file, 1-based ids
    as stored (the order of the entries in the file is kept)."""
    with open(path) as f:
        head = f.readline().split()
        if [w.lower() for w in head[:5]] != ["%%matrixmarket", "matrix", "coordinate", "real", "general"]:
            raise ValueError(f"{path}: not a 'matrix coordinate real general' MatrixMarket file")
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        nr, nc, nnz = (int(t) for t in line.split())
        data = np.loadtxt(f, dtype=np.float64, ndmin=2) if nnz else np.zeros((0, 3))
    if data.shape != (nnz, 3):
        raise ValueError(f"{path}: expected {nnz} entries, found {data.shape[0]}")
    return nr, nc, data[:, 0].astype(np.int64), data[:, 1].astype(np.int64), data[:, 2].copy()


def read_dense(path):
    """-> the n x 1 vector of an `array real general` file"""
    with open(path) as f:
        head = f.readline().split()
        if [w.lower() for w in head[:5]] != ["%%matrixmarket", "matrix", "array", "real", "general"]:
            raise ValueError(f"{path}: not a 'matrix array real general' MatrixMarket file")
        line = f.readline()
        while line.startswith("%"):
            line = f.readline()
        n, m = (int(t) for t in line.split())
        v = np.loadtxt(f, dtype=np.float64, ndmin=1) if n * m else np.zeros(0)
    if v.size != n * m:
        raise ValueError(f"{path}: expected {n * m} values, found {v.size}")
    return v
