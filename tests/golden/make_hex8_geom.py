"""Generate tests/golden/hex8_geom.json from the REFERENCE's own F77 kernels.

Runs only in the build container: oracle/Makefile compiles
/root/reference/src/master_element/{MasterElementWork,CalculateQuadArea,CalculateHexVolume,
CalculatePolyhedralVolume,CalculateOctohedronVolume,CalculatePolyhedralArea}.F in place into
oracle/_ref/libnalu_ref_geom.so; this script calls hex_scs_det_/hex_scv_det_ (nelem = 1) on
a few hexes and stores inputs + outputs (data only).
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import pyoracle as po  # noqa: E402

r = po.ref_geom()
assert r is not None, "build oracle/_ref first (make -C oracle ref)"
unit = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0],
                 [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], float)
# the perturbed reference hex of unit_tests/UnitTestUtils.C:420-431 is built there from an
# LCG-perturbed unit cube; here: seeded perturbations of the unit cube and a sheared brick.
rng = np.random.default_rng(20141)
cases = [unit,
         unit * [2.0, 0.5, 1.25],
         unit + 0.125 * (2 * rng.random((8, 3)) - 1),
         unit + 0.25 * (2 * rng.random((8, 3)) - 1),
         (unit * [1.0, 2.0, 3.0]) @ np.array([[1, 0.2, 0.1], [0, 1, 0.3], [0, 0, 1.0]])]
out = []
for c in cases:
    out.append(dict(coords=c.tolist(), scs_area=po.ref_hex_scs_det(r, c).tolist(),
                    scv_vol=po.ref_hex_scv_det(r, c).tolist()))
with open(os.path.join(os.path.dirname(__file__), "hex8_geom.json"), "w") as f:
    json.dump(out, f, indent=0)
print("wrote", len(out), "cases")
