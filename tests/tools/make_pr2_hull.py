"""Generates tests/golden/pr2_forearm_convex_vertices.npy from the reference's PR2 forearm convex mesh
(trajopt_common/data/pr2/meshes/forearm_v0/convex/forearm_convex.stla): the unique vertices of the file.  Run in the build container
(the GPU box has no /root/reference): python tests/tools/make_pr2_hull.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from trajopt_amd import meshes  # noqa: E402

src = "/root/reference/trajopt_common/data/pr2/meshes/forearm_v0/convex/forearm_convex.stla"
v = meshes.load_mesh_vertices(src)
np.save(os.path.join(ROOT, "tests", "golden", "pr2_forearm_convex_vertices.npy"), v)
print(len(v), "vertices, extent", v.min(axis=0), v.max(axis=0))
