from .synth import lattice_mesh, m10k_batch, stress_triangles, perspective, translation, random_pose  # noqa: F401
