from .synth import (lattice_mesh, m10k_batch, stress_triangles, perspective, translation, random_pose,  # noqa: F401
                    dense_batch, big_mesh_batch, small_rotation)
