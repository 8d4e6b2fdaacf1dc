"""CPU oracle for the SHACIRA hash-grid / latent path.

TEST INFRASTRUCTURE ONLY. Nothing under ``shacira_amd/`` imports this package; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may, and only as the checker.

* ``oracle.hashgrid_c``     -- ctypes binding of ``hashgrid_oracle.c`` (scalar C restatement of the CUDA kernels).
* ``oracle.hashgrid_torch`` -- pure-PyTorch restatement of the same kernels ("reference pure-PyTorch CPU path").
* ``oracle.latent``         -- torch restatement of LatentDecoder / BitEstimator / ent_loss / size.
"""
