#!/usr/bin/env python3
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from shacira_amd import harness
gen = torch.Generator().manual_seed(0)
t0 = time.perf_counter()
for _ in range(20): pts = harness.ray_points(4096, 16, gen)
print("ray_points ms", (time.perf_counter() - t0) / 20 * 1e3)
dev = torch.device("cuda:0")
t0 = time.perf_counter()
for _ in range(20): x = pts.to(dev); torch.cuda.synchronize()
print("h2d ms", (time.perf_counter() - t0) / 20 * 1e3)
t0 = time.perf_counter()
for _ in range(20): y = harness.analytic_field(x); torch.cuda.synchronize()
print("field ms", (time.perf_counter() - t0) / 20 * 1e3)
from shacira_amd.wisp.models.grids import HashGrid
grid = HashGrid.from_geometric(feature_dim=2, num_lods=16, multiscale_type="cat", resolution_dim=3, feature_std=0.01, codebook_bitwidth=19, min_grid_res=16, max_grid_res=2048, blas_level=3)
nef = harness.NeuralField3D(grid).to(dev)
from shacira_amd.optim import FusedAdam
opt = FusedAdam([g for g in harness.param_groups(nef, lr=1e-3, grid_lr=1e-2) if g["params"]], eps=1e-15)
def step():
    opt.zero_grad(set_to_none=True)
    loss = (nef.rgb(x) - y).abs().sum() / (65536 * 3)
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print("train step ms", (time.perf_counter() - t0) / 20 * 1e3)
