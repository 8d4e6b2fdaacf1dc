import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from shacira_amd.wisp.models.prob_models import BitEstimator
from shacira_amd.wisp.models.latent_decoders import LatentDecoder
dev = torch.device("cuda:0")
T, ld, F = 6098925, 2, 2
be = BitEstimator(ld, num_layers=2).to(dev)
dec = LatentDecoder(ld, F, "none", "sq", True, ldec_std=0.1).to(dev)
lat = (torch.rand(T, ld, device=dev) * 8 - 4).requires_grad_(True)
noise = torch.rand(T, ld, device=dev) - 0.5
gy = torch.randn(T, F, device=dev)
for _ in range(10):
    t = be.total_bits(lat, noise); t.backward()
    y = dec(lat); y.backward(gy)
torch.cuda.synchronize()
dec.use_sga = True
dec.temperature = 0.5
for _ in range(10):
    y = dec(lat); y.backward(gy)
torch.cuda.synchronize()
