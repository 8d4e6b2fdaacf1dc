#!/usr/bin/env python3
"""Why some seeds of the config-B image fit end at 12.3 dB: fits one seed and reports, every 50 steps, the PSNR,
the share of hidden units of the decoder MLP that are dead on the whole batch (ReLU output 0 for every pixel), the spread of the
prediction over the pixels and of the decoded table.   usage: collapse_diag.py <seed> [height width steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from shacira_amd import harness
from shacira_amd.dist import shard_batch

seed = int(sys.argv[1]); H, W, steps = (int(a) for a in (sys.argv[2:5] if len(sys.argv) >= 5 else (256, 384, 400)))
dev = torch.device("cuda:0")
torch.manual_seed(seed)
grid, cdec, cent = harness.kodak_like_grid(num_lods=16)
nef = harness.NeuralImage(grid, hidden_dim=16, num_layers=1).to(dev)
img = torch.from_numpy(harness.make_test_image(H, W, seed)).reshape(-1, 3)
perm = torch.randperm(H * W, generator=torch.Generator().manual_seed(seed))
coords = harness.image_coords(H, W)[perm].contiguous().to(dev); rgb = img[perm].contiguous().to(dev)
fitter = harness.ImageFitter(nef, coords, rgb, steps, cdec, cent, world=1, global_pixels=H * W)
mlp = nef.decoder_color
for it in range(steps):
    out = fitter.step()
    if it % 50 == 0 or it == steps - 1:
        with torch.no_grad():
            table = grid.latent_dec(grid.codebook)
            feats = grid.interpolate(coords, 0)
            h = feats
            dead = []
            for layer in list(mlp.layers):
                h = torch.relu(layer(h))
                dead.append(float((h.max(dim=0).values <= 0).float().mean()))
            pred = nef.rgb(coords)
            print(f"step {it:4d} psnr {out[1]:6.2f} dead hidden units per layer {['%.2f' % d for d in dead]} "
                  f"pred std over pixels {float(pred.std(dim=0).mean()):.4f} (image {float(rgb.std(dim=0).mean()):.4f}) "
                  f"decoded table |max| {float(table.abs().max()):.3f} latent |max| {float(grid.codebook.abs().max()):.2f} "
                  f"div {float(grid.latent_dec.div.abs().max()):.3f}", flush=True)
