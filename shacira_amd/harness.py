"""Minimal caller of the hash-grid path, restating what the reference's image trainer does per step, so that
"PSNR at a fixed step" can be measured without the reference's trainers/datasets (out of scope).

Restated from the reference (behaviour only):
  NeuralImage.rgb            wisp/models/nefs/image.py:127-154      feats = grid.interpolate(coords) -> decoder_color
  BasicDecoder               -> shacira_amd/wisp/models/decoders (mirror; fused HIP MLP on the GPU)
  ImageTrainer.step          wisp/trainers/image_trainer.py:269-359  MSE + lambda(epoch) * avg_bits, `div` normaliser
                             update at iterations where norm_every % iteration == 0 (sic), Adam step
  optimizer parameter groups wisp/trainers/base_trainer.py:206-266   by parameter-NAME substring
  schedules                  wisp/utils/schedulers.py:4-31           cosine entropy weight
  metric                     wisp/ops/image/metrics.py:39-58         clamped_psnr
Datasets are replaced by a procedurally generated image (no files, no network): `make_test_image`.
"""
import math

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from .wisp.models.decoders import BasicDecoder
from .wisp.models.grids import LatentGrid
from .wisp.models.latent_decoders import LatentDecoder
from .wisp.ops.image.metrics import clamped_psnr
from .wisp.utils.schedulers import DecayScheduler


def make_test_image(height=512, width=768, seed=0):
    """Deterministic RGB test image in [0,1]: smooth gradients + band-limited noise + hard edges. [H, W, 3] fp32."""
    rng = np.random.default_rng(seed)
    y, x = np.meshgrid(np.linspace(0, 1, height), np.linspace(0, 1, width), indexing="ij")
    img = np.stack([0.5 + 0.4 * np.sin(2 * np.pi * (1.5 * x + 0.3 * y)),
                    0.5 + 0.4 * np.cos(2 * np.pi * (0.7 * x - 1.1 * y)),
                    0.2 + 0.6 * x * (1 - y)], -1)
    for _ in range(24):  # band-limited texture: random low/mid-frequency plane waves
        fx, fy = rng.uniform(-24, 24, 2)
        ph, amp = rng.uniform(0, 2 * np.pi), rng.uniform(0.01, 0.05)
        img += amp * np.sin(2 * np.pi * (fx * x + fy * y) + ph)[..., None] * rng.uniform(0.3, 1.0, 3)
    for _ in range(6):   # edges: discs and rectangles
        cx, cy, r = rng.uniform(0.1, 0.9), rng.uniform(0.1, 0.9), rng.uniform(0.03, 0.15)
        m = ((x - cx) ** 2 + (y - cy) ** 2) < r * r
        img[m] = 0.6 * img[m] + 0.4 * rng.uniform(0, 1, 3)
        x0, y0, w, h = rng.uniform(0, 0.8), rng.uniform(0, 0.8), rng.uniform(0.05, 0.2), rng.uniform(0.05, 0.2)
        m = (x > x0) & (x < x0 + w) & (y > y0) & (y < y0 + h)
        img[m] = 0.5 * img[m] + 0.5 * rng.uniform(0, 1, 3)
    return np.clip(img, 0, 1).astype(np.float32)


def image_coords(height, width):
    """Pixel lattice in [-1,1)^2, axis 0 = image row (reference multi_image_dataset.py:148-153). [H*W, 2]"""
    rows = (torch.arange(height, dtype=torch.float32) / height - 0.5) * 2
    cols = (torch.arange(width, dtype=torch.float32) / width - 0.5) * 2
    rr, cc = torch.meshgrid(rows, cols, indexing="ij")
    return torch.stack([rr, cc], -1).reshape(-1, 2)


class NeuralImage(nn.Module):
    """coords [N,2] -> rgb [N,3]: grid lookup then `decoder_color` MLP (num_layers+1 hidden layers of hidden_dim)."""

    def __init__(self, grid, hidden_dim=16, num_layers=1):
        super().__init__()
        self.grid = grid
        feat = grid.feature_dim * grid.num_lods if grid.multiscale_type == "cat" else grid.feature_dim
        # image.py:107-116: BasicDecoder(in, 3, relu, bias=True, nn.Linear, num_layers + 1, hidden_dim, skip=[])
        self.decoder_color = BasicDecoder(feat, 3, torch.relu, True, nn.Linear, num_layers + 1, hidden_dim, [])

    def rgb(self, coords, lod_idx=None):
        if lod_idx is None:
            lod_idx = len(self.grid.active_lods) - 1
        feats = self.grid.interpolate(coords, lod_idx).reshape(coords.shape[0], -1)
        return self.decoder_color(feats)


def kodak_like_grid(num_lods=16, feature_dim=2, latent_dim=1, codebook_bitwidth=11, min_grid_res=16, max_grid_res=512,
                    use_sga=False, entropy_reg=1.0e-3, entropy_reg_end=1.0e-4, resolution_dim=2, blas_level=3):
    """Config B of SURVEY.md section 8: 16-level 2-D LatentGrid, F=2, quantisation + entropy model on."""
    cdec = dict(ldecode_enabled=True, ldecode_type="single", use_sga=use_sga, diff_sampling=True, use_shift=True,
                ldecode_matrix="sq", latent_dim=latent_dim, norm="max", norm_every=10, ldec_std=0.1, decay_period=0.9,
                temperature=0.1, num_layers_dec=0, hidden_dim_dec=0, activation="none", final_activation="none",
                clamp_weights=0.0, num_decoders=1, alpha_std=1.0)
    cent = dict(num_prob_layers=2, entropy_reg=entropy_reg, entropy_reg_end=entropy_reg_end,
                entropy_reg_sched="cosine", noise_freq=1)
    grid = LatentGrid.from_geometric(feature_dim=feature_dim, num_lods=num_lods, latent_dim=latent_dim,
                                     multiscale_type="cat", resolution_dim=resolution_dim, feature_std=0.1,
                                     feature_bias=0.0, codebook_bitwidth=codebook_bitwidth, min_grid_res=min_grid_res,
                                     max_grid_res=max_grid_res, blas_level=blas_level, init_grid="uniform",
                                     conf_latent_decoder=cdec, conf_entropy_reg=cent)
    return grid, cdec, cent


def param_groups(nef, lr=1e-3, grid_lr=0.02, ldec_lr=0.01, weight_decay=0.0, weight_decay_decoder=0.01):
    """The reference's grouping by parameter-name substring (base_trainer.py:206-266)."""
    groups = {k: [] for k in ("decoder", "grid", "latent_dec", "prob_models", "rest")}
    for name, p in nef.named_parameters():
        if "decoder" in name:
            groups["decoder"].append(p)
        elif "grid" in name:
            groups["latent_dec" if "latent_dec" in name else "prob_models" if "prob_model" in name else "grid"].append(p)
        else:
            groups["rest"].append(p)
    return [
        {"params": groups["decoder"], "lr": lr, "weight_decay": 0.0, "name": "decoder"},
        {"params": groups["grid"], "lr": grid_lr, "weight_decay": weight_decay, "name": "grid"},
        {"params": groups["latent_dec"], "lr": ldec_lr, "weight_decay": weight_decay_decoder, "name": "latent_dec"},
        {"params": groups["prob_models"], "lr": 1.0e-4, "weight_decay": weight_decay_decoder, "name": "prob_models"},
        {"params": groups["rest"], "lr": lr, "weight_decay": 0.0, "name": "rest"},
    ]


class ImageFitter:
    """One model, one static full-image batch per step (kodak.yaml: sample_mode 'full', batch_size 1)."""

    def __init__(self, nef, coords, rgb, total_steps, cdec, cent, lr=1e-3, grid_lr=0.02, ldec_lr=0.01,
                 weight_decay_decoder=0.01, world=1, global_pixels=None):
        """`coords`/`rgb` are this rank's shard; `global_pixels` the size of the whole batch (data parallel:
        parameters replicated, one all-reduce of the flat gradient buffer per step, shacira_amd/dist.py)."""
        self.nef, self.coords, self.rgb = nef, coords, rgb
        self.cdec, self.cent = cdec, cent
        self.total_steps = total_steps
        self.iteration = 0
        self.world = world
        self.global_pixels = global_pixels or coords.shape[0]
        groups = [g for g in param_groups(nef, lr, grid_lr, ldec_lr, 0.0, weight_decay_decoder) if g["params"]]
        if coords.is_cuda:   # row f1: one fused HIP kernel per parameter instead of torch's foreach chain
            from .optim import FusedAdam
            self.optimizer = FusedAdam(groups, eps=1e-8)
        else:
            self.optimizer = torch.optim.Adam(groups, eps=1e-8)
        self.bucket = None
        if world > 1:
            from .dist import FlatGradients
            self.bucket = FlatGradients([p for g in groups for p in g["params"]])
        self.lambda_sched = DecayScheduler(total_steps, cent["entropy_reg_sched"], cent["entropy_reg"],
                                           cent["entropy_reg_end"])
        # SGA warm-up of the shipped configs (base_trainer.py:155-157, image_trainer.py:132-137): temperature decays
        # exponentially from 1 to `temperature`, SGA is switched off after `decay_period` of the run
        self.temperature_sched = None
        if cdec.get("use_sga"):
            self.temperature_sched = DecayScheduler(total_steps, "exp", 1.0, cdec["temperature"],
                                                    {"temperature": cdec["temperature"],
                                                     "decay_period": cdec["decay_period"]})

    def _update_div(self):
        grid = self.nef.grid
        if (isinstance(grid, LatentGrid) and self.cdec["ldecode_enabled"] and isinstance(grid.latent_dec, LatentDecoder)
                and self.cdec["norm"] != "none" and self.cdec["norm_every"] % self.iteration == 0):  # (sic)
            lat = grid.codebook
            if self.cdec["norm"] == "max":
                grid.latent_dec.div.data = torch.max(torch.abs(lat.min(dim=0)[0]), torch.abs(lat.max(dim=0)[0]))
            elif self.cdec["norm"] == "std":
                grid.latent_dec.div.data = lat.std(dim=0)

    def step(self, read=True):
        """Returns (rgb_loss, clamped PSNR, avg bits per latent row) of this step's prediction. `read=False` leaves the
        step's statistics on the device (`read_last()` fetches them later) and returns None: the three host read-backs per
        step are device synchronisations -- the reference's trainers make them too (`.item()` on the losses,
        image_trainer.py:320-324) -- and with them the step costs host time PLUS kernel time instead of the larger of the two
        (config-B image fit: 0.84 -> ~0.6 ms per eager step)."""
        self.iteration += 1
        if self.bucket is not None:
            self.bucket.zero_()
        else:
            self.optimizer.zero_grad(set_to_none=True)
        with torch.no_grad():
            self._update_div()
        if self.temperature_sched is not None:
            dec = self.nef.grid.latent_dec
            dec.temperature = self.temperature_sched(self.iteration)
            if self.iteration / self.total_steps > self.cdec["decay_period"]:
                dec.use_sga = False
        pred = self.nef.rgb(self.coords)
        # mean over the GLOBAL batch: local sum / global element count (gradients are summed over ranks)
        sq_sum = ((pred - self.rgb) ** 2).sum()
        loss = sq_sum / (self.global_pixels * 3)
        lam = self.lambda_sched(self.iteration - 1)
        avg_bits = torch.zeros(())
        if self.cdec["ldecode_enabled"] and lam > 0:
            avg_bits, _ = self.nef.grid.ent_loss(self.iteration - 1, is_val=not self.nef.training)
            loss = loss + lam * avg_bits / self.world   # every rank evaluates the (table-only) entropy term
        loss.backward()
        hidden = []
        if self.bucket is not None:
            self.bucket.allreduce()
            hidden = self.bucket.hide_untouched()      # e.g. the entropy model while lambda == 0
        self.optimizer.step()
        if self.bucket is not None:
            self.bucket.restore(hidden)
        # the step's statistics (clamped PSNR: a dozen tensor ops of pure logging) are formed only when they are read
        self._last = (sq_sum.detach(), pred.detach(), avg_bits.detach() if torch.is_tensor(avg_bits) else avg_bits)
        return self.read_last() if read else None

    def read_last(self):
        """(rgb_loss, clamped PSNR, avg bits) of the last step: the statistics and their host read-back (a device
        synchronisation; with several ranks also one small all-reduce -- every rank must read the same steps)."""
        sq_sum, pred, avg_bits = self._last
        with torch.no_grad():
            q = lambda t: (torch.clamp(t, 0, 1) * 255).to(torch.uint8).float()
            if getattr(self, "_rgb_q", None) is None:
                self._rgb_q = q(self.rgb)              # (the target is the same every step)
            stats = torch.stack([sq_sum, ((q(pred) - self._rgb_q) ** 2).sum()]).double()
            if self.bucket is not None and dist.is_initialized():
                dist.all_reduce(stats)
        n = self.global_pixels * 3
        st = stats.tolist()                                                                  # one read-back for both sums
        rgb_loss = st[0] / n
        psnr = 20 * math.log10(255.0) - 10 * math.log10(max(st[1] / n, 1e-12))               # clamped_psnr, globally
        return rgb_loss, psnr, float(avg_bits)


def _require_device_temperature(latent_dec, who):
    """The graph-captured fitters keep the SGA temperature in ONE device float that the decode kernel reads. Only the single
    fused SGA decode (`shacira_latent_decode_sga_*_tdev`) has such an entry point: the per-row MLP decoder, the hierarchical and
    the multi decoders call `float(self.temperature)` -- a host read-back, which aborts a stream capture with an opaque HIP
    error or, taken outside the capture, freezes the temperature into the graph. Refuse those up front."""
    import torch as _torch
    from .wisp.models.latent_decoders.basic_latent_decoder import LatentDecoder
    ok = (isinstance(latent_dec, LatentDecoder) and type(latent_dec) is LatentDecoder
          and latent_dec._fusable(_torch.empty(1, latent_dec.latent_dim, device=latent_dec.div.device)))
    if not ok:
        raise ValueError(f"{who}: an SGA temperature schedule inside a captured step needs the single fused latent decoder "
                         f"(no hidden layers, identity activations, a latent_dim / feature_dim pair the fused kernel has); "
                         f"got {type(latent_dec).__name__}. Run this decoder with the eager fitter instead.")


class GraphedImageFitter(ImageFitter):
    """The same step recorded ONCE into a HIP graph and replayed: the per-step work of the image configs is a few
    hundred microseconds of GPU time behind ~150 launches, i.e. launch-bound when issued eagerly from Python.

    Differences from the eager fitter, all forced by capture: the entropy noise is drawn with the device generator
    (`grid.device_noise`), Adam keeps its step count on the device (`FusedAdam(capturable=True)`), the entropy weight
    lambda(step) is a device scalar refreshed before each replay, and PSNR is evaluated on request (`psnr()`), not
    every step; with the SGA warm-up of the shipped configs the temperature is a device float too. The `div` normaliser updates (iterations 1, 2, 5, 10 with norm_every = 10) happen in the eager
    warm-up steps that precede the capture."""

    def __init__(self, nef, coords, rgb, total_steps, cdec, cent, warmup_steps=11, **kw):
        from .optim import FusedAdam
        super().__init__(nef, coords, rgb, total_steps, cdec, cent, **kw)
        assert coords.is_cuda and self.world == 1, "graph capture: single-GPU path"
        groups = [{k: v for k, v in g.items() if k != "params"} | {"params": g["params"]}
                  for g in self.optimizer.param_groups]
        self.optimizer = FusedAdam(groups, eps=1e-8, capturable=True)
        nef.grid.device_noise = True
        self.lam = torch.zeros((), device=coords.device)
        self.stats = torch.zeros(2, dtype=torch.float64, device=coords.device)
        self.warmup_steps = warmup_steps
        self.graph = None
        # SGA warm-up (the shipped configs): the temperature decays every iteration -- it lives in one device float the decode
        # kernel reads (shacira_latent_decode_sga_*_tdev) and is refreshed before each replay; switching SGA off after
        # `decay_period` of the run changes the kernel, so the step is captured a second time then
        self.temperature = None
        if self.temperature_sched is not None:
            _require_device_temperature(nef.grid.latent_dec, "GraphedImageFitter")
            self.temperature = torch.ones(1, device=coords.device)
            nef.grid.latent_dec.temperature = self.temperature

    def _body(self):
        self.optimizer.zero_grad(set_to_none=True)
        pred = self.nef.rgb(self.coords)
        sq_sum = ((pred - self.rgb) ** 2).sum()
        avg_bits, _ = self.nef.grid.ent_loss(1, is_val=False)
        loss = sq_sum / (self.global_pixels * 3) + self.lam * avg_bits
        loss.backward()
        self.optimizer.step()
        with torch.no_grad():
            q = lambda t: (torch.clamp(t, 0, 1) * 255).to(torch.uint8).float()
            self.stats.copy_(torch.stack([sq_sum.detach(), ((q(pred) - q(self.rgb)) ** 2).sum()]).double())

    def step(self):
        self.iteration += 1
        self.lam.fill_(float(self.lambda_sched(self.iteration - 1)))
        if self.temperature is not None:
            dec = self.nef.grid.latent_dec
            self.temperature.fill_(float(self.temperature_sched(self.iteration)))
            if dec.use_sga and self.iteration / self.total_steps > self.cdec["decay_period"]:
                dec.use_sga = False
                self.graph = None                   # rounding instead of sampling from here on: re-capture
        if self.iteration <= self.warmup_steps:
            with torch.no_grad():
                self._update_div()
            self._body()
            return
        if self.graph is None:
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._body()
        else:
            self.graph.replay()

    def psnr(self):
        n = self.global_pixels * 3
        s = self.stats.tolist()
        return 20 * math.log10(255.0) - 10 * math.log10(max(s[1] / n, 1e-12)), s[0] / n


def fit_image(device, steps=1000, height=512, width=768, seed=0, num_lods=16, log_every=0, hidden_dim=16, rank=0,
              world=1, graphed=False, use_sga=False):
    """Fit the procedural image with the config-B LatentGrid; returns dict(psnr, rgb_loss, avg_bits, bpp, history).
    With world > 1 the (shuffled) pixel batch is sharded over ranks; results are identical on every rank."""
    from .dist import shard_batch
    torch.manual_seed(seed)
    grid, cdec, cent = kodak_like_grid(num_lods=num_lods, use_sga=use_sga)
    nef = NeuralImage(grid, hidden_dim=hidden_dim, num_layers=1).to(device)
    img = torch.from_numpy(make_test_image(height, width, seed)).reshape(-1, 3)
    perm = torch.randperm(height * width, generator=torch.Generator().manual_seed(seed))  # dataset shuffle_idx
    coords = shard_batch(image_coords(height, width)[perm], rank, world).contiguous().to(device)
    rgb = shard_batch(img[perm], rank, world).contiguous().to(device)
    history = []
    out = None
    if device.type == "cuda":
        torch.cuda.synchronize()
    import time
    t_loop = time.perf_counter()
    if graphed:
        fitter = GraphedImageFitter(nef, coords, rgb, steps, cdec, cent, global_pixels=height * width)
        for it in range(steps):
            fitter.step()
            if log_every and (it + 1) % log_every == 0:
                ps, rl = fitter.psnr()
                history.append((it + 1, rl, ps, float("nan")))
        ps, rl = fitter.psnr()
        out = (rl, ps, float("nan"))
    else:
        fitter = ImageFitter(nef, coords, rgb, steps, cdec, cent, world=world, global_pixels=height * width)
        for it in range(steps):
            # statistics are read back only where they are used (a logged step, the last one): the host runs ahead of the GPU
            want = (log_every and (it + 1) % log_every == 0) or it == steps - 1
            out = fitter.step(read=bool(want))
            if log_every and (it + 1) % log_every == 0:
                history.append((it + 1,) + out)
    if device.type == "cuda":
        torch.cuda.synchronize()
    t_loop = time.perf_counter() - t_loop
    ldec_bits, latent_bits = grid.size(use_torchac=False, use_prob_model=False)
    rest_bits = sum(p.numel() * 32 for n, p in nef.named_parameters() if "grid" not in n)
    bpp = (latent_bits + ldec_bits + rest_bits) / (height * width)
    from . import codec
    file_bytes = len(codec.save_model(nef))     # the real thing: range-coded latents + raw fp32 for everything else
    return dict(psnr=out[1], rgb_loss=out[0], avg_bits=out[2], bpp=bpp, bpp_file=8.0 * file_bytes / (height * width),
                file_bytes=file_bytes, steps=steps, history=history, ms_per_step=t_loop / steps * 1e3)


# =====================================================================================================================
# 3-D: NeRF-like ray points against an analytic field (configs D / E of SURVEY.md section 8)
# =====================================================================================================================
def analytic_field(points):
    """Smooth + sharp analytic colour field on [-1,1]^3 -> rgb in [0,1]. points [N,3]."""
    x, y, z = points[:, 0], points[:, 1], points[:, 2]
    r = torch.sqrt(x * x + y * y + z * z + 1e-12)
    shell = torch.sigmoid((0.6 - r) * 40.0)                                  # a sphere with a soft edge
    rgb = torch.stack([0.5 + 0.5 * torch.sin(6.0 * x + 2.0 * y) * shell,
                       0.5 + 0.5 * torch.cos(5.0 * y - 3.0 * z) * shell,
                       0.25 + 0.5 * shell * (0.5 + 0.5 * torch.sin(9.0 * z))], -1)
    box = ((x.abs() < 0.3) & (y.abs() < 0.15) & (z.abs() < 0.45)).float().unsqueeze(-1)   # a hard-edged box inside
    box_rgb = torch.stack([torch.full_like(x, 0.9), torch.full_like(x, 0.2), torch.full_like(x, 0.1)], -1)
    return torch.clamp(rgb * (1 - box) + box * box_rgb, 0.0, 1.0)


def ray_points(num_rays, samples_per_ray, generator, jitter=True):
    """NeRF-like sample positions: origins on the radius-3 sphere, directions toward a jittered point in
    [-0.5,0.5]^3, `samples_per_ray` stratified samples on the chord inside [-1,1]^3. -> [num_rays*samples_per_ray, 3]"""
    o = torch.randn(num_rays, 3, generator=generator)
    o = 3.0 * o / o.norm(dim=1, keepdim=True)
    target = torch.rand(num_rays, 3, generator=generator) - 0.5
    d = target - o
    d = d / d.norm(dim=1, keepdim=True)
    # slab intersection with the cube [-1,1]^3
    inv = 1.0 / torch.where(d.abs() < 1e-9, torch.full_like(d, 1e-9), d)
    t0, t1 = (-1.0 - o) * inv, (1.0 - o) * inv
    tnear = torch.minimum(t0, t1).max(dim=1)[0]
    tfar = torch.maximum(t0, t1).min(dim=1)[0]
    u = (torch.arange(samples_per_ray, dtype=torch.float32).unsqueeze(0)
         + (torch.rand(num_rays, samples_per_ray, generator=generator) if jitter else 0.5)) / samples_per_ray
    t = tnear.unsqueeze(1) + (tfar - tnear).unsqueeze(1) * u
    pts = o.unsqueeze(1) + d.unsqueeze(1) * t.unsqueeze(-1)
    return pts.reshape(-1, 3).clamp(-1.0, 1.0)


class NeuralField3D(nn.Module):
    """positions [N,3] -> rgb [N,3]: HashGrid (nerf_hash.yaml: 16 levels, F=2, bw 19, res 16..2048) + small MLP."""

    def __init__(self, grid, hidden_dim=64):
        super().__init__()
        self.grid = grid
        self.decoder_color = BasicDecoder(grid.feature_dim * grid.num_lods, 3, torch.relu, True, nn.Linear, 1,
                                          hidden_dim, [])

    def rgb(self, coords):
        feats = self.grid.interpolate(coords, len(self.grid.active_lods) - 1).reshape(coords.shape[0], -1)
        return torch.sigmoid(self.decoder_color(feats))


def fit_field_3d(device, steps=500, rays=4096, samples_per_ray=16, seed=0, codebook_bitwidth=19, max_grid_res=2048,
                 num_lods=16, rank=0, world=1, val_points=65536):
    """Config D/E: every step draws `rays` x `samples_per_ray` fresh ray points (global batch, sharded over ranks),
    L1 loss to the analytic field, Adam; returns dict(psnr on a fixed validation set, ms_per_step)."""
    import time
    from .dist import FlatGradients, shard_batch
    from .optim import FusedAdam
    from .wisp.models.grids import HashGrid
    from .wisp.ops.image.metrics import psnr as psnr_fn
    torch.manual_seed(seed)
    grid = HashGrid.from_geometric(feature_dim=2, num_lods=num_lods, multiscale_type="cat", resolution_dim=3,
                                   feature_std=0.01, codebook_bitwidth=codebook_bitwidth, min_grid_res=16,
                                   max_grid_res=max_grid_res, blas_level=3)
    nef = NeuralField3D(grid).to(device)
    groups = [g for g in param_groups(nef, lr=1e-3, grid_lr=1e-2) if g["params"]]
    opt = FusedAdam(groups, eps=1e-15) if device.type == "cuda" else torch.optim.Adam(groups, eps=1e-15)
    bucket = FlatGradients([p for g in groups for p in g["params"]]) if world > 1 else None
    gen = torch.Generator().manual_seed(seed + 1)                     # the SAME sample stream on every rank
    n_global = rays * samples_per_ray
    if device.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        pts = ray_points(rays, samples_per_ray, gen)
        local = shard_batch(pts, rank, world).contiguous().to(device)
        target = analytic_field(local)
        if bucket is not None:
            bucket.zero_()
        else:
            opt.zero_grad(set_to_none=True)
        loss = (nef.rgb(local) - target).abs().sum() / (n_global * 3)     # global-mean L1 (multiview_trainer.py:105-108)
        loss.backward()
        hidden = []
        if bucket is not None:
            bucket.allreduce()
            hidden = bucket.hide_untouched()
        opt.step()
        if bucket is not None:
            bucket.restore(hidden)
    if device.type == "cuda":
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    with torch.no_grad():
        vp = (torch.rand(val_points, 3, generator=torch.Generator().manual_seed(99)) * 2 - 1).to(device)
        pred = nef.rgb(vp)
        val = psnr_fn(pred.clamp(0, 1), analytic_field(vp))
    return dict(psnr=val, ms_per_step=ms, steps=steps, samples_per_step=n_global)


# ------------------------------------------------------------------------------------------------ NeRF render-and-fit
_scene_cache = {}


def _scene_constants(device):
    """Blob centres / radii on `device`, uploaded once (a host->device copy is not allowed inside a graph capture)."""
    if device not in _scene_cache:
        _scene_cache[device] = (torch.tensor([[0.35, 0.0, 0.1], [-0.3, 0.25, -0.2], [0.0, -0.4, 0.3]], device=device),
                                torch.tensor([0.32, 0.26, 0.22], device=device))
    return _scene_cache[device]


def analytic_scene(points):
    """Closed-form radiance field inside [-1,1]^3: density = three soft blobs (particles per unit length), colour =
    a smooth position-dependent albedo. -> (density [N, 1], rgb [N, 3])."""
    centers, radii = _scene_constants(points.device)
    d2 = ((points[:, None, :] - centers[None]) ** 2).sum(-1)
    density = (40.0 * torch.sigmoid((radii[None] ** 2 - d2) * 60.0)).sum(-1, keepdim=True)
    rgb = 0.5 + 0.5 * torch.stack([torch.sin(3.0 * points[:, 0] + 0.5), torch.sin(4.0 * points[:, 1] - 1.0),
                                   torch.cos(3.5 * points[:, 2] + 0.3)], -1)
    return density, rgb


def camera_rays(num_rays, generator, device):
    """Unit-direction rays from the radius-3 sphere toward a jittered point near the origin."""
    o = torch.randn(num_rays, 3, generator=generator)
    o = 3.0 * o / o.norm(dim=1, keepdim=True)
    d = (torch.rand(num_rays, 3, generator=generator) - 0.5) * 1.4 - o
    d = d / d.norm(dim=1, keepdim=True)
    return o.to(device), d.to(device)


class _AnalyticNef:
    """The closed-form scene behind the radiance-field interface, so the SAME tracer renders the targets."""

    def __init__(self, grid):
        self.grid = grid

    def __call__(self, coords, ray_d, lod_idx=None, channels=None):
        density, rgb = analytic_scene(coords)
        return dict(rgb=rgb, density=density)


class GraphedNerfFitter:
    """The NeRF render-and-fit step (multiview_trainer.py:80-156 through packed_rf_tracer.py:68-170) recorded into a HIP
    graph and replayed. Eagerly the step is ~120 launches issued from Python with a count read-back in the middle (the
    marcher's sample count sizes every later tensor): 1.8 ms per step with the GPU busy half of it.

    What capture forces, and how it is met:
    * no read-back: the marcher emits into buffers of a fixed CAPACITY (`shacira_raymarch_ray_emit_capped`); rows behind
      the last survivor are padding that belongs to no ray pack, so whatever the field computes for them integrates into
      nothing and receives a zero gradient (their positions are spread over the cube: identical positions would pile LDS
      atomics onto a handful of table rows). Capacity = the largest count of a few probe batches x `margin`, re-measured
      after every occupancy prune (the count only changes statistically in between: 4 096 rays average it to ~1.5 %);
      a step whose count exceeds it drops the excess samples and is counted in `overflow_steps`;
    * the occupancy lives in ONE device tensor that prunes update in place (the graph holds its address);
    * the batch comes from a pool of rays with their target colours resident on the device (the role of the reference's
      MultiviewDataset: rays + pixels of the training images), indexed by a step counter kept on the device; Adam keeps its
      step count on the device too (FusedAdam(capturable=True)).
    A graph is re-captured when a prune changes the capacity (a few times per run)."""

    def __init__(self, nef, tracer, groups, pool, near, far, device, margin=1.08, quantum=16384, latent=None):
        from .optim import FusedAdam
        from .wisp.accelstructs import OctreeAS
        self.nef, self.tracer = nef, tracer
        self.pool_o, self.pool_d, self.pool_rgb = pool             # [P, rays, 3] each
        self.near, self.far, self.device = near, far, device
        self.margin, self.quantum = margin, quantum
        self.opt = FusedAdam(groups, eps=1e-15, capturable=True)
        self.index = torch.zeros(1, dtype=torch.int64, device=device)     # which pool batch the next step takes
        self.loss = torch.zeros((), device=device)
        self.overflow = torch.zeros((), dtype=torch.int64, device=device)
        # the occupancy the graph reads: one persistent structure, updated in place by `after_prune`
        blas = nef.grid.blas
        self.blas = OctreeAS(blas.max_level, blas.occupancy_grid.to(device).clone())
        nef.grid.blas = self.blas
        self.graph, self.capacity, self.captures, self.capture_seconds = None, None, 0, 0.0
        self.graphs = {}                # capacity -> captured graph (a later prune may return to an earlier capacity)
        # compressed variant (3-D LatentGrid, the reference's nerf_lego.yaml mode): `latent` = dict(temperature_sched,
        # decay_period, steps, entropy_reg). The SGA temperature lives in ONE device float the decode kernels read
        # (shacira_latent_decode_sga_*_tdev), refreshed before every replay; the entropy noise is drawn on the device; when
        # SGA is switched off (after decay_period of the run) the graphs are dropped and re-captured once.
        self.latent = latent
        self.iteration = 0
        self.extra_eager_steps = 0      # optimizer steps taken by prepare()'s shape warm-ups (reported, see fit_nerf)
        if latent is not None:
            _require_device_temperature(nef.grid.latent_dec, "GraphedNerfFitter")
            self.temperature = torch.ones(1, device=device)
            nef.grid.latent_dec.temperature = self.temperature
            nef.grid.device_noise = True

    def _batch(self, k=None):
        from .wisp.core import Rays
        if k is None:     # in the step: the device-side index (index_select keeps it a kernel argument, not a host value)
            o = self.pool_o.index_select(0, self.index)[0]
            d = self.pool_d.index_select(0, self.index)[0]
            return Rays(o, d, dist_min=self.near, dist_max=self.far), self.pool_rgb.index_select(0, self.index)[0]
        return Rays(self.pool_o[k], self.pool_d[k], dist_min=self.near, dist_max=self.far), self.pool_rgb[k]

    def _body(self):
        batch, target = self._batch()
        self.opt.zero_grad(set_to_none=True)
        rb = self.tracer(self.nef, batch)
        loss = torch.abs(rb.rgb[..., :3] - target[..., :3]).mean()
        if self.latent is not None and self.latent["entropy_reg"] > 0:
            avg_bits, _ = self.nef.grid.ent_loss(1, is_val=False)       # multiview_trainer.py:109-113
            loss = loss + self.latent["entropy_reg"] * avg_bits
        loss.backward()
        self.opt.step()
        self.loss.copy_(loss.detach())
        self.index.add_(1).remainder_(self.pool_o.shape[0])
        if self.capacity is not None:
            self.overflow.add_((self.blas.last_sample_count > self.capacity).to(torch.int64))

    def _probe_capacity(self, probes=6):
        """Sample counts of a few pool batches under the current occupancy (eager, with read-backs) -> capacity."""
        self.blas.sample_capacity = None
        need = 1
        P = self.pool_o.shape[0]
        with torch.no_grad():
            for j in range(probes):
                batch, _ = self._batch((j * 7919) % P)
                m = self.nef.grid.raymarch(batch, level=None, num_samples=self.tracer.num_steps, raymarch_type="ray")
                need = max(need, m.samples.shape[0])
        return int(-(-int(need * self.margin) // self.quantum) * self.quantum)

    def prepare(self):
        """(Re)capture for the current occupancy. Call once after a few eager steps, and after every prune."""
        import time
        cap = self._probe_capacity()
        self.blas.sample_capacity = cap
        if self.graph is not None and cap == self.capacity:
            return
        self.capacity = cap
        if cap in self.graphs:
            self.graph = self.graphs[cap]
            return
        t0 = time.perf_counter()
        if len(self.graphs) >= 4:       # bound the memory held by graph pools
            self.graphs.clear()
        # one eager step at the new shapes (workspaces, kernel attributes, padding rows). It IS a real optimizer step on the
        # next pool batch -- counted, so that the reported step totals and ms_per_step include it (round-4 advisor finding: the
        # graph-replay figures came from runs with uncounted parameter updates)
        self._body()
        self.extra_eager_steps += 1
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._body()
        self.graph = self.graphs[cap] = graph
        self.captures += 1
        torch.cuda.synchronize()
        self.capture_seconds += time.perf_counter() - t0

    def step(self):
        self.iteration += 1
        if self.latent is not None:
            dec = self.nef.grid.latent_dec
            self.temperature.fill_(float(self.latent["temperature_sched"](self.iteration)))
            if dec.use_sga and self.iteration / self.latent["steps"] > self.latent["decay_period"]:
                dec.use_sga = False                 # rounding from here on: another kernel, so the step is re-captured
                had_graph = self.graph is not None
                self.graphs.clear()
                self.graph = None
                if had_graph:
                    self.capacity = None
                    self.prepare()
                    return self.graph.replay()
        if self.graph is None:
            self._body()                # eager (warm-up) steps before the first capture
        else:
            self.graph.replay()

    def after_prune(self):
        """`nef.prune()` built a new acceleration structure: move its cells into the persistent one, then re-capture if the
        sample count moved."""
        new = self.nef.grid.blas
        if new is not self.blas:
            self.blas.occupancy_grid.copy_(new.occupancy_grid.to(self.device))
            self.blas.points, self.blas.pyramid = new.points, new.pyramid
            self.nef.grid.blas = self.blas
        self.prepare()

    def eager_mode(self):
        """Back to read-back sized tensors (validation renders)."""
        self.blas.sample_capacity = None


def fit_nerf(device, steps=300, rays=4096, num_steps=128, seed=0, codebook_bitwidth=19, max_grid_res=2048,
             num_lods=16, blas_level=5, prune_every=100, val_rays=8192, hidden_dim=64, latent=False,
             entropy_reg=1.0e-4, feature_dim=2, graphed=False, ray_pool=0):
    """NeRF-style fit of the analytic scene through the full pipeline the reference runs per step
    (multiview_trainer.py:88-150): ray marching on the occupancy grid ('ray' sampler) -> hash-grid lookup -> density /
    colour decoders -> volume integration -> L1 to the target pixels -> Adam; occupancy pruned every `prune_every` steps.
    Targets are rendered from the closed-form scene by the same tracer with 4x the samples.
    `latent=True`: the compressed variant the reference's nerf_lego.yaml trains -- a 3-D LatentGrid (latent_dim 1, SGA
    warm-up with temperature 1.0 until decay_period 0.9, entropy model with one layer, lambda = `entropy_reg`); the
    result then also carries the size estimate and the bytes of the entropy-coded model file. nerf_lego.yaml's shape is
    feature_dim=4, num_lods=24, max_grid_res=512, hidden_dim=128 (tools/attic/lego_fit.py).
    `ray_pool=P` > 0: the P batches of rays and their target colours are rendered ONCE before the timed loop and the steps
    walk them in order -- the role of the reference's MultiviewDataset (rays + pixels of the training images); with 0 every
    step draws fresh rays and renders their targets from the closed-form scene inside the step (rounds 1-3 protocol).
    `graphed=True` (implies a ray pool, default 128 batches): the step replayed from HIP graphs (`GraphedNerfFitter`); with
    `latent=True` the SGA temperature is annealed through a device float between replays.
    Returns dict(psnr on held-out rays, ms_per_step, samples_per_step)."""
    import time
    from .optim import FusedAdam
    from .wisp.core import Rays
    from .wisp.models.grids import HashGrid
    from .wisp.models.nefs import NeuralRadianceField
    from .wisp.ops.image.metrics import psnr as psnr_fn
    from .wisp.tracers import PackedRFTracer
    torch.manual_seed(seed)
    cdec = None
    if latent:
        cdec = dict(ldecode_enabled=True, ldecode_type="single", use_sga=True, diff_sampling=True, use_shift=True,
                    ldecode_matrix="sq", latent_dim=1, norm="none", norm_every=10, ldec_std=0.1, decay_period=0.9,
                    temperature=1.0, num_layers_dec=0, hidden_dim_dec=0, activation="none", final_activation="none",
                    clamp_weights=0.0, num_decoders=1, alpha_std=1.0)
        cent = dict(num_prob_layers=1, entropy_reg=entropy_reg, entropy_reg_end=entropy_reg,
                    entropy_reg_sched="cosine", noise_freq=1)
        grid = LatentGrid.from_geometric(feature_dim=feature_dim, num_lods=num_lods, latent_dim=1, multiscale_type="cat",
                                         resolution_dim=3, feature_std=0.02, feature_bias=0.0,
                                         codebook_bitwidth=codebook_bitwidth, min_grid_res=16,
                                         max_grid_res=max_grid_res, blas_level=blas_level, init_grid="normal",
                                         conf_latent_decoder=cdec, conf_entropy_reg=cent)
        temperature_sched = DecayScheduler(steps, "exp", 1.0, cdec["temperature"],
                                           {"temperature": cdec["temperature"], "decay_period": cdec["decay_period"]})
        grid.device_noise = True   # 6.1 M uniforms per step: drawn on the device (the reference draws on the host + H2D)
    else:
        grid = HashGrid.from_geometric(feature_dim=feature_dim, num_lods=num_lods, multiscale_type="cat", resolution_dim=3,
                                       feature_std=0.01, codebook_bitwidth=codebook_bitwidth, min_grid_res=16,
                                       max_grid_res=max_grid_res, blas_level=blas_level)
    nef = NeuralRadianceField(grid, view_embedder="positional", view_multires=4, hidden_dim=hidden_dim, num_layers=1,
                              prune_density_decay=0.95, prune_min_density=0.01 * 512 / 3 ** 0.5).to(device)
    truth = _AnalyticNef(HashGrid.from_geometric(feature_dim=2, num_lods=2, multiscale_type="cat", resolution_dim=3,
                                                 feature_std=0.0, codebook_bitwidth=4, min_grid_res=2,
                                                 max_grid_res=4, blas_level=blas_level))   # carries the occupancy only
    tracer = PackedRFTracer(raymarch_type="ray", num_steps=num_steps, bg_color="white")
    gt_tracer = PackedRFTracer(raymarch_type="ray", num_steps=4 * num_steps, bg_color="white")
    groups = [g for g in param_groups(nef, lr=1e-3, grid_lr=1e-2) if g["params"]]
    near, far = 1.2, 4.8
    warm = min(20, steps // 10)          # first steps pay one-off costs (kernel attribute set-up, allocator growth)
    gen = torch.Generator().manual_seed(seed + 1)
    pool = None
    if graphed and not ray_pool:
        ray_pool = 128
    if ray_pool:
        po, pd, prgb = [], [], []
        with torch.no_grad():
            for _ in range(ray_pool):
                o, d = camera_rays(rays, gen, device)
                po.append(o)
                pd.append(d)
                prgb.append(gt_tracer(truth, Rays(o, d, dist_min=near, dist_max=far)).rgb)
        pool = (torch.stack(po), torch.stack(pd), torch.stack(prgb))
    if graphed:
        lat = None
        if latent:
            lat = dict(temperature_sched=temperature_sched, decay_period=cdec["decay_period"], steps=steps,
                       entropy_reg=entropy_reg)
        fitter = GraphedNerfFitter(nef, tracer, groups, pool, near, far, device, latent=lat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eager_before = 0                     # shape warm-up steps taken BEFORE the clock below starts: not in the timed region
        for it in range(steps):
            if it == warm:
                fitter.prepare()             # capture after the eager warm-up steps (captures are timed, see capture_seconds)
                torch.cuda.synchronize()
                eager_before = fitter.extra_eager_steps
                t0 = time.perf_counter()
            fitter.step()
            if prune_every and (it + 1) % prune_every == 0:
                nef.prune()
                fitter.after_prune()
        torch.cuda.synchronize()
        # (the timed region holds steps - warm loop steps plus the shape warm-up steps taken AFTER the clock started -- those of
        # re-captures behind a prune; the first prepare()'s ran before t0 was reset: round-5 advisor finding)
        ms = (time.perf_counter() - t0) / max(1, steps - warm + fitter.extra_eager_steps - eager_before) * 1e3
        fitter.eager_mode()
        with torch.no_grad():
            o, d = camera_rays(val_rays, torch.Generator().manual_seed(4242), device)
            batch = Rays(o, d, dist_min=near, dist_max=far)
            val = psnr_fn(tracer(nef, batch).rgb.clamp(0, 1), gt_tracer(truth, batch).rgb)
        out = dict(psnr=val, ms_per_step=ms, steps=steps, rays_per_step=rays, candidate_samples_per_step=rays * num_steps,
                   occupied_cells=int(nef.grid.blas.points.shape[0]), total_cells=int(grid.num_cells),
                   graph_captures=fitter.captures, sample_capacity=fitter.capacity,
                   optimizer_steps=steps + fitter.extra_eager_steps, shape_warmup_steps=fitter.extra_eager_steps,
                   overflow_steps=int(fitter.overflow.item()), ray_pool=ray_pool,
                   capture_seconds=fitter.capture_seconds)
        if latent:
            from . import codec
            grid.latent_dec.temperature = float(fitter.temperature.item())
            ldec_bits, latent_bits = grid.size(use_torchac=False, use_prob_model=False)
            out.update(table_bytes_fp32=grid.codebook.numel() * 4, latent_bytes_estimate=latent_bits / 8,
                       file_bytes=len(codec.save_model(nef)))
        return out
    opt = FusedAdam(groups, eps=1e-15)
    samples_seen = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        if it == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if pool is not None:
            k = it % ray_pool
            batch, target = Rays(pool[0][k], pool[1][k], dist_min=near, dist_max=far), pool[2][k]
        else:
            o, d = camera_rays(rays, gen, device)
            batch = Rays(o, d, dist_min=near, dist_max=far)
            with torch.no_grad():
                target = gt_tracer(truth, batch).rgb
        opt.zero_grad(set_to_none=True)
        if latent:
            grid.latent_dec.temperature = temperature_sched(it + 1)
            if (it + 1) / steps > cdec["decay_period"]:
                grid.latent_dec.use_sga = False
        rb = tracer(nef, batch)
        loss = torch.abs(rb.rgb[..., :3] - target[..., :3]).mean()
        if latent and entropy_reg > 0:
            avg_bits, _ = grid.ent_loss(it, is_val=False)       # multiview_trainer.py:109-113
            loss = loss + entropy_reg * avg_bits
        loss.backward()
        opt.step()
        samples_seen += rays * num_steps
        if prune_every and (it + 1) % prune_every == 0:
            nef.prune()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / max(1, steps - warm) * 1e3
    with torch.no_grad():
        o, d = camera_rays(val_rays, torch.Generator().manual_seed(4242), device)
        batch = Rays(o, d, dist_min=near, dist_max=far)
        val = psnr_fn(tracer(nef, batch).rgb.clamp(0, 1), gt_tracer(truth, batch).rgb)
    occupied = int(nef.grid.blas.points.shape[0])
    out = dict(psnr=val, ms_per_step=ms, steps=steps, rays_per_step=rays, candidate_samples_per_step=rays * num_steps,
               occupied_cells=occupied, total_cells=int(grid.num_cells), ray_pool=ray_pool)
    if latent:
        from . import codec
        ldec_bits, latent_bits = grid.size(use_torchac=False, use_prob_model=False)
        out.update(table_bytes_fp32=grid.codebook.numel() * 4, latent_bytes_estimate=latent_bits / 8,
                   file_bytes=len(codec.save_model(nef)))
    return out
