#!/usr/bin/env python3
"""A LightGlue checkpoint that was actually TRAINED — by gradient descent, on a synthetic matching task — for the one question the calibrated stand-ins (recipes D / E)
cannot settle: what do the attention logits, LayerNorm gains and residual norms of a trained matcher look like, and does the engine's precision design hold on them?
(The released checkpoints are network-only: lightglue.py:349, :416-421.)

Runs in the BUILD CONTAINER only (CPU, autograd through the unmodified reference module loaded standalone from /root/reference; nothing of it travels): the modules'
own forward methods are called in the order of `_forward` (ref :483-566) and the assignment matrix of EVERY layer is supervised, as in the paper (deep supervision:
negative log-likelihood of the ground-truth matches + the dustbin terms of unmatched points, ref :265-277 for the matrix); the token-confidence heads learn to predict
whether a point's match at layer i already equals its final one (on detached features).  Task: keypoints uniform in a 640 x 480 image, image 1 = a random
similarity + mild perspective warp of a random subset (points that leave the frame or are dropped become unmatched) plus distractors; descriptors = unit-norm Gaussians
with heavy noise between the views, and a third of the points in clusters that SHARE a descriptor (repetitive structure: only geometry tells them apart).

Output: a state dict in the reference's module-tree names (what `LightGlue.load_state_dict` and tools/verify_pretrained.py take).  ~47 MB, so it is NOT committed:
tests/golden/_local/ is git-ignored; the script, its seed and the numbers it printed are (profiles/r06tr_*).

usage: train_synthetic_checkpoint.py [--steps 900] [--out tests/golden/_local/synthetic_trained_L9.pth] [--threads 8]"""
import argparse
import importlib.util
import math
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/lightglue/lightglue.py")


def load_reference():
    spec = importlib.util.spec_from_file_location("lg_ref", str(REF))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_pairs(gen: torch.Generator, B: int, n: int, W=640.0, H=480.0, noise=(0.06, 0.11), cluster_frac=0.33, drop=0.25, distract=0.25):
    """B pairs with exactly n keypoints per image.  Returns keypoints / descriptors of both images and gt0 [B, n]: index in image 1 or -1."""
    r = lambda *s: torch.rand(*s, generator=gen)
    g = lambda *s: torch.randn(*s, generator=gen)
    k0 = torch.stack([r(B, n) * W, r(B, n) * H], -1)
    d0 = g(B, n, 256)
    # repetitive structure: clusters of 4 points that share a base descriptor (+ a small individual part)
    nc = int(cluster_frac * n) // 4
    for b in range(B):
        base = g(nc, 256)
        idx = torch.randperm(n, generator=gen)[: nc * 4].view(nc, 4)
        d0[b, idx.reshape(-1)] = (base[:, None, :] + 0.15 * g(nc, 4, 256)).reshape(-1, 256)
    d0 = torch.nn.functional.normalize(d0, dim=-1)
    k1 = torch.zeros(B, n, 2); d1 = torch.zeros(B, n, 256); gt0 = torch.full((B, n), -1, dtype=torch.long)
    for b in range(B):
        ang = (r(1).item() - 0.5) * math.pi / 2.5; sc = 0.75 + 0.6 * r(1).item()
        A = sc * torch.tensor([[math.cos(ang), -math.sin(ang)], [math.sin(ang), math.cos(ang)]])
        c = torch.tensor([W / 2, H / 2]); t = (r(2) - 0.5) * torch.tensor([0.3 * W, 0.3 * H])
        p = (r(2) - 0.5) * 4e-4
        x = k0[b] - c
        den = 1.0 + x @ p
        y = (x @ A.T) / den[:, None] + c + t + 1.5 * g(n, 2)
        inside = (y[:, 0] >= 0) & (y[:, 0] < W) & (y[:, 1] >= 0) & (y[:, 1] < H) & (r(n) > drop)
        src = torch.where(inside)[0]
        src = src[torch.randperm(len(src), generator=gen)][: int((1.0 - distract) * n)]
        m = len(src)
        sig = noise[0] + (noise[1] - noise[0]) * r(1).item()
        kk = torch.cat([y[src], torch.stack([r(n - m) * W, r(n - m) * H], -1)])
        dd = torch.cat([d0[b, src] + sig * g(m, 256), torch.nn.functional.normalize(g(n - m, 256), dim=-1)])
        perm = torch.randperm(n, generator=gen)
        k1[b] = kk[perm]; d1[b] = torch.nn.functional.normalize(dd[perm], dim=-1)
        inv = torch.empty(n, dtype=torch.long); inv[perm] = torch.arange(n)
        gt0[b, src] = inv[:m]
    size = torch.tensor([[W, H]]).expand(B, 2).contiguous()
    return k0, d0, k1, d1, gt0, size


def layer_scores(model, lg, k0, d0, k1, d1, size):
    """The reference's modules in `_forward`'s order (ref :492-494, :522-525, :538-543, :591); returns the log-assignment matrix and the descriptors of every layer."""
    kp0 = lg.normalize_keypoints(k0, size).clone(); kp1 = lg.normalize_keypoints(k1, size).clone()
    x0, x1 = model.input_proj(d0), model.input_proj(d1)
    e0, e1 = model.posenc(kp0), model.posenc(kp1)
    out = []
    for i in range(model.conf.n_layers):
        x0, x1 = model.transformers[i](x0, x1, e0, e1)
        scores, _ = model.log_assignment[i](x0, x1)
        out.append((scores, x0, x1))
    return out


def nll(scores, gt0):
    B, n = gt0.shape
    m = scores.shape[2] - 1
    matched0 = gt0 >= 0
    bi, ii = torch.where(matched0)
    pos = scores[bi, ii, gt0[bi, ii]]
    matched1 = torch.zeros(B, m, dtype=torch.bool); matched1[bi, gt0[bi, ii]] = True
    un0 = scores[:, :-1, -1][~matched0]; un1 = scores[:, -1, :-1][~matched1]
    return -(pos.mean() + 0.5 * un0.mean() + 0.5 * un1.mean())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=900)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--kpts", type=int, default=192)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden" / "_local" / "synthetic_trained_L9.pth"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    torch.manual_seed(a.seed)
    lg = load_reference()
    model = lg.LightGlue(features=None, depth_confidence=-1, width_confidence=-1).train()
    gen = torch.Generator().manual_seed(1000 + a.seed)
    opt = torch.optim.AdamW(model.parameters(), lr=a.lr, weight_decay=1e-4)
    L = model.conf.n_layers
    t0 = time.time()
    for step in range(a.steps):
        lr = a.lr * min(1.0, (step + 1) / 50) * (0.5 * (1 + math.cos(math.pi * step / a.steps)) * 0.9 + 0.1)
        for grp in opt.param_groups:
            grp["lr"] = lr
        k0, d0, k1, d1, gt0, size = make_pairs(gen, a.batch, a.kpts)
        layers = layer_scores(model, lg, k0, d0, k1, d1, size)
        loss = sum(nll(s, gt0) for s, _, _ in layers) / L
        # token confidence (ref :84-94, trained on detached features as in the paper's second stage): does the point's match at layer i equal the final one?
        with torch.no_grad():
            fin0 = layers[-1][0][:, :-1, :].argmax(-1); fin1 = layers[-1][0][:, :, :-1].argmax(-2)
        closs = 0.0
        for i in range(L - 1):
            s, x0, x1 = layers[i]
            with torch.no_grad():
                y0 = (s[:, :-1, :].argmax(-1) == fin0).float(); y1 = (s[:, :, :-1].argmax(-2) == fin1).float()
            c0, c1 = model.token_confidence[i](x0.detach(), x1.detach())
            closs = closs + torch.nn.functional.binary_cross_entropy(c0, y0) + torch.nn.functional.binary_cross_entropy(c1, y1)
        total = loss + closs / (2 * (L - 1))
        opt.zero_grad(set_to_none=True)
        total.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        if step % 25 == 0 or step == a.steps - 1:
            with torch.no_grad():
                s = layers[-1][0]
                pred = s[:, :-1, :].argmax(-1); pred[pred == s.shape[2] - 1] = -1
                acc = (pred == gt0).float().mean().item()
                first = (layers[0][0][:, :-1, :].argmax(-1) == fin0).float().mean().item()
            print(f"step {step:4d}  nll {loss.item():7.4f}  conf-bce {float(closs) / (2 * (L - 1)):6.4f}  row accuracy (last layer, incl. unmatched) {acc:.3f}  layer-0 agreement with the last {first:.3f}  lr {lr:.2e}  {time.time() - t0:6.0f} s", flush=True)
    model.eval()
    out = Path(a.out); out.parent.mkdir(parents=True, exist_ok=True)
    torch.save({k: v.detach().clone() for k, v in model.state_dict().items()}, str(out))
    print(f"saved {out} ({out.stat().st_size / 1e6:.1f} MB)")
    # one pair of the TASK's own distribution at the product's size (tools/verify_pretrained.py <checkpoint> <this file>)
    k0, d0, k1, d1, gt0, size = make_pairs(torch.Generator().manual_seed(77), 1, 1024)
    np.savez(str(out.with_name("synthetic_task_pair_1024.npz")), keypoints0=k0[0].numpy(), descriptors0=d0[0].numpy(), keypoints1=k1[0].numpy(), descriptors1=d1[0].numpy(),
             image_size0=size[0].numpy(), image_size1=size[0].numpy(), gt0=gt0[0].numpy())


if __name__ == "__main__":
    main()
