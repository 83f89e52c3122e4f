"""Fixed-seed synthetic slices / volumes (the reference's data layer is absent).
Pure torch-CPU generators of *inputs*: used by bench.py, the tools and -- through oracle.train_step, which re-exports
them -- the tests.  Nothing here computes an expected output."""
import torch


def synthetic_batch(seed, n_lab, n_unlab, h, w, n_classes=4):
    """Fixed-seed synthetic slices: sum of anisotropic Gaussian blobs + noise,
    min-max normalised to [0,1]; labels = nested thresholds of the dominant blob (one region per class)."""
    g = torch.Generator().manual_seed(seed)
    n = n_lab + n_unlab
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    imgs = torch.zeros(n, 1, h, w)
    labs = torch.zeros(n, h, w, dtype=torch.int64)
    for i in range(n):
        cy, cx = (0.3 + 0.4 * torch.rand(2, generator=g)) * torch.tensor([h, w])
        sy, sx = (0.12 + 0.1 * torch.rand(2, generator=g)) * torch.tensor([h, w])
        main = torch.exp(-(((yy - cy) / sy) ** 2 + ((xx - cx) / sx) ** 2))
        img = main.clone()
        for _ in range(int(torch.randint(2, 5, (1,), generator=g))):
            by, bx = torch.rand(2, generator=g) * torch.tensor([h, w])
            bs = (0.05 + 0.1 * torch.rand(1, generator=g)) * h
            img += 0.5 * torch.rand(1, generator=g) * torch.exp(-(((yy - by) / bs) ** 2 + ((xx - bx) / bs) ** 2))
        img += 0.05 * torch.randn(h, w, generator=g)
        img = (img - img.min()) / (img.max() - img.min() + 1e-8)
        imgs[i, 0] = img
        lab = torch.zeros(h, w, dtype=torch.int64)
        for c in range(1, n_classes):
            lab[main > (0.25 + 0.5 * c / n_classes)] = c
        labs[i] = lab
    return imgs, labs


def synthetic_batch_3d(seed, n_lab, n_unlab, d, h, w):
    """Fixed-seed synthetic volumes (LA-like, 2 classes): one anisotropic Gaussian blob = foreground."""
    g = torch.Generator().manual_seed(seed)
    n = n_lab + n_unlab
    zz, yy, xx = torch.meshgrid(torch.arange(d, dtype=torch.float32), torch.arange(h, dtype=torch.float32),
                                torch.arange(w, dtype=torch.float32), indexing="ij")
    imgs = torch.zeros(n, 1, d, h, w)
    labs = torch.zeros(n, d, h, w, dtype=torch.int64)
    dims = torch.tensor([d, h, w], dtype=torch.float32)
    for i in range(n):
        c = (0.35 + 0.3 * torch.rand(3, generator=g)) * dims
        s = (0.12 + 0.1 * torch.rand(3, generator=g)) * dims
        main = torch.exp(-(((zz - c[0]) / s[0]) ** 2 + ((yy - c[1]) / s[1]) ** 2 + ((xx - c[2]) / s[2]) ** 2))
        img = main.clone()
        for _ in range(2):
            b = torch.rand(3, generator=g) * dims
            bs = (0.05 + 0.1 * torch.rand(1, generator=g)) * d
            img += 0.5 * torch.rand(1, generator=g) * torch.exp(-(((zz - b[0]) / bs) ** 2 + ((yy - b[1]) / bs) ** 2 + ((xx - b[2]) / bs) ** 2))
        img += 0.05 * torch.randn(d, h, w, generator=g)
        imgs[i, 0] = (img - img.min()) / (img.max() - img.min() + 1e-8)
        labs[i] = (main > 0.5).long()
    return imgs, labs
