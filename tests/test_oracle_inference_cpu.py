"""CPU checks of oracle/inference.py (the restatement the GPU inference tests compare against)."""
import numpy as np
import torch

from oracle import inference as oinf


def test_single_case_padding_and_argmax():
    # a "network" whose class-1 logit is the voxel value itself: label = (value > 0) wherever a patch landed
    net = lambda x: torch.cat([torch.zeros_like(x), x], dim=1)
    rng = np.random.default_rng(0)
    img = rng.standard_normal((10, 6, 9)).astype(np.float32)          # h < patch -> zero padding branch
    label, score = oinf.test_single_case(net, img, 3, 2, (8, 8, 8), num_classes=2)
    assert label.shape == img.shape and score.shape == (2,) + img.shape
    assert np.array_equal(label, (img > 0).astype(np.int64))
    assert np.allclose(score.sum(0), 1.0, atol=1e-6)                    # averaged softmax scores still sum to one


def test_predict_volume_modes():
    g = torch.Generator().manual_seed(0)
    w1, w2 = torch.randn(3, 1, 1, 1, generator=g), torch.randn(3, 1, 1, 1, generator=g)
    net = lambda x: (x * w1.view(1, 3, 1, 1), x * w2.view(1, 3, 1, 1))
    img = np.random.default_rng(1).random((3, 20, 24), dtype=np.float32) + 0.1
    p1 = oinf.predict_volume(img, net, (16, 16), "model1")
    assert p1.shape == img.shape and set(np.unique(p1)) <= {0, 1, 2}
    assert (p1 == int(torch.argmax(w1.flatten()))).all()              # positive inputs: argmax follows the weight
    p2 = oinf.predict_volume(img, net, (16, 16), "model2")
    assert (p2 == int(torch.argmax(w2.flatten()))).all()
    pl = oinf.predict_volume(img, net, (16, 16), "logit_ensemble")
    assert (pl == int(torch.argmax((w1 + w2).flatten()))).all()


def test_zoom0_matches_scipy():
    """chap_amd.inference.zoom0 == scipy.ndimage.zoom(order=0), including the outputs scipy zeroes at the far edge."""
    from scipy.ndimage import zoom
    from chap_amd.inference import zoom0
    rng = np.random.default_rng(0)
    n = 0
    for (x, y) in [(50, 44), (216, 256), (256, 216), (224, 224), (63, 31), (512, 300), (30, 59), (7, 9), (100, 53)]:
        for (px, py) in [(64, 64), (256, 256), (224, 224), (31, 100), (216, 8)]:
            a = rng.random((x, y)).astype(np.float32) + 1.0
            ref = zoom(a, (px / x, py / y), order=0)
            got = zoom0(torch.from_numpy(a), (px / x, py / y)).numpy()
            assert got.shape == ref.shape and np.array_equal(got, ref), (x, y, px, py)
            lab = (rng.random((px, py)) * 4).astype(np.uint8)
            refb = zoom(lab, (x / px, y / py), order=0)
            gotb = zoom0(torch.from_numpy(lab), (x / px, y / py)).numpy()
            assert np.array_equal(gotb, refb), (x, y, px, py)
            n += 1
    assert n == 45
