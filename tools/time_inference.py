"""Timing of the inference callers (chap_amd.inference: test_single_volume = code/val_2D.py:54-97, test_single_case = code/test_3D_util.py:14-79) on
synthetic volumes of the datasets' typical sizes, bf16 and fp32.  Not the headline metric (that is the training iteration): a record for DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chap_amd.networks import DualDecoder, DualDecoder3d
from chap_amd.inference import test_single_volume, test_single_case

dev = "cuda:0"
rng = np.random.default_rng(0)
for dt in (torch.bfloat16, torch.float32):
    m = DualDecoder(1, 4, {"decoder_type": "mcnet"}).to(dev).eval().set_compute_dtype(dt)
    img = torch.from_numpy(rng.random((1, 10, 216, 256), dtype=np.float32))           # an ACDC volume [1, S, X, Y]: ~10 slices, zoomed to 256 x 256 per slice
    lab = torch.from_numpy(((rng.random((1, 10, 216, 256)) > 0.5) * rng.integers(1, 4, (1, 10, 216, 256))).astype(np.uint8))
    test_single_volume(img, lab, m, classes=4, patch_size=[256, 256], model_type="logit_ensemble", device=dev)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        test_single_volume(img, lab, m, classes=4, patch_size=[256, 256], model_type="logit_ensemble", device=dev)
    torch.cuda.synchronize()
    print("2D test_single_volume 10x216x256 %s: %.1f ms / volume (Dice + HD95 on the host included)" % (str(dt).split(".")[1], (time.perf_counter() - t) / 5 * 1e3), flush=True)
    m3 = DualDecoder3d(1, 2, normalization="batchnorm", has_dropout=False).to(dev).eval().set_compute_dtype(dt)
    vol = rng.random((160, 160, 88), dtype=np.float32)            # an LA volume after cropping; 112 x 112 x 80 patches, stride 18 / 4 (test_3D_util.py)
    test_single_case(m3, vol, 18, 4, (112, 112, 80), num_classes=2, device=dev)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        test_single_case(m3, vol, 18, 4, (112, 112, 80), num_classes=2, device=dev)
    torch.cuda.synchronize()
    sx = int(np.ceil((160 - 112) / 18)) + 1
    sz = int(np.ceil((88 - 80) / 4)) + 1
    print("3D test_single_case 160x160x88, %d patches %s: %.1f ms / volume" % (sx * sx * sz, str(dt).split(".")[1], (time.perf_counter() - t) / 3 * 1e3), flush=True)
