"""Throughput of the inference callers on one MI355X, with the oracle (CPU, same weights) beside it.
2D: test_single_volume recipe on a synthetic 16-slice 256x256 volume (DualDecoder, logit ensemble).
3D: test_single_case recipe, V-Net, 112x112x80 patches over a 160x160x96 volume, stride 18 / 4 as in test_LA.py:50-53
     (bounded: stride_xy 48, stride_z 16 so the CPU leg finishes)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from chap_amd import inference
from chap_amd.networks.net_factory import net_factory
from chap_amd.networks import net_factory_3d
from oracle import inference as oinf, init as oinit, nets as onets

DEV = "cuda:0"


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


if __name__ == "__main__":
    out = {}
    rng = np.random.default_rng(0)
    for dt in (torch.float32, torch.bfloat16):
        sd = oinit.dual_decoder_2d_state(3)
        m = net_factory("dualdecoder", 1, 4, DEV, {"decoder_type": "mcnet"}); m.load_state_dict(sd); m.set_compute_dtype(dt).eval()
        vol = rng.random((16, 256, 216), dtype=np.float32)
        s = timed(lambda: inference.predict_volume(vol, m, (256, 256), "logit_ensemble", DEV, batch=16), 5)
        out["2d_slices_per_s_%s" % ("f32" if dt == torch.float32 else "bf16")] = round(16 / s, 1)
    t = time.perf_counter(); oinf.predict_volume(vol[:4], lambda x: onets.dual_decoder_2d(sd, x, train=False), (256, 256), "logit_ensemble")
    out["2d_slices_per_s_cpu_oracle"] = round(4 / (time.perf_counter() - t), 2)
    sd3 = oinit.vnet_state(5)
    img = rng.random((160, 160, 96), dtype=np.float32)
    for dt in (torch.float32, torch.bfloat16):
        v = net_factory_3d("vnet", 1, 2, "test", DEV); v.load_state_dict(sd3); v.set_compute_dtype(dt).eval()
        npatch = (int(np.ceil((160 - 112) / 48)) + 1) ** 2 * (int(np.ceil((96 - 80) / 16)) + 1)
        s = timed(lambda: inference.test_single_case(v, img, 48, 16, (112, 112, 80), num_classes=2, batch=4, device=DEV), 3)
        out["3d_patches_per_s_%s" % ("f32" if dt == torch.float32 else "bf16")] = round(npatch / s, 2)
    t = time.perf_counter(); oinf.test_single_case(lambda x: onets.vnet_3d(sd3, x, train=False), img[:112, :112, :80], 48, 16, (112, 112, 80), num_classes=2)
    out["3d_patches_per_s_cpu_oracle"] = round(1 / (time.perf_counter() - t), 3)
    out["cpu_threads"] = torch.get_num_threads()
    print(json.dumps(out))
