"""Inference callers of the hot path, device-side and batched.

`test_single_volume` mirrors `code/val_2D.py:54-97` (per-slice zoom to `patch_size`, forward, head ensemble,
softmax, argmax, zoom back) and `test_single_case` mirrors `code/test_3D_util.py:14-79` (zero padding, sliding
window, softmax score accumulation, count normalisation, argmax).  Same names, arguments and return values; the
differences are internal: slices / patches go through the network in batches, the ensemble + softmax + argmax and
the score-map accumulation run in HIP kernels (`chap_ensemble_argmax`, `chap_window_accumulate/finalize`) and
nothing is copied to the host per slice / per patch.  The nearest-neighbour zoom is an index plan that reproduces
scipy.ndimage.zoom(order=0) bit for bit (including its edge quirk) and runs on the device.
"""
import math

import numpy as np
import torch

from . import ops


_ZOOM_PLANS = {}


def _zoom0_axis(n_in, factor):
    """Index plan of scipy.ndimage.zoom(order=0, mode='constant', grid_mode=False) along one axis: output size
    round(n_in * factor); output o reads input floor(o * (n_in-1)/(n_out-1) + 0.5); a coordinate that floating point
    pushes past n_in - 1 is *outside* and yields the constant 0 (scipy's edge quirk, reproduced on purpose: the
    reference's predictions contain it).  Checked against scipy for 6 800 size pairs (tests/test_oracle_inference_cpu.py)."""
    key = (int(n_in), float(factor))
    plan = _ZOOM_PLANS.get(key)
    if plan is None:
        n_out = int(round(n_in * factor))
        if n_out <= 1:
            idx, inside = np.zeros(max(n_out, 1), dtype=np.int64), np.ones(max(n_out, 1), dtype=bool)
        else:
            c = np.arange(n_out, dtype=np.float64) * (float(n_in - 1) / float(n_out - 1))
            inside = (c >= 0) & (c <= n_in - 1)
            idx = np.clip(np.floor(c + 0.5).astype(np.int64), 0, n_in - 1)
        plan = (torch.from_numpy(idx), torch.from_numpy(inside))
        _ZOOM_PLANS[key] = plan
    return plan


def zoom0(t, factors):
    """Nearest-neighbour zoom of the last two dims of a torch tensor, bit-identical to scipy.ndimage.zoom(order=0)."""
    iy, my = _zoom0_axis(t.shape[-2], factors[0])
    ix, mx = _zoom0_axis(t.shape[-1], factors[1])
    iy, my, ix, mx = iy.to(t.device), my.to(t.device), ix.to(t.device), mx.to(t.device)
    out = t.index_select(-2, iy).index_select(-1, ix)
    mask = my[:, None] & mx[None, :]
    return torch.where(mask, out, torch.zeros((), dtype=t.dtype, device=t.device))


def _dice_hd95(pred, gt):
    """calculate_metric_percase (val_2D.py:43-51): Dice and HD95 of two binary masks (medpy when installed)."""
    pred = (pred > 0)
    gt = (gt > 0)
    if pred.sum() > 0:
        try:
            from medpy import metric
            return metric.binary.dc(pred, gt), metric.binary.hd95(pred, gt)
        except ImportError:
            inter = np.count_nonzero(pred & gt)
            size = np.count_nonzero(pred) + np.count_nonzero(gt)
            return (2.0 * inter / float(size) if size else 0.0), float("nan")     # HD95 needs medpy
    return 0, 0


def predict_volume(image, net, patch_size=(256, 256), model_type="logit_ensemble", device="cuda:0", batch=32):
    """image: numpy [S, X, Y] -> prediction uint8 [S, X, Y] (the loop body of test_single_volume, all slices batched)."""
    S, x, y = image.shape
    vol = torch.from_numpy(np.ascontiguousarray(image)).to(device)
    zoomed = zoom0(vol, (patch_size[0] / x, patch_size[1] / y))          # the reference's per-slice zoom(order=0), on the device
    net.eval()
    out = torch.empty((S,) + tuple(zoomed.shape[1:]), dtype=torch.uint8, device=device)
    with torch.no_grad():
        for s0 in range(0, S, batch):
            inp = zoomed[s0:s0 + batch].unsqueeze(1).float().contiguous()
            o = net(inp)
            if isinstance(o, (tuple, list)):
                o1, o2 = o[0], o[1]
            else:
                o1, o2 = o, None
            mode = model_type if o2 is not None else "model1"
            label, _ = ops.ensemble_argmax(o1.contiguous(), None if o2 is None else o2.contiguous(), mode)
            out[s0:s0 + batch] = label
    pred = zoom0(out, (x / patch_size[0], y / patch_size[1]))
    return pred.cpu().numpy()


def test_single_volume(image, label, net, classes, patch_size=[256, 256], model_type="unet", device="cuda:0", batch=32):
    """val_2D.py:54-97.  image/label: tensors [1, S, X, Y]; returns [(dice, hd95)] for classes 1..classes-1.
    `model_type` in {'model1', 'model2', 'logit_ensemble', 'prob_ensemble'} (any other value raises NameError in the
    reference: `prob` is undefined; here it raises ValueError)."""
    image, label = image.squeeze(0).cpu().detach().numpy(), label.squeeze(0).cpu().detach().numpy()
    if model_type not in ops.ENSEMBLE_MODES:
        raise ValueError("model_type must be one of %s" % sorted(ops.ENSEMBLE_MODES))
    prediction = predict_volume(image, net, tuple(patch_size), model_type, device, batch).astype(label.dtype)
    metric_list = []
    for i in range(1, classes):
        metric_list.append(_dice_hd95(prediction == i, label == i))
    return metric_list


def test_single_case(net, image, stride_xy, stride_z, patch_size, num_classes=1, batch=4, device="cuda:0", return_score=False):
    """test_3D_util.py:14-79.  image: numpy [w, h, d]; returns label_map int64 [w, h, d]."""
    w, h, d = image.shape
    add_pad = False
    w_pad = max(patch_size[0] - w, 0)
    h_pad = max(patch_size[1] - h, 0)
    d_pad = max(patch_size[2] - d, 0)
    add_pad = bool(w_pad or h_pad or d_pad)
    wl_pad, wr_pad = w_pad // 2, w_pad - w_pad // 2
    hl_pad, hr_pad = h_pad // 2, h_pad - h_pad // 2
    dl_pad, dr_pad = d_pad // 2, d_pad - d_pad // 2
    if add_pad:
        image = np.pad(image, [(wl_pad, wr_pad), (hl_pad, hr_pad), (dl_pad, dr_pad)], mode="constant", constant_values=0)
    ww, hh, dd = image.shape
    sx = math.ceil((ww - patch_size[0]) / stride_xy) + 1
    sy = math.ceil((hh - patch_size[1]) / stride_xy) + 1
    sz = math.ceil((dd - patch_size[2]) / stride_z) + 1
    origins = []
    for x in range(0, sx):
        xs = min(stride_xy * x, ww - patch_size[0])
        for y in range(0, sy):
            ys = min(stride_xy * y, hh - patch_size[1])
            for z in range(0, sz):
                zs = min(stride_z * z, dd - patch_size[2])
                origins.append((xs, ys, zs))
    vol = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).to(device)
    score = torch.zeros((num_classes, ww, hh, dd), dtype=torch.float32, device=device)
    cnt = torch.zeros((ww, hh, dd), dtype=torch.float32, device=device)
    net.eval()
    with torch.no_grad():
        for k0 in range(0, len(origins), batch):
            og = origins[k0:k0 + batch]
            patches = torch.stack([vol[xs:xs + patch_size[0], ys:ys + patch_size[1], zs:zs + patch_size[2]] for xs, ys, zs in og]).unsqueeze(1)
            y1 = net(patches.contiguous())
            if isinstance(y1, (tuple, list)):
                y1 = y1[0]
            ops.window_accumulate(y1.contiguous(), torch.tensor(og, dtype=torch.int32, device=device), score, cnt)
    label = ops.window_finalize(score, cnt)
    label_map = label.cpu().numpy().astype(np.int64)
    score_map = score.cpu().numpy() if return_score else None
    if add_pad:
        label_map = label_map[wl_pad:wl_pad + w, hl_pad:hl_pad + h, dl_pad:dl_pad + d]
        if return_score:
            score_map = score_map[:, wl_pad:wl_pad + w, hl_pad:hl_pad + h, dl_pad:dl_pad + d]
    return (label_map, score_map) if return_score else label_map
