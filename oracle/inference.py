"""ORACLE (test infrastructure only): CPU restatement of the reference's inference callers.

`predict_volume` follows the loop body of `test_single_volume` (`code/val_2D.py:54-92`): per slice nearest zoom to
the patch size, forward, head ensemble (`:64-80`), softmax, argmax, nearest zoom back.  `test_single_case` follows
`code/test_3D_util.py:14-79` line by line.  `net` is any callable on CPU tensors (e.g. a closure over oracle.nets).
Pinning: these are line-by-line restatements (the modules import medpy/h5py/SimpleITK, absent here, so they cannot be
imported); the kernels they check are additionally compared with plain torch ops in tests/test_inference_gpu.py.
"""
import math

import numpy as np
import torch
from scipy.ndimage import zoom


def predict_volume(image, net, patch_size=(256, 256), model_type="logit_ensemble"):
    prediction = np.zeros(image.shape, dtype=np.uint8)
    for ind in range(image.shape[0]):
        slice_ = image[ind, :, :]
        x, y = slice_.shape[0], slice_.shape[1]
        slice_ = zoom(slice_, (patch_size[0] / x, patch_size[1] / y), order=0)
        inp = torch.from_numpy(slice_).unsqueeze(0).unsqueeze(0).float()
        with torch.no_grad():
            if model_type == "model1":
                prob = torch.softmax(net(inp)[0], dim=1)
            elif model_type == "model2":
                prob = torch.softmax(net(inp)[1], dim=1)
            elif model_type == "logit_ensemble":
                o1, o2 = net(inp)
                prob = torch.softmax((o1 + o2) / 2.0, dim=1)
            elif model_type == "prob_ensemble":
                o1, o2 = net(inp)
                prob = (torch.softmax(o1, dim=1) + torch.softmax(o2, dim=1)) / 2.0
            else:
                raise ValueError(model_type)
            out = torch.argmax(prob, dim=1).squeeze(0).numpy()
        prediction[ind] = zoom(out, (x / patch_size[0], y / patch_size[1]), order=0)
    return prediction


def test_single_case(net, image, stride_xy, stride_z, patch_size, num_classes=1):
    w, h, d = image.shape
    add_pad = False
    if w < patch_size[0]:
        w_pad = patch_size[0] - w; add_pad = True
    else:
        w_pad = 0
    if h < patch_size[1]:
        h_pad = patch_size[1] - h; add_pad = True
    else:
        h_pad = 0
    if d < patch_size[2]:
        d_pad = patch_size[2] - d; add_pad = True
    else:
        d_pad = 0
    wl_pad, wr_pad = w_pad // 2, w_pad - w_pad // 2
    hl_pad, hr_pad = h_pad // 2, h_pad - h_pad // 2
    dl_pad, dr_pad = d_pad // 2, d_pad - d_pad // 2
    if add_pad:
        image = np.pad(image, [(wl_pad, wr_pad), (hl_pad, hr_pad), (dl_pad, dr_pad)], mode="constant", constant_values=0)
    ww, hh, dd = image.shape
    sx = math.ceil((ww - patch_size[0]) / stride_xy) + 1
    sy = math.ceil((hh - patch_size[1]) / stride_xy) + 1
    sz = math.ceil((dd - patch_size[2]) / stride_z) + 1
    score_map = np.zeros((num_classes,) + image.shape).astype(np.float32)
    cnt = np.zeros(image.shape).astype(np.float32)
    for x in range(0, sx):
        xs = min(stride_xy * x, ww - patch_size[0])
        for y in range(0, sy):
            ys = min(stride_xy * y, hh - patch_size[1])
            for z in range(0, sz):
                zs = min(stride_z * z, dd - patch_size[2])
                test_patch = image[xs:xs + patch_size[0], ys:ys + patch_size[1], zs:zs + patch_size[2]]
                test_patch = torch.from_numpy(np.expand_dims(np.expand_dims(test_patch, axis=0), axis=0).astype(np.float32))
                with torch.no_grad():
                    y1 = net(test_patch)
                    yy = torch.softmax(y1, dim=1)
                yy = yy.numpy()[0]
                score_map[:, xs:xs + patch_size[0], ys:ys + patch_size[1], zs:zs + patch_size[2]] += yy
                cnt[xs:xs + patch_size[0], ys:ys + patch_size[1], zs:zs + patch_size[2]] += 1
    score_map = score_map / np.expand_dims(cnt, axis=0)
    label_map = np.argmax(score_map, axis=0)
    if add_pad:
        label_map = label_map[wl_pad:wl_pad + w, hl_pad:hl_pad + h, dl_pad:dl_pad + d]
        score_map = score_map[:, wl_pad:wl_pad + w, hl_pad:hl_pad + h, dl_pad:dl_pad + d]
    return label_map, score_map
