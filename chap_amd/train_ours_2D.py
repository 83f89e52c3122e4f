"""train(args, snapshot_path): the thin host loop around the MI355X iteration, with the reference's entry-point
signature and outputs (code/train_ours_2D.py:219-463): `args` is the dict of its argparse flags (`vars(args)`, :525-526);
it writes `latest.pth` and `{model}_best_model.pth` (= torch.save(model.state_dict()), :428-435), `val.csv` (:437-449)
and `log.txt` (:567-570) under `snapshot_path`.

What is NOT here (out of scope): the ACDC h5 readers, RandomGenerator augmentation, TensorBoard /
wandb.  The data layer is an argument: `args["trainloader"]` / `args["val_volumes"]` may carry any iterable of
{'image': [B,1,H,W] fp32, 'label': [B,H,W]} dicts / list of (image [1,S,H,W], label [1,S,H,W]) tensors (what the reference's
valloader yields, :274); without them the
fixed-seed synthetic generator of chap_amd.synthetic stands in (there is no dataset on the GPU box)."""
import csv
import logging
import os
import time

import numpy as np
import torch

from .inference import test_single_volume
from .networks.net_factory import net_factory
from .synthetic import synthetic_batch
from .train import DEFAULT_ARGS, ChapStep

FLAG_DEFAULTS = dict(DEFAULT_ARGS, model="dualdecoder", decoder_type="mcnet", gpu=0, seed=1337, image_size=[256, 256],
                     val_interval=200, labeled_num=7, use_graph=True)          # train_ours_2D.py:469-526


def _synthetic_loader(a):
    h, w = a["image_size"]
    lbs, ubs = a["labeled_bs"], a["batch_size"] - a["labeled_bs"]
    pool = [synthetic_batch(a["seed"] + i, lbs, ubs, h, w, a["num_classes"]) for i in range(8)]
    while True:
        for v, l in pool:
            yield {"image": v, "label": l}


def train(args, snapshot_path):
    a = dict(FLAG_DEFAULTS)
    a.update(args)
    os.makedirs(snapshot_path, exist_ok=True)
    log = logging.getLogger("chap_amd.train")
    log.setLevel(logging.INFO)
    fh = logging.FileHandler(os.path.join(snapshot_path, "log.txt"))
    log.addHandler(fh)
    device = torch.device("cuda", a["gpu"])
    torch.manual_seed(a["seed"])
    np.random.seed(a["seed"])
    model = net_factory(net_type=a["model"], in_chns=1, class_num=a["num_classes"], device=device, args=a)      # :240
    model.train()
    if a.get("dtype", "fp32") == "bf16":
        model.set_compute_dtype(torch.bfloat16)
    step = ChapStep(model, a)
    loader = a.get("trainloader") or _synthetic_loader(a)
    val = a.get("val_volumes")
    if val is None:
        vi, vl = synthetic_batch(a["seed"] + 4242, 8, 0, *a["image_size"], a["num_classes"])
        val = [(vi[:, 0].unsqueeze(0), vl.unsqueeze(0))]
    best, captured = 0.0, False

    def batches():
        """The epoch loop of the reference (:299-302, 459-463): `max_epoch = max_iterations // len(trainloader) + 1` passes
        over the loader, i.e. a finite loader is re-iterated until max_iterations is reached (the generator of the synthetic
        stand-in never ends; a one-shot iterator that runs dry ends the run, loudly)."""
        while True:
            n = 0
            for b in loader:
                n += 1
                yield b
            if n == 0:
                raise RuntimeError("chap_amd.train: the train loader yielded no batch (exhausted one-shot iterator or empty dataset) "
                                   "at iteration %d of %d" % (step.iter_num, a["max_iterations"]))

    gen = batches()
    sampled_batch = next(gen)
    while True:
        if a["use_graph"]:
            if not captured:
                step.capture(sampled_batch["image"].to(device), sampled_batch["label"].to(device))
                captured = True
                step.stage(sampled_batch["image"], sampled_batch["label"])
            out = step.replay()                      # the staged batch (host -> device copy done beside the previous iteration)
        else:
            out = step.step(sampled_batch["image"].to(device, non_blocking=True), sampled_batch["label"].to(device, non_blocking=True))
        it = step.iter_num
        if it % 50 == 0:                                                         # (:404 logs every iteration: a host sync each)
            log.info("iteration %d : bcp loss : %f vat loss : %f" % (it, sum(float(l[2]) for l in out["mix_losses"]), float(out["vat_loss"])))
        if it > 0 and it % a["val_interval"] == 0:                               # :407-456
            model.eval()
            metric = sum(np.array(test_single_volume(im, lb, model, classes=a["num_classes"], patch_size=a["image_size"],
                                                     model_type="logit_ensemble", device=device)) for im, lb in val) / len(val)
            performance = float(np.mean(metric, axis=0)[0])
            torch.save(model.state_dict(), os.path.join(snapshot_path, "latest.pth"))
            if performance > best:
                best = performance
                torch.save(model.state_dict(), os.path.join(snapshot_path, "{}_best_model.pth".format(a["model"])))
                with open(os.path.join(snapshot_path, "val.csv"), "a", newline="") as f:
                    csv.writer(f).writerow([time.strftime("%Y-%m-%d %H:%M:%S"), it, round(best, 4)])
            log.info("iteration %d : model1_mean_dice : %f" % (it, performance))
            model.train()
        if it >= a["max_iterations"]:
            break
        sampled_batch = next(gen)
        if a["use_graph"]:
            step.stage(sampled_batch["image"], sampled_batch["label"])      # travels while the iteration enqueued above runs
    log.removeHandler(fh)
    fh.close()
    return model
