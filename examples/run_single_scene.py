"""Single-scene training / evaluation through the HIP path: the slice of the reference's ``run.py`` that touches the
rendered path (dataset -> ray batches -> ``LitNeRF.training_step`` / Adam + LR rule -> ``validation_step`` ->
checkpoint -> ``test_step`` / ``test_epoch_end``), without Lightning, wandb or the CLI of ``opt.py``.

    python examples/run_single_scene.py --root_dir /data/sapien/laptop --img_wh 640 480 --steps 20000
    python examples/run_single_scene.py --synthetic /tmp/scene --img_wh 64 48 --steps 300      # self-contained demo

Reference flow: run.py:100-173 (Trainer.fit / Trainer.test), model.py:245-294 (setup, training_step),
model.py:421-448 (dataloaders: 2048-ray shuffled batches, one image per validation/test item)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root_dir", default=None)
    ap.add_argument("--synthetic", default=None, help="write a small synthetic scene here (reference on-disk format) and train on it")
    ap.add_argument("--img_wh", type=int, nargs=2, default=(64, 48))
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=2048)          # model.py:426
    ap.add_argument("--val_every", type=int, default=100)
    ap.add_argument("--exp_dir", default="ckpts/demo")
    ap.add_argument("--resume", default=None)
    ap.add_argument("--seed", type=int, default=0, help="torch / numpy / random seed (model init, ray batches, stratified draws)")
    args = ap.parse_args()
    import random as _random
    _random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)

    import aon_amd  # noqa: F401
    from aon_amd.datasets.sapien import SapienDataset, write_synthetic_scene
    from aon_amd.models.vanilla_nerf.model import LitNeRF
    from aon_amd.utils import load_checkpoint, save_checkpoint

    if args.synthetic:
        args.root_dir = write_synthetic_scene(args.synthetic, n_train=8, n_val=1, img_wh=tuple(args.img_wh))
    assert args.root_dir, "--root_dir or --synthetic"
    dev = torch.device("cuda:0")
    from aon_amd import ops
    train = SapienDataset(args.root_dir, "train", tuple(args.img_wh), white_back=True, device=dev)
    val = SapienDataset(args.root_dir, "val", tuple(args.img_wh), white_back=True, device=dev)
    test = SapienDataset(args.root_dir, "test", tuple(args.img_wh), white_back=True, eval_inference="render", device=dev)

    lit = LitNeRF({"chunk": 65536, "img_wh": tuple(args.img_wh), "run_max_steps": args.steps},
                  near=train.near, far=train.far, white_bkgd=True).to(dev)
    opt = lit.configure_optimizers()
    if args.resume:
        load_checkpoint(args.resume, lit, opt)
    gen = torch.Generator(device=dev).manual_seed(0)
    t0, step = time.perf_counter(), lit.global_step
    log = []
    while step < args.steps:
        for batch in train.train_batches(args.batch, generator=gen):
            lit.fit_step({k: v.unsqueeze(0) for k, v in batch.items()}, step, opt)
            step = lit.global_step
            if step % args.val_every == 0 or step == args.steps:
                lit.validation_step({k: v.unsqueeze(0) for k, v in val[0].items()}, 0)
                rec = {"step": step, "train_psnr_fine": lit.logged["train/psnr1"][-1], "val_psnr": lit.logged["val/psnr"][-1],
                       "lr": opt.param_groups[0]["lr"], "rays_per_s": step * args.batch / (time.perf_counter() - t0)}
                log.append(rec)
                print(json.dumps(rec), flush=True)
            if step >= args.steps:
                break
    os.makedirs(args.exp_dir, exist_ok=True)
    lit.finish_fit()   # the deferred check of the last data-parallel gradient exchange
    save_checkpoint(os.path.join(args.exp_dir, "last.ckpt"), lit, opt, epoch=0)
    outs = [lit.test_step({k: v.unsqueeze(0) for k, v in test[i].items()}, i) for i in range(len(test))]
    psnr, psnr_obj = lit.test_epoch_end(outs, test.image_sizes, out_dir=os.path.join(args.exp_dir, "render"))
    print(json.dumps({"test_psnr": psnr["test"], "test_psnr_obj": psnr_obj["test"], "ckpt": os.path.join(args.exp_dir, "last.ckpt")}))
    return log, psnr


if __name__ == "__main__":
    main()
