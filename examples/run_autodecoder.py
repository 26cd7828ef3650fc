"""Articulated multi-instance training / test-time interpolation through the HIP path: the slice of the reference's
``run.py`` for ``--dataset_name sapien_multi --model vanilla_autodecoder`` (dataset items -> ``LitNeRF_AutoDecoder``
training_step / Adam + LR rule -> validation -> checkpoint -> the 19-pose test split with interpolated articulation codes
-> JPEGs + results.json), without Lightning, wandb or ``opt.py``.

    python examples/run_autodecoder.py --root_dir /data/sapien_multi/laptops --img_wh 320 240 --steps 100000
    python examples/run_autodecoder.py --synthetic /tmp/multi --img_wh 32 24 --steps 60        # self-contained demo

Reference flow: run.py:100-173; model_autodecoder.py:359-391 (setup), :393-477 (training_step), :588-605 (test_step with
``code_library(batch, is_test=True)``), :665-701 (test_epoch_end); datasets/sapien_multi.py (items)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def collate(item, dev):
    """What DataLoader(batch_size=1) does to a dataset item: a leading batch dimension on everything."""
    out = {}
    for k, v in item.items():
        t = torch.as_tensor(np.asarray(v)) if not isinstance(v, torch.Tensor) else v
        out[k] = t.unsqueeze(0).to(dev)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root_dir", default=None)
    ap.add_argument("--synthetic", default=None, help="write a small synthetic multi-instance tree here and train on it")
    ap.add_argument("--img_wh", type=int, nargs=2, default=(32, 24))
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--val_every", type=int, default=20)
    ap.add_argument("--exp_dir", default="ckpts/demo_autodecoder")
    ap.add_argument("--seed", type=int, default=0, help="torch / numpy / random seed (model init, ray batches, stratified draws)")
    args = ap.parse_args()
    import random as _random
    _random.seed(args.seed); np.random.seed(args.seed); torch.manual_seed(args.seed)

    import aon_amd  # noqa: F401
    from aon_amd import ops
    from aon_amd.datasets.sapien_multi import SapienDatasetMulti, write_synthetic_multi_scene
    from aon_amd.models.vanilla_nerf.model_autodecoder import LitNeRF_AutoDecoder
    from aon_amd.utils import save_checkpoint

    if args.synthetic:
        args.root_dir = write_synthetic_multi_scene(args.synthetic, n_instances=2, n_degrees=3, n_views=60, img_wh=tuple(args.img_wh))
    assert args.root_dir, "--root_dir or --synthetic"
    dev = torch.device("cuda:0")
    kw = dict(img_wh=tuple(args.img_wh), white_back=True, device=dev)
    train = SapienDatasetMulti(args.root_dir, "train", **kw)
    val = SapienDatasetMulti(args.root_dir, "val", **kw)
    test = SapienDatasetMulti(args.root_dir, "test_val", eval_inference="render", **kw)

    lit = LitNeRF_AutoDecoder({"chunk": 65536, "img_wh": tuple(args.img_wh), "run_max_steps": args.steps, "N_max_objs": len(train.ids),
                               "N_obj_code_length": 128}).to(dev)
    lit.setup(train)
    opt = lit.configure_optimizers()
    log, t0 = [], time.perf_counter()
    for step in range(args.steps):
        lit.fit_step(collate(train[step], dev), step, opt)   # zero_grad, training_step, backward, (DDP mean), LR rule + Adam
        if (step + 1) % args.val_every == 0 or step + 1 == args.steps:
            lit.validation_step(collate(val[0], dev), 0)
            rec = {"step": step + 1, "train_psnr_fine": lit.logged["train/psnr1"][-1], "val_psnr": lit.logged["val/psnr"][-1],
                   "val_psnr_obj": lit.logged["val/psnr_obj"][-1], "rays_per_s": (step + 1) * train.ray_batch_size / (time.perf_counter() - t0)}
            log.append(rec)
            print(json.dumps(rec), flush=True)
    os.makedirs(args.exp_dir, exist_ok=True)
    lit.finish_fit()   # the deferred check of the last data-parallel gradient exchange
    save_checkpoint(os.path.join(args.exp_dir, "last.ckpt"), lit, opt, epoch=0)
    # test split: 19 poses on the spheric path, articulation code i of the 19-row interpolated table (code_library.py:41-71)
    outs = [lit.test_step(collate(test[i], dev), i) for i in range(len(test))]
    psnr, psnr_obj = lit.test_epoch_end(outs, test.image_sizes, out_dir=os.path.join(args.exp_dir, "render"))
    print(json.dumps({"test_psnr": psnr["test"], "test_psnr_obj": psnr_obj["test"], "images": len(outs)}))
    return log, psnr


if __name__ == "__main__":
    main()
