"""TIM detection training steps on the MI355X path, on synthetic data - the loss assembly of
detection/scripts/train.py:205-349 with tim_amd: dense query pyramid + IoU labelling inside the model
(`tim_amd.detection.TIM`, det models/tim.py:140-270), focal classification loss with IoU row weights and 1-D DIoU
regression loss (`tim_amd.losses`, det models/helpers/losses/{sigmoid,iou,loss}.py), AdamW.

    python examples/train_detection_synthetic.py [--steps 20] [--batch 4]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tim_amd import losses, synth  # noqa: E402
from tim_amd.config import named_config  # noqa: E402
from tim_amd.detection import TIM  # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--lr", type=float, default=2e-3)
    args = ap.parse_args(argv)
    dev = torch.device("cuda", 0)
    cfg = named_config("tiny")
    cfg.variant = "detection"
    nf, B, ngt = cfg.num_feats, args.batch, 3
    model = TIM(cfg.num_class, visual_input_dim=cfg.visual_input_dim, audio_input_dim=cfg.audio_input_dim, d_model=cfg.d_model,
                nhead=cfg.nhead, num_layers=cfg.num_layers, num_feats=nf, feat_drop=0.1, seq_drop=0.1, enc_dropout=0.1,
                precision=args.precision)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, seed=0).items()})
    model = model.to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=args.lr, weight_decay=1e-4)
    rs = np.random.RandomState(0)
    inp = {k: torch.from_numpy(v).to(dev) for k, v in synth.make_inputs(cfg, B, 0, 0, seed=5).items()}
    seg = lambda: torch.from_numpy(np.sort(rs.rand(B, ngt, 2), axis=-1).astype(np.float32)).to(dev)
    vc, ac = cfg.num_class[0], cfg.num_class[1]
    ri = lambda hi: torch.from_numpy(rs.randint(0, hi, (B, ngt))).to(dev)
    target = {"v_gt_segments": seg(), "a_gt_segments": seg(), "verb": ri(vc[0]), "noun": ri(vc[1]), "action": ri(vc[2]),
              "class_id": ri(ac)}
    # the loop's EMA normaliser (det train.py:230) as a device scalar the loss kernels advance in place: no host read-back per
    # step (the reference's `max(num_pos, 1)` synchronises), and a captured step (tim_amd.graph.GraphedStep) advances it per replay
    normaliser, hist = torch.full((), 100.0, dtype=torch.float32, device=dev), []
    for step in range(args.steps):
        output, offsets, labels, _, ious = model([inp["visual"], inp["audio"]], "encoder", inp["times"], target, label_queries=True)
        loss = None
        for m, (cls_ids, lab) in enumerate((((0, 1, 2), labels[0]), ((3,), [labels[1]]))):   # visual, audio
            # one modality side of det train.py:222-349 in a handful of launches: valid_cls = iou >= 0, row weight = iou below
            # the threshold ? 1 : iou, positives = offsets != inf, focal sums of the side's heads / (heads * normaliser) + lambda_reg *
            # DIoU sum of the positive rows / normaliser (lambda_reg = 1 here)
            side = losses.detection_side_loss([output[0][c] for c in cls_ids], lab, output[1][m], offsets[m], ious[m], normaliser,
                                              model.iou_threshold, lambda_reg=1.0, momentum=0.9)
            loss = side if loss is None else loss + side
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        hist.append(loss.item())
        if step % 5 == 0 or step == args.steps - 1:
            print("step %3d  loss %.4f" % (step, hist[-1]), flush=True)
    return hist


if __name__ == "__main__":
    main()
