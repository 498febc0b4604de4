"""`build_model(args, gpu_id=None) -> (model, args)`: drop-in for
recognition/time_interval_machine/models/build.py:5-64 (detection build.py:5-66).

Differences, all forced by the hardware target:
  * the model only runs on a GPU (the HIP path has no CPU fallback); `num_gpus == 0` builds the
    parameter container but calling it on CPU tensors raises;
  * `SyncBatchNorm.convert_sync_batchnorm` is dropped (TIM has no BatchNorm; build.py:47 is a no-op);
  * `num_gpus > 1` wraps the model in `tim_amd.dp.DataParallel` (one process per GPU, RCCL
    gradient all-reduce launched per layer bucket during the backward) instead of torch DDP.
"""
import torch

from .tim import TIM


def build_model(args, gpu_id=None):
    if torch.cuda.is_available():
        assert args.num_gpus <= torch.cuda.device_count(), "Cannot use more GPU devices than available"
    else:
        assert args.num_gpus == 0, "Cuda is not available. Please set `NUM_GPUS: 0 for running on CPUs."
    precision = getattr(args, "precision", "fp16")   # the fastest mode within 1e-3 of the fp32 logits
    # which of the reference's two `time_interval_machine` packages this is: the detection parser (detection/
    # time_interval_machine/utils/parser.py:37,43) is the one that defines --iou_threshold and spells the FFN argument
    # `feedfoward_scale` (det build.py:21-38);
    # an explicit args.variant, if somebody sets one, wins
    variant = getattr(args, "variant", None)
    if variant is None:
        variant = "detection" if hasattr(args, "iou_threshold") and hasattr(args, "feedfoward_scale") else "recognition"
    if variant == "detection":
        from .detection import TIM as DetTIM
        model = DetTIM(args.num_class, visual_input_dim=args.visual_input_dim,
                       audio_input_dim=args.audio_input_dim, feat_drop=args.feat_dropout,
                       seq_drop=args.seq_dropout, d_model=args.d_model,
                       feedfoward_scale=args.feedfoward_scale, nhead=args.nhead, num_layers=args.num_layers,
                       enc_dropout=args.enc_dropout, input_modality=args.model_modality,
                       data_modality=args.data_modality, num_feats=args.num_feats,
                       include_verb_noun=args.include_verb_noun, iou_threshold=args.iou_threshold,
                       label_smoothing=args.label_smoothing, precision=precision)
    else:
        model = TIM(args.num_class, visual_input_dim=args.visual_input_dim, audio_input_dim=args.audio_input_dim,
                    feat_drop=args.feat_dropout, seq_drop=args.seq_dropout, d_model=args.d_model,
                    feedforward_scale=args.feedforward_scale, nhead=args.nhead, num_layers=args.num_layers,
                    enc_dropout=args.enc_dropout, input_modality=args.model_modality,
                    data_modality=args.data_modality, num_feats=args.num_feats,
                    include_verb_noun=args.include_verb_noun, pool_features=args.apply_feature_pooling,
                    precision=precision)
    if args.num_gpus:
        cur_device = torch.cuda.current_device() if gpu_id is None else gpu_id
        model = model.cuda(device=cur_device)
    if args.num_gpus > 1:
        if args.workers < args.num_gpus or args.workers == 0:
            args.workers = 0
        else:
            args.workers = int((args.workers + args.num_gpus - 1) / args.num_gpus)
        from .dp import DataParallel
        model = DataParallel(model)
    return model, args
