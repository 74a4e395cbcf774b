"""Thin trainer: the counterpart of the reference's ``train_model.py`` (tf2.5/scripts/train_model.py) for the
MI355X-native M1 -- same UPPER_CASE flags and defaults (T:46-94), same model construction (T:189-207), compile (T:231),
callbacks (T:234-239) and ``fit`` call (T:253-259), on synthetic volumes or ``.npy`` files.

    python -m model.train_model --NAME run1 --NUM_EPOCHS 4 --UNET_PROBABILISTIC 1 --SYNTHETIC_SAMPLES 8
    python -m torch.distributed.run --nproc-per-node 8 -m model.train_model ...        (data parallel, one rank per GPU)

What differs from the reference, and why (its harness bugs are not reproduced):
  * data: the reference reads ``.npy`` paths from ``.xlsx`` sheets (data_generators.py:30-90; needs pandas+openpyxl+cv2 and
    data that is not shipped).  Here ``--TRAIN_NPY_DIR`` takes a directory of ``image_*.npy`` / ``label_*.npy`` pairs, and
    without it ``--SYNTHETIC_SAMPLES`` whitened-noise volumes with a ball lesion are generated (same I/O contract:
    ``({"image": x}, {"detection": y[, "KL": 0]})``, data_generators.py:79-88).  Augmentation (model/augmentations.py) is
    the reference's CPU input pipeline and out of scope.
  * multi-GPU: one process per GPU under torchrun + RCCL all-reduce (ddp.py) instead of ``--GPU_DEVICE_IDs`` +
    tf.distribute.MirroredStrategy inside one process (T:167-170); the flag is accepted and must agree with WORLD_SIZE.
  * ``--LR_MODE CLR`` refers to a CyclicLR the reference never imports (T:247-251): not offered.  ``--OPTIMIZER momentum``
    (T:121) and ``--LOSS_MODE region_boundary`` (T:125, CPU distance transforms) are outside the hot path: rejected loudly.
  * checkpoints are ``model_weights_NNN.npz`` (callbacks.py); the fold-finished test uses the intended file name
    (the reference formats a set literal into it, T:103).
"""
from __future__ import annotations

import argparse
import math
import os
import sys
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import callbacks as cbs
from . import ddp, initializers, losses, optim, unets


def _triples(v: Sequence, n: int) -> Tuple[Tuple[int, int, int], ...]:
    """``--UNET_STRIDES 1 1 1 1 2 2 ...`` (argparse ``nargs='+'`` of ints, T:73-75) or already a list of 3-tuples."""
    v = list(v)
    if v and isinstance(v[0], (tuple, list)):
        out = tuple(tuple(int(a) for a in t) for t in v)
    else:
        assert len(v) == 3 * n, f"expected {n} triples ({3 * n} integers), got {len(v)}"
        out = tuple(tuple(int(a) for a in v[3 * i:3 * i + 3]) for i in range(n))
    assert len(out) == n
    return out


def build_parser() -> argparse.ArgumentParser:
    """Flags and defaults of train_model.py:43-97 (names kept verbatim)."""
    prsr = argparse.ArgumentParser(description='Command Line Arguments for Training Script')
    # Dataset Definition (T:46-64)
    prsr.add_argument('--TRAIN_OBJ', type=str, default='lesion', help="Training Objective: 'zonal'/'lesion'")
    prsr.add_argument('--NAME', type=str, default='diagnosis/', help='Path to Load/Store Model Weights and Performance Metrics')
    prsr.add_argument('--NUM_EPOCHS', type=int, default=250, help="Number of Training Epochs")
    prsr.add_argument('--FOLDS', type=int, default=[0, 1, 2, 3, 4], nargs='+', help="Folds Selected For Training")
    prsr.add_argument('--TRAIN_XLSX_PREFIX', type=str, default='./models/2021/medneurips2021/data_feed/prostateX_200_train-fold-')
    prsr.add_argument('--VALID_XLSX_PREFIX', type=str, default='./models/2021/medneurips2021/data_feed/prostateX_200_valid-fold-')
    prsr.add_argument('--WEIGHTS_DIR', type=str, default='./models/2021/medneurips2021/weights/', help="Path to Load/Store Model Weights")
    prsr.add_argument('--METRICS_DIR', type=str, default='./models/2021/medneurips2021/weights/')
    prsr.add_argument('--USE_PRETRAINED_WEIGHTS', type=str, default=False, help="Path to Pretrained Weights or 'False' (Optional)")
    prsr.add_argument('--FREEZE_LAYERS', type=int, default=9999, help="Freeze First N Layers [e.g. 184]; 9999 = none")
    prsr.add_argument('--WEIGHTS_MIN_EPOCH', type=int, default=5, help="Minimum Epoch to Start Exporting Weights")
    prsr.add_argument('--VALIDATE_PER_N_EPOCHS', type=int, default=5)
    prsr.add_argument('--STORE_WEIGHTS_PER_N_EPOCHS', type=int, default=5, help="Store Weights Every N Epochs")
    prsr.add_argument('--WEIGHTS_OVERWRITE', type=int, default=0, help="Store All Weights or Most Recent One")
    prsr.add_argument('--VALIDATE_MIN_EPOCH', type=int, default=5)
    prsr.add_argument('--SHOW_SUMMARY', type=int, default=0, help="Display Overview")
    prsr.add_argument('--RESUME_TRAIN', type=int, default=0, help="Enable Resume Training")
    prsr.add_argument('--CACHE_TDS_PATH', type=str, default=None)
    prsr.add_argument('--GPU_DEVICE_IDs', type=str, default="0", help="GPUs Available for Computation")
    # U-Net Hyperparameters (T:67-80)
    prsr.add_argument('--UNET_DENSE_SKIP', type=int, default=0)
    prsr.add_argument('--UNET_DEEP_SUPERVISION', type=int, default=0)
    prsr.add_argument('--UNET_PROBABILISTIC', type=int, default=0)
    prsr.add_argument('--UNET_PROBA_LATENT_DIMS', type=int, default=[3, 2, 1, 0], nargs='+')
    prsr.add_argument('--UNET_PROBA_ITER', type=int, default=1)
    prsr.add_argument('--UNET_FEATURE_CHANNELS', type=int, default=[16, 32, 64, 128, 256], nargs='+')
    prsr.add_argument('--UNET_STRIDES', type=int, default=[(1, 1, 1), (1, 2, 2), (1, 2, 2), (2, 2, 2), (2, 2, 2)], nargs='+')
    prsr.add_argument('--UNET_KERNEL_SIZES', type=int, default=[(1, 3, 3), (1, 3, 3), (3, 3, 3), (3, 3, 3), (3, 3, 3)], nargs='+')
    prsr.add_argument('--UNET_ATT_SUBSAMP', type=int, default=[(1, 1, 1), (1, 1, 1), (1, 1, 1), (1, 1, 1)], nargs='+')
    prsr.add_argument('--UNET_SE_REDUCTION', type=int, default=[8, 8, 8, 8, 8], nargs='+')
    prsr.add_argument('--UNET_KERNEL_REGULARIZER_L2', type=float, default=1e-5)
    prsr.add_argument('--UNET_BIAS_REGULARIZER_L2', type=float, default=1e-5)
    prsr.add_argument('--UNET_DROPOUT_MODE', type=str, default="monte-carlo")
    prsr.add_argument('--UNET_DROPOUT_RATE', type=float, default=0.50)
    # Training Hyperparameters (T:83-94)
    prsr.add_argument('--BATCH_SIZE', type=int, default=2, help="Batch Size (global: split over the GPUs, T:170)")
    prsr.add_argument('--BASE_LR', type=float, default=1e-3)
    prsr.add_argument('--LR_MODE', type=str, default="CALR", help="'CALR' (CosineDecayRestarts) / 'CONST'")
    prsr.add_argument('--CALR_PARAMS', type=float, default=[2.00, 1.00, 1e-3], nargs='+', help="'CosineDecayRestarts': t_mul, m_mul, alpha")
    prsr.add_argument('--CLR_PARAMS', type=float, default=[5e-5, 1.00, 1.25], nargs='+')
    prsr.add_argument('--OPTIMIZER', type=str, default="adam")
    prsr.add_argument('--LOSS_MODE', type=str, default="distribution_focal")
    prsr.add_argument('--FOCAL_LOSS_ALPHA', type=float, default=[1.00, 1.00], nargs='+')
    prsr.add_argument('--FOCAL_LOSS_GAMMA', type=float, default=2.0)
    prsr.add_argument('--DSC_BD_LOSS_WEIGHTS', type=float, default=[0.50, 0.50], nargs='+')
    prsr.add_argument('--ELBO_LOSS_PARAMS', type=float, default=[10], nargs='+')
    prsr.add_argument('--AUGM_PARAMS', type=float, default=[1.00, 0.25, 0.15, 10.0, True, 1.20, 0.10, 0.025, True, [0.50, 1.50]], nargs='+')
    # This build's data source (the reference's .xlsx sheets point at data that is not shipped)
    prsr.add_argument('--TRAIN_NPY_DIR', type=str, default=None, help="directory of image_*.npy (D,H,W,C) / label_*.npy (D,H,W) pairs")
    prsr.add_argument('--SYNTHETIC_SAMPLES', type=int, default=8, help="training samples per fold when no .npy directory is given")
    prsr.add_argument('--IMAGE_SPATIAL_DIMS', type=int, default=[20, 160, 160], nargs=3, help="(D,H,W) of synthetic volumes")
    prsr.add_argument('--COMPUTE_DTYPE', type=str, default="bf16", choices=["bf16", "fp32"], help="activation storage type")
    prsr.add_argument('--SEED', type=int, default=0)
    return prsr


# ---- data -----------------------------------------------------------------------------------------------------
def synthetic_case(rng: np.random.Generator, dims, image_channels: int, num_classes: int):
    """One whitened-noise volume (preprocess.py:29-39 whitens each channel) with a ball 'lesion' (radius 6 voxels), its
    one-hot label (data_generators.py:72) -- what bench.py trains on."""
    D, H, W = dims
    image = rng.standard_normal((D, H, W, image_channels)).astype(np.float32)
    zz, yy, xx = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    c = [rng.integers(1, max(2, D - 1)), rng.integers(6, max(7, H - 6)), rng.integers(6, max(7, W - 6))]
    ball = (((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) <= 36).astype(np.float32)
    cls = [1.0 - ball] + [ball if k == 1 else np.zeros_like(ball) for k in range(1, num_classes)]
    return image, np.stack(cls, axis=-1)


def custom_data_generator(cases: List[Tuple[np.ndarray, np.ndarray]], probabilistic=False, mode='train') -> Iterator:
    """The reference generator's contract (data_generators.py:29-88) over in-memory (image, one-hot label) cases: cycles
    for ever; probabilistic -> the posterior's label channels are appended to the image (zeros outside training) and a
    zero "KL" target is added."""
    i = 0
    while True:
        if (i + 1) > len(cases):
            i = 0
        image, label = cases[i]
        i += 1
        postq_lbl = np.zeros_like(label)[..., 1:] if mode in ('test', 'valid') else label.copy()[..., 1:]
        if probabilistic:
            yield {"image": np.concatenate((image.copy(), postq_lbl), axis=-1)}, {"detection": label.copy(),
                                                                                   "KL": np.zeros(shape=label.shape, dtype=np.float32)}
        else:
            yield {"image": image.copy()}, {"detection": label.copy()}


def load_npy_cases(path: str, num_classes: int) -> List[Tuple[np.ndarray, np.ndarray]]:
    names = sorted(f for f in os.listdir(path) if f.startswith("image_") and f.endswith(".npy"))
    if not names:
        raise FileNotFoundError(f"no image_*.npy under {path}")
    cases = []
    for f in names:
        image = np.load(os.path.join(path, f)).astype(np.float32)
        lab = np.load(os.path.join(path, f.replace("image_", "label_"))).astype(np.int64)
        cases.append((image, np.stack([(lab == k) for k in range(num_classes)], axis=-1).astype(np.float32)))
    return cases


def batches(gen: Iterator, batch_size: int, device, rank: int = 0, world: int = 1) -> Iterator:
    """``dataset.batch(BATCH_SIZE)`` (T:182); under data parallelism every rank draws the same global batch and keeps its
    shard (ddp.shard_batch), so the union over ranks is the reference's batch."""
    mine = ddp.shard_batch(batch_size, rank, world)
    while True:
        xs, ys = zip(*[next(gen) for _ in range(batch_size)])
        bx = {k: torch.from_numpy(np.stack([x[k] for x in xs])[mine.start:mine.stop]).to(device) for k in xs[0]}
        by = {k: torch.from_numpy(np.stack([y[k] for y in ys])[mine.start:mine.stop]).float().to(device) for k in ys[0]}
        yield bx, by


def claim_fold_dir(fold_dir: str, resume: bool, rank: int = 0, world: int = 1) -> None:
    """T:226-229 under one process per GPU: rank 0 alone looks at the target folder and creates it, every rank learns the
    verdict from a broadcast and only then continues -- a rank that arrives after rank 0 has made the folder must not take it
    for a left-over of an earlier run ("Target Folder Already Exists")."""
    flag = [False]
    if rank == 0:
        flag[0] = (not resume) and os.path.exists(fold_dir)
        if not flag[0]:
            os.makedirs(fold_dir, exist_ok=True)
    if world > 1:
        import torch.distributed as dist
        dist.broadcast_object_list(flag, src=0)      # (also the barrier: nobody passes before rank 0 has decided)
    if flag[0]:
        raise Exception("Target Folder Already Exists! Either Remove It or Enable 'RESUME_TRAIN'.")


# ---- one fold ---------------------------------------------------------------------------------------------------
def train_fold(args, f: int, device, rank: int = 0, world: int = 1):
    fold_dir = os.path.join(args.WEIGHTS_DIR + args.NAME, 'F' + str(f + 1))
    # Verify whether training had completed (T:103, with the file name it means)
    if os.path.isfile(cbs.weights_path(fold_dir, args.NUM_EPOCHS)):
        print(f"Fold {f + 1}: final weights exist, skipping.", flush=True)
        return None

    NUM_CLASSES = 2 if args.TRAIN_OBJ == 'lesion' else 3                        # T:152
    IMAGE_NUM_CHANNELS = 3 if args.TRAIN_OBJ == 'lesion' else 1                 # T:151
    prob = bool(args.UNET_PROBABILISTIC)
    rng = np.random.default_rng(args.SEED + 1000 * f)
    if args.TRAIN_NPY_DIR:
        cases = load_npy_cases(args.TRAIN_NPY_DIR, NUM_CLASSES)
    else:
        cases = [synthetic_case(rng, tuple(args.IMAGE_SPATIAL_DIMS), IMAGE_NUM_CHANNELS, NUM_CLASSES)
                 for _ in range(args.SYNTHETIC_SAMPLES)]
    TRAIN_DATA_SAMPLES = len(cases)
    IMAGE_SPATIAL_DIMS = tuple(int(v) for v in cases[0][0].shape[:3])           # T:150
    steps_per_epoch = int(math.ceil(TRAIN_DATA_SAMPLES / args.BATCH_SIZE))      # T:255

    # Cosine annealing with warm restarts (T:112-117)
    if args.LR_MODE == 'CALR':
        BASE_LR = optim.CosineDecayRestarts(initial_learning_rate=args.BASE_LR, first_decay_steps=steps_per_epoch * args.NUM_EPOCHS,
                                            t_mul=args.CALR_PARAMS[0], m_mul=args.CALR_PARAMS[1], alpha=args.CALR_PARAMS[2])
    elif args.LR_MODE == 'CONST':
        BASE_LR = args.BASE_LR
    else:
        raise NotImplementedError("--LR_MODE CLR needs the CyclicLR callback the reference never imports (train_model.py:247-251)")
    # Optimizer (T:120-121)
    if args.OPTIMIZER != 'adam':
        raise NotImplementedError("only --OPTIMIZER adam (Adam amsgrad, train_model.py:120) runs on the fused HIP optimiser")
    OPTIMIZER_SET = optim.Adam(learning_rate=BASE_LR, amsgrad=True)
    # Losses (T:124-131)
    if args.LOSS_MODE != 'distribution_focal':
        raise NotImplementedError("--LOSS_MODE region_boundary needs CPU distance transforms (losses.py:66-130): out of scope")
    if len(args.FOCAL_LOSS_ALPHA) != NUM_CLASSES:                               # T:154-155
        raise Exception("Number of Class Weights Declared in Loss Function != Number of Classes in Labels/Loss Objective")
    LOSSES = [losses.Focal(alpha=args.FOCAL_LOSS_ALPHA, gamma=args.FOCAL_LOSS_GAMMA).loss]
    LOSS_WEIGHTS = [1.00]
    if prob:
        LOSSES += [losses.EvidenceLowerBound().loss]
        LOSS_WEIGHTS += [args.ELBO_LOSS_PARAMS[0]]
        IMAGE_NUM_CHANNELS += NUM_CLASSES - 1                                   # T:157
    assert np.mod(args.BATCH_SIZE, world) == 0, \
        'Batch size (%d) should be a multiple of the number of GPUs (%d).' % (args.BATCH_SIZE, world)          # T:170

    train_gen = batches(custom_data_generator(cases, probabilistic=prob, mode='train'), args.BATCH_SIZE, device, rank, world)

    # U-Net definition (T:189-207)
    unets.network_blocks.set_init_seed(args.SEED)
    unet_model = unets.networks.M1(input_spatial_dims=IMAGE_SPATIAL_DIMS,
                                   input_channels=IMAGE_NUM_CHANNELS,
                                   num_classes=NUM_CLASSES,
                                   filters=tuple(args.UNET_FEATURE_CHANNELS),
                                   dropout_rate=args.UNET_DROPOUT_RATE,
                                   strides=_triples(args.UNET_STRIDES, 5),
                                   kernel_sizes=_triples(args.UNET_KERNEL_SIZES, 5),
                                   dropout_mode=args.UNET_DROPOUT_MODE,
                                   se_reduction=tuple(args.UNET_SE_REDUCTION),
                                   att_sub_samp=_triples(args.UNET_ATT_SUBSAMP, 4),
                                   probabilistic=prob,
                                   prob_latent_dims=tuple(args.UNET_PROBA_LATENT_DIMS),
                                   dense_skip=bool(args.UNET_DENSE_SKIP),
                                   deep_supervision=bool(args.UNET_DEEP_SUPERVISION),
                                   summary=bool(args.SHOW_SUMMARY),
                                   bias_initializer=initializers.TruncatedNormal(mean=0.0, stddev=0.001),
                                   bias_regularizer=initializers.l2(args.UNET_BIAS_REGULARIZER_L2),
                                   kernel_initializer=initializers.Orthogonal(gain=1.0),
                                   kernel_regularizer=initializers.l2(args.UNET_KERNEL_REGULARIZER_L2)).to(device)
    dtype = torch.bfloat16 if args.COMPUTE_DTYPE == "bf16" else torch.float32

    def configure(model):
        """Per-run state that lives outside the weight file: storage type, this rank's dropout stream, frozen layers (T:210-215).
        Applied again whenever the model object is replaced by a loaded one (pre-trained weights, resume)."""
        model.set_compute_dtype(dtype)
        model.seed_dropout(args.SEED + 2 + rank)
        if args.FREEZE_LAYERS != 9999:
            for layer in model.layers[:args.FREEZE_LAYERS]:
                for p in layer.parameters():
                    p.requires_grad_(False)
        return model
    configure(unet_model)

    # Load pre-trained weights (T:218-219)
    if str(args.USE_PRETRAINED_WEIGHTS) != 'False':
        unet_model = configure(unets.networks.M1.load(path=args.USE_PRETRAINED_WEIGHTS).to(device))
    # Number of layers / frozen layers (T:210-215)
    print("Number of Model Layers: ", len(unet_model.layers), flush=True)
    if args.FREEZE_LAYERS != 9999:
        print("Trainable Layers: ", len(unet_model.layers) - args.FREEZE_LAYERS, flush=True)

    # Restart / resume (T:222-229).  The resumed optimiser starts from zero moments (the reference's weight files hold no
    # optimiser state either, callbacks.py:62,209); the learning-rate schedule continues at init_epoch * steps_per_epoch.
    if bool(args.RESUME_TRAIN):
        loaded, init_epoch = cbs.ResumeTraining(model=unet_model, weights_dir=fold_dir)
        if loaded is not unet_model:
            unet_model = configure(loaded)
    else:
        init_epoch = 0
    claim_fold_dir(fold_dir, bool(args.RESUME_TRAIN), rank, world)

    # Compile (T:231); the schedule continues where the resumed run stopped
    OPTIMIZER_SET.iterations = init_epoch * steps_per_epoch
    unet_model.compile(optimizer=OPTIMIZER_SET, loss=LOSSES, loss_weights=LOSS_WEIGHTS)
    if world > 1:
        OPTIMIZER_SET.attach_reducer(ddp.GradReducer(world_size=world))

    # Callbacks (T:234-239) and training (T:253-259)
    callbacks = [cbs.WeightsSaver(unet_model, weights_overwrite=bool(args.WEIGHTS_OVERWRITE), weights_dir=fold_dir,
                                  min_epoch=args.WEIGHTS_MIN_EPOCH, weights_num_epochs=args.STORE_WEIGHTS_PER_N_EPOCHS,
                                  init_epoch=init_epoch, rank=rank)]
    history = unet_model.fit(x=train_gen, epochs=args.NUM_EPOCHS, steps_per_epoch=steps_per_epoch, initial_epoch=init_epoch,
                             verbose=2 if rank == 0 else 0, callbacks=callbacks, use_multiprocessing=True)
    return unet_model, history, callbacks[0]


def main(argv: Optional[Sequence[str]] = None):
    args, _ = build_parser().parse_known_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("train_model needs a GPU: the HIP extension is the only compute path of this package")
    world = ddp.init_process_group_from_env()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ids = [s for s in str(args.GPU_DEVICE_IDs).split(",") if s != ""]
    if world == 1 and len(ids) > 1:
        raise SystemExit(f"--GPU_DEVICE_IDs names {len(ids)} GPUs: launch one process per GPU with "
                         f"`python -m torch.distributed.run --nproc-per-node {len(ids)} -m model.train_model ...`")
    torch.cuda.set_device(local if world > 1 else int(ids[0]) if ids else 0)
    device = torch.device("cuda", torch.cuda.current_device())
    out = []
    for f in args.FOLDS:
        out.append(train_fold(args, f, device, rank, world))
        if world > 1:
            import torch.distributed as dist
            dist.barrier()               # the next fold's folder checks must see this fold's files from every rank's view
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main(sys.argv[1:])
