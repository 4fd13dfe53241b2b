"""Training driver: per-epoch scalar schedules, the optimisation step, best-checkpoint saving, early stop.

Mirrors reference method/train.py:52-247 for the part that touches the hot path; logging to
TensorBoard, result directories and code zips are out of scope.  With torch.distributed initialised the step
becomes data parallel: local in-batch losses, one flat gradient all-reduce (dist.all_reduce_flat)."""
import logging
import math

import torch
from torch.utils.data import DataLoader

from .data import collate_train
from .eval import eval_epoch
from .optimization import BertAdam

logger = logging.getLogger(__name__)


def _decay(kind, initial, floor, epoch_i, opt, sigmoid_k):
    """alpha / belta schedules (train.py:85-125)."""
    if kind == "exp":
        return max(initial * (opt.exponential_k ** epoch_i), floor)
    if kind == "linear":
        return max(initial + ((floor - initial) / opt.n_epoch) * epoch_i, floor)
    if kind == "sigmoid":
        return max(initial * (sigmoid_k / (sigmoid_k + math.exp(epoch_i * 100 / sigmoid_k))), floor)
    if kind == "cosine":
        return max(floor + 0.5 * (initial - floor) * (1 + math.cos(math.pi * epoch_i / opt.n_epoch)), floor)
    if kind == "None":
        return initial
    raise AssertionError(kind)


def epoch_schedules(opt, epoch_i):
    """(kd weight, alpha, belta) for an epoch; None where the reference leaves the attribute untouched."""
    weight = None
    d = getattr(opt, "distill_loss_decay", None)
    if d is not None:
        assert d in ["exp", "sigmoid", "linear", "None"]
        if d == "exp":
            weight = opt.exponential_k ** epoch_i                                           # train.py:76
        elif d == "linear":
            weight = max(opt.linear_k * epoch_i + opt.linear_b, 0.05)
        elif d == "sigmoid":
            weight = opt.sigmoid_k / (opt.sigmoid_k + math.exp(epoch_i * 100 / opt.sigmoid_k))
        else:
            weight = 1
    sk = opt.selfDistil_sigmoid_k
    alpha = belta = None
    if getattr(opt, "alpha_decay", None) is not None:
        assert opt.alpha_decay in ["exp", "sigmoid", "linear", "cosine", "None"]
        alpha = _decay(opt.alpha_decay, opt.alpha, 0.0, epoch_i, opt, sk)                   # min_alpha is 0 either way (:89-92)
    if getattr(opt, "belta_decay", None) is not None:
        assert opt.belta_decay in ["exp", "sigmoid", "linear", "cosine", "None"]
        belta = _decay(opt.belta_decay, opt.belta, 0 if opt.belta < 0.5 else 0.5, epoch_i, opt, sk)   # :109-113
    return weight, alpha, belta


def make_optimizer(model, opt, steps_per_epoch):
    """BertAdam over two groups: weight decay 0.01 except biases and LayerNorm parameters (train.py:203-213)."""
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    named = list(model.named_parameters())
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    return BertAdam(groups, lr=opt.lr, weight_decay=opt.wd, warmup=opt.lr_warmup_proportion,
                    t_total=steps_per_epoch * opt.n_epoch, schedule="warmup_linear")


DDP_MIN_WORLD = 2      # tests set 1 to drive the all-reduce branch with a one-rank group


def train_step(model, batch, optimizer, opt):
    """zero_grad / forward / backward / [global clip] / step (train.py:141-151).  Returns (loss, loss_dict)."""
    optimizer.zero_grad()
    loss, loss_dict = model(batch)
    loss.backward()
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() >= DDP_MIN_WORLD:
        from . import dist as ddist
        optimizer.fp.rebind_grads()
        ddist.all_reduce_flat(optimizer.fp.grad)
    if getattr(opt, "grad_clip", -1) != -1:
        torch.nn.utils.clip_grad_norm_(model.parameters(), opt.grad_clip)
    optimizer.step()
    return loss, loss_dict


def train_epoch(model, train_loader, optimizer, opt, epoch_i, training=True):
    """One epoch (train.py:52-183).  Returns the mean of every loss entry."""
    from .data import host_threads
    with host_threads():
        return _train_epoch(model, train_loader, optimizer, opt, epoch_i, training)


def _train_epoch(model, train_loader, optimizer, opt, epoch_i, training=True):
    model.train(mode=training)
    if opt.hard_negative_start_epoch != -1 and epoch_i >= opt.hard_negative_start_epoch:
        model.set_hard_negative(True, opt.hard_pool_size)
    weight, alpha, belta = epoch_schedules(opt, epoch_i)
    if weight is not None:
        model.weight = weight
    if alpha is not None:
        model.alpha = alpha
    if belta is not None:
        model.belta = belta
    logger.info(f"Epoch {epoch_i}, Alpha: {model.alpha}, belta: {model.belta}")
    sums, n = {}, 0
    for batch_idx, batch in enumerate(train_loader):
        batch = {k: (v.to(opt.device, non_blocking=True) if k != "text_labels" else v) for k, v in batch.items()}
        if training:
            _, loss_dict = train_step(model, batch, optimizer, opt)
        else:
            with torch.no_grad():
                _, loss_dict = model(batch)
        for k, v in loss_dict.items():
            sums[k] = sums.get(k, 0.0) + float(v.detach() if torch.is_tensor(v) else v)
        n += 1
        if getattr(opt, "debug", False) and batch_idx == 3:
            break
    return {k: v / max(n, 1) for k, v in sums.items()}


def save_checkpoint(model, epoch_i, path):
    """{"model", "model_cfg", "epoch"} (train.py:231-235), loadable by the reference's setup_model and ours."""
    torch.save({"model": model.state_dict(), "model_cfg": model.config, "epoch": epoch_i}, path)


def load_checkpoint(path, opt, map_location=None):
    """Counterpart of setup_model (eval.py:266-283)."""
    from .model import DLDKD
    ck = torch.load(path, map_location=map_location, weights_only=False)
    model = DLDKD(ck["model_cfg"], opt)
    model.load_state_dict(ck["model"])
    return model, ck["epoch"]


def train(model, train_dataset, val_video_dataset, val_text_dataset, opt):
    """Epoch loop with eval after each epoch, best-checkpoint saving and early stop (train.py:191-247)."""
    model.to(opt.device)
    loader = DataLoader(train_dataset, batch_size=opt.bsz, shuffle=True, pin_memory=opt.pin_memory,
                        num_workers=opt.num_workers, collate_fn=collate_train)
    optimizer = make_optimizer(model, opt, len(loader))
    best, es_cnt = 0.0, 0
    history = []
    for epoch_i in range(-1 if getattr(opt, "eval_untrained", False) else 0, opt.n_epoch):
        losses = train_epoch(model, loader, optimizer, opt, epoch_i, training=True) if epoch_i > -1 else {}
        with torch.no_grad():
            rsum = eval_epoch(model, val_video_dataset, val_text_dataset, opt)
        history.append((epoch_i, losses, rsum))
        if rsum > best:
            best, es_cnt = rsum, 0
            if getattr(opt, "ckpt_filepath", None):
                save_checkpoint(model, epoch_i, opt.ckpt_filepath)
        else:
            es_cnt += 1
            if opt.max_es_cnt != -1 and es_cnt > opt.max_es_cnt:
                break
        if getattr(opt, "debug", False):
            break
    return history
