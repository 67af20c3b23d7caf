"""Generates tests/golden/train_ref.npz by running the REFERENCE's own optimizer wrapper and schedule from source:
    det3d/solver/fastai_optim.py (OptimWrapper, true_wd) as built by det3d/torchie/apis/train_sessd.py:169-175,
    det3d/solver/learning_schedules_fastai.py (OneCycle, config.py:260),
    torch.nn.utils.clip_grad_norm_ as called by det3d/torchie/trainer/hooks/optimizer.py:50-53,
    and the two EMA lines of det3d/torchie/trainer/trainer_sessd.py:315-318 (the Trainer class itself needs the whole
    det3d import tree, so those two lines are restated here verbatim in effect).
Run in the build container only (needs /root/reference):  python tests/golden/make_golden_train.py
Shim: `collections.Iterable` (removed in Python 3.10) is aliased for the import of fastai_optim.py."""
import collections
import collections.abc
import importlib.util
import os
import sys
from functools import partial

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load_ref(relpath, modname):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def flat(params):
    return np.concatenate([p.detach().reshape(-1).numpy() for p in params]).astype(np.float32)


def main():
    assert os.path.isdir(REF)
    collections.Iterable = collections.abc.Iterable
    fo = load_ref("det3d/solver/fastai_optim.py", "ref_fastai_optim")
    ls = load_ref("det3d/solver/learning_schedules_fastai.py", "ref_lr_sched")
    torch.manual_seed(0)

    def make():
        return nn.Sequential(nn.Conv2d(3, 8, 3, bias=False), nn.BatchNorm2d(8), nn.ReLU(), nn.Flatten(), nn.Linear(8 * 4 * 4, 5))

    model, ema = make(), make()
    ema.load_state_dict(model.state_dict())
    flatten_model = lambda m: sum(map(flatten_model, m.children()), []) if len(list(m.children())) else [m]
    groups = [nn.Sequential(*flatten_model(model))]                      # get_layer_groups (train_sessd.py:163-164)
    opt_func = partial(torch.optim.Adam, betas=(0.9, 0.99), amsgrad=0.0)   # train_sessd.py:169
    opt = fo.OptimWrapper.create(opt_func, 3e-3, groups, wd=0.01, true_wd=True, bn_wd=True)
    total = 10
    sched = ls.OneCycle(opt, total, 0.003, [0.95, 0.85], 10.0, 0.4)
    params = list(model.parameters())
    out = dict(p0=flat(params))
    g = torch.Generator().manual_seed(1)
    lrs, moms, grads, ps, ts, norms = [], [], [], [], [], []
    for step in range(6):
        sched.step(step)                                                  # trainer_sessd.py:341-342
        lrs.append(float(opt.lr[-1] if isinstance(opt.lr, (list, tuple)) else opt.lr))
        moms.append(float(opt.mom[-1] if isinstance(opt.mom, (list, tuple)) else opt.mom))
        opt.zero_grad()
        scale = 40.0 if step % 2 == 0 else 0.01                           # alternate clipped / unclipped steps
        for p in params:
            p.grad = torch.randn(p.shape, generator=g) * scale
        grads.append(flat([p.grad for p in params]))
        n = torch.nn.utils.clip_grad_norm_(filter(lambda p: p.requires_grad, params), max_norm=35, norm_type=2)
        norms.append(float(n))
        opt.step()
        alpha = min(1 - 1 / (step + 1), 0.999)                           # trainer_sessd.py:316
        for ep, p in zip(ema.parameters(), model.parameters()):
            ep.data.mul_(alpha).add_(p.data, alpha=1 - alpha)             # :318 (add_(1 - alpha, param.data))
        ps.append(flat(params))
        ts.append(flat(list(ema.parameters())))
    out.update(lr=np.array(lrs), mom=np.array(moms), grads=np.stack(grads), params=np.stack(ps), teacher=np.stack(ts),
               norms=np.array(norms))
    # the schedule itself over a long run
    class Dummy:
        lr, mom = 0.0, 0.0
    d = Dummy()
    big = ls.OneCycle(d, 1000, 0.003, [0.95, 0.85], 10.0, 0.4)
    tab = []
    for s in range(0, 1000, 7):
        big.step(s)
        tab.append((s, d.lr, d.mom))
    out["onecycle_1000"] = np.array(tab, np.float64)
    np.savez_compressed(os.path.join(HERE, "train_ref.npz"), **out)
    print("train golden written:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
