"""Training-step host logic and its oracle, on CPU:
  * oracle/optim.py is pinned to tests/golden/train_ref.npz = the reference's OWN OptimWrapper(true_wd) + torch Adam +
    clip_grad_norm_ + OneCycle run from source (tests/golden/make_golden_train.py) and the EMA lines of the trainer;
  * sessd_hip.train.one_cycle against the reference schedule table;
  * FlatParams aliasing / autograd accumulation; the single flat all-reduce on two gloo ranks."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import optim as ooptim
from sessd_hip import train as strain


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "train_ref.npz"))


def test_optim_oracle_matches_reference_run(golden_dir):
    g = _golden(golden_dir)
    p = g["p0"].copy()
    teacher = g["p0"].copy()
    m, v = np.zeros_like(p), np.zeros_like(p)
    clipped = 0
    for step in range(g["grads"].shape[0]):
        norm, coef = ooptim.clip_coef(g["grads"][step], 35.0)
        assert abs(norm - g["norms"][step]) <= 1e-4 * g["norms"][step]
        clipped += coef < 1.0
        ooptim.adam_true_wd_ema_step(p, g["grads"][step].copy(), m, v, teacher, float(g["lr"][step]), 0.01, float(g["mom"][step]),
                                     0.99, 1e-8, step + 1, max_norm=35.0, alpha=ooptim.ema_alpha(step))
        assert np.allclose(p, g["params"][step], rtol=2e-6, atol=2e-7), (step, np.abs(p - g["params"][step]).max())
        assert np.allclose(teacher, g["teacher"][step], rtol=2e-6, atol=2e-7), step
    assert 0 < clipped < g["grads"].shape[0]  # both branches of the clip were exercised


def test_one_cycle_matches_reference_schedule(golden_dir):
    g = _golden(golden_dir)
    for s, lr, mom in g["onecycle_1000"]:
        got = strain.one_cycle(int(s), 1000)
        assert abs(got[0] - lr) <= 1e-12 + 1e-9 * lr and abs(got[1] - mom) <= 1e-12
    for step in range(len(g["lr"])):
        got = strain.one_cycle(step, 10)
        assert abs(got[0] - g["lr"][step]) < 1e-12 and abs(got[1] - g["mom"][step]) < 1e-12
    assert abs(strain.ema_alpha(0)) == 0.0 and strain.ema_alpha(10 ** 6) == 0.999


def test_flat_params_alias_and_accumulate():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.BatchNorm1d(7), torch.nn.Linear(7, 3))
    before = [p.detach().clone() for p in net.parameters()]
    fp = strain.FlatParams(net)
    assert fp.numel % 4 == 0 and all(o % 4 == 0 for o in fp.offsets)
    for p, b, o in zip(net.parameters(), before, fp.offsets):
        assert torch.equal(p.detach(), b) and p.data_ptr() == fp.data.data_ptr() + 4 * o
    x = torch.randn(6, 5)
    net(x).pow(2).sum().backward()
    g1 = fp.grad.clone()
    assert float(g1.abs().sum()) > 0
    for p, o in zip(net.parameters(), fp.offsets):
        assert p.grad.data_ptr() == fp.grad.data_ptr() + 4 * o and torch.equal(p.grad.reshape(-1), fp.grad[o:o + p.numel()])
    net(x).pow(2).sum().backward()  # accumulates in place into the same flat buffer
    assert torch.allclose(fp.grad, 2 * g1)
    fp.zero_grad()
    assert float(fp.grad.abs().sum()) == 0.0
    fp.data.mul_(0.5)  # an update of the flat buffer IS an update of the model
    for p, b in zip(net.parameters(), before):
        assert torch.allclose(p.detach(), 0.5 * b)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Linear(4, 3)
    fp = strain.FlatParams(net)
    fp.grad.copy_(torch.arange(fp.numel, dtype=torch.float32) * (rank + 1))
    strain.allreduce_flat(fp.grad)
    want = torch.arange(fp.numel, dtype=torch.float32) * (1 + 2) / 2.0
    q.put((rank, bool(torch.allclose(fp.grad, want)), bool(torch.equal(net.weight.grad.reshape(-1), fp.grad[:12]))))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_two_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok and alias for _, ok, alias in res), res


def test_sync_bn_switch_keeps_group_and_reduce_hook():
    """Round-4 advisor finding: toggling SyncBN (TrainStep does, around every iteration) forgot a configured process group /
    reduce hook. set_sync_bn keeps what it is not given; sync_bn_state / restore_sync_bn carry the whole state."""
    from sessd_hip import ops
    before = ops.sync_bn_state()
    try:
        f = lambda t: t
        ops.set_sync_bn(True, group="G", reduce_fn=f)
        ops.set_sync_bn(False)
        assert ops.sync_bn_state() == (False, "G", f)
        st = ops.sync_bn_state()
        ops.set_sync_bn(True, group=None)
        assert ops.sync_bn_state() == (True, None, f)
        ops.restore_sync_bn(st)
        assert ops.sync_bn_state() == (False, "G", f)
    finally:
        ops.restore_sync_bn(before)
