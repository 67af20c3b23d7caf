"""Dump the whole-detector gradients of the test batch (tests/test_train_gpu.py::test_whole_model_gradients_vs_oracle) to a .pt
file, or compare two dumps: python scripts/dbg_grad_dump.py dump out.pt | compare a.pt b.pt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "se-ssd_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
if sys.argv[1] == "dump":
    import test_train_gpu as T
    dev = torch.device("cuda:0")
    model = T.configs.build_synthetic_detector(dev, seed=0)
    model.train()
    frames, ex = T._example(dev, (41, 42), 8000, 8000)
    preds = model.forward_preds(ex)
    loss = T._loss(preds); loss.backward()
    out = {n: p.grad.cpu() for n, p in model.named_parameters()}
    out["__loss__"] = loss.detach().cpu()
    for k, v in preds[0].items():
        if torch.is_tensor(v):
            out["__pred__" + k] = v.detach().cpu()
    torch.save(out, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = (a[k].double() - b[k].double())
        e = float(d.abs().max()) / max(1e-30, float(b[k].abs().max()))
        l2 = float(d.norm()) / max(1e-30, float(b[k].double().norm()))
        if e > 1e-6:
            print("%-40s max %.2e l2 %.2e" % (k, e, l2))
    print("compared", len(a), "tensors")
