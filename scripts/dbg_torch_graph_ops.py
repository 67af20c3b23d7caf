"""Which torch ops of the training iteration misbehave inside a captured graph on this stack? Each candidate is captured alone,
replayed 12 times on CHANGING inputs and compared with the eager result."""
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)

def check(name, make_inputs, fn, n=12, tol=2e-4):
    ins = make_inputs()
    for _ in range(2):
        fn(*ins)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = fn(*ins)
    outs = outs if isinstance(outs, (list, tuple)) else [outs]
    bad = 0
    for it in range(n):
        fresh = make_inputs()
        for a, b in zip(ins, fresh):
            a.detach().copy_(b.detach())
        g.replay()
        torch.cuda.synchronize()
        got = [o.detach().clone() for o in outs]
        want = fn(*ins)
        want = want if isinstance(want, (list, tuple)) else [want]
        torch.cuda.synchronize()
        for k, (a, b) in enumerate(zip(got, want)):
            err = float((a - b).abs().max()) / max(1e-12, float(b.abs().max()))
            if not err <= tol:
                bad += 1
                if bad <= 3:
                    print("   %s: replay %d output %d rel err %.3e (got max %.4g want max %.4g)" % (name, it, k, err, float(a.abs().max()), float(b.abs().max())))
    print("%-44s %s" % (name, "OK" if bad == 0 else "WRONG in %d output checks" % bad))

B, H, W = 2, 200, 176
check("mean of 1.4M elements", lambda: [torch.randn(B, 200, 176, 20, device=dev)], lambda x: x.mean())
check("sigmoid().mean() + abs().mean() + pow(2).mean()", lambda: [torch.randn(B, 200, 176, 4, device=dev), torch.randn(B, 200, 176, 14, device=dev)],
      lambda a, b: torch.sigmoid(a).mean() + a.abs().mean() + b.pow(2).mean())
check("sum over (0,2,3) of (2,22,200,176)", lambda: [torch.randn(B, 22, H, W, device=dev)], lambda x: x.sum((0, 2, 3)))

def conv_bwd(x, w, b, go):
    x = x.detach().requires_grad_(True); w = w.detach().requires_grad_(True); b = b.detach().requires_grad_(True)
    y = F.conv2d(x, w, b)
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), go)
    return [y.detach(), gx, gw, gb]
for co in (14, 2, 4):
    check("conv2d 1x1 128->%d fwd + grads (MIOpen)" % co,
          lambda co=co: [torch.randn(B, 128, H, W, device=dev), torch.randn(co, 128, 1, 1, device=dev) * 0.1, torch.randn(co, device=dev), torch.randn(B, co, H, W, device=dev)],
          conv_bwd)
check("conv2d 1x1 128->1 no bias fwd + grads", lambda: [torch.randn(B, 128, H, W, device=dev), torch.randn(1, 128, 1, 1, device=dev) * 0.1, torch.zeros(1, device=dev), torch.randn(B, 1, H, W, device=dev)], conv_bwd)
def softmax_blend(a, b, x0, x1, go):
    a = a.detach().requires_grad_(True); b = b.detach().requires_grad_(True)
    w = torch.softmax(torch.cat([a, b], 1), 1)
    y = x0 * w[:, 0:1] + x1 * w[:, 1:]
    ga, gb = torch.autograd.grad(y, (a, b), go)
    return [y.detach(), ga, gb]
check("softmax(cat) blend fwd + grads", lambda: [torch.randn(B, 1, H, W, device=dev), torch.randn(B, 1, H, W, device=dev), torch.randn(B, 128, H, W, device=dev),
                                                   torch.randn(B, 128, H, W, device=dev), torch.randn(B, 128, H, W, device=dev)], softmax_blend)
check("permute(0,2,3,1).contiguous()", lambda: [torch.randn(B, 14, H, W, device=dev)], lambda x: x.permute(0, 2, 3, 1).contiguous())
check("_foreach_copy_ of 100 tensors", lambda: [torch.randn(1000 + 37 * i, device=dev) for i in range(100)],
      lambda *xs: [torch._foreach_copy_([torch.empty_like(x) for x in xs], list(xs))[0]] if False else [torch.stack([x.sum() for x in xs[:3]])])
