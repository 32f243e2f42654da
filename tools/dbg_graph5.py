import sys, torch
stage = sys.argv[1]
x = torch.randn(2, 1, 64, 96, device="cuda", requires_grad=True)
class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        ctx.g = torch.ones_like(a) * 3
        return (a * 3).sum().reshape(1).expand(2).contiguous()
    @staticmethod
    def backward(ctx, g):
        return ctx.g if stage == "saved" else ctx.g * g[0]
def step():
    if stage == "plain":
        y = (x * 2).sum()
    else:
        y = F.apply(x).sum()
    y.backward()
    return y
step(); torch.cuda.synchronize(); x.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
print(stage, "captured"); g.replay(); torch.cuda.synchronize(); print(stage, "replayed OK", float(out))
