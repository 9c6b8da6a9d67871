"""The ORACLE PointTransformer (oracle/ref_cpu.py, CPU) under torch.autocast -- the reference's own mixed-precision step
(model_trainer.py:75-76,157) -- against its fp32 run: mean / max |logit error| and parameter-gradient cosine, fp16 and bf16, at
fill_state_dict weights, at torch's default initialisation, and after 20 Adam steps.  The yardstick for the bf16 operand mode of
the HIP path (tests/test_gpu_parity.py: test_pointtransformer_bf16_mode_*).  Runs here (no GPU): python tools/pt_autocast_oracle.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from golden_util import cloud, fill_state_dict  # noqa: E402
from oracle import ref_cpu  # noqa: E402


def stats(tag, ref, x, gr, dt):
    for p in ref.parameters():
        p.grad = None
    y32 = ref(x)
    y32.backward(gr)
    g32 = torch.cat([p.grad.reshape(-1) for p in ref.parameters()]).double()
    for p in ref.parameters():
        p.grad = None
    with torch.autocast("cpu", dtype=dt):
        y16 = ref(x)
    y16.float().backward(gr)
    g16 = torch.cat([p.grad.reshape(-1) for p in ref.parameters()]).double()
    d = (y16.float() - y32).abs().detach()
    print("%-26s %-15s logit error mean %.4f max %.3f (logit scale %.3f)  gradient cosine %.4f" % (
        tag, str(dt), float(d.mean()), float(d.max()), float(y32.abs().mean()), float(g16 @ g32 / (g16.norm() * g32.norm()))))


x = torch.from_numpy(cloud(4310, 2, 6, 2048))
gr = torch.from_numpy(np.random.default_rng(4311).standard_normal((2, 4, 2048)).astype(np.float32))
ref = fill_state_dict(ref_cpu.PointTransformerCompatibility(6, 4), 803).train()
for dt in (torch.float16, torch.bfloat16):
    stats("fill_state_dict weights", ref, x, gr, dt)
torch.manual_seed(0)
ref = ref_cpu.PointTransformerCompatibility(6, 4).train()
for dt in (torch.float16, torch.bfloat16):
    stats("default initialisation", ref, x, gr, dt)
lab = torch.randint(0, 4, (2, 2048))
opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
for _ in range(20):
    opt.zero_grad()
    torch.nn.functional.cross_entropy(ref(x), lab).backward()
    opt.step()
for dt in (torch.float16, torch.bfloat16):
    stats("after 20 Adam steps", ref, x, gr, dt)
