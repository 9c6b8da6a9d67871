import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, Ci, Co, L) in [(2, 30, 64, 5120), (2, 6, 64, 2560), (8, 64, 64, 40960), (8, 192, 1024, 2048), (2, 128, 64, 1024)]:
    x = torch.randn(B, Ci, L, device=dev); w = torch.randn(Co, Ci, device=dev) / Ci ** 0.5
    ref = torch.matmul(w.double(), x.double())
    y1 = torch.matmul(w, x)
    y2 = F.conv1d(x, w.unsqueeze(-1))
    xp = x.transpose(1, 2).contiguous()
    y3 = F.linear(xp, w).transpose(1, 2)
    y4 = torch.einsum("oc,bcl->bol", w, x)
    print((B, Ci, Co, L), "matmul", float((y1 - ref).abs().max()), "conv1d", float((y2 - ref).abs().max()),
          "linear_pm", float((y3 - ref).abs().max()), "einsum", float((y4 - ref).abs().max()))
