"""Histogram of mirror-count levels and count values of a config-5-regime layer (development aid, GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimure_amd.synthetic import standard_sbm
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
net = standard_sbm(N=N, M=1000, L=1, K=3, C=2, avg_degree=5.0, eta=0.5, seed=0, device="cuda:0")
X = net.X[0]
hist = torch.zeros(64, dtype=torch.long, device="cuda")
xh = torch.zeros(64, dtype=torch.long, device="cuda")
for i in range(0, N, 100):
    a = X[i:i + 100]                       # [100, N, M]
    b = X[:, i:i + 100].transpose(0, 1)    # mirror: X[j, i, m] -> [100, N, M]
    nz = a > 0
    hist += torch.bincount(b[nz].long(), minlength=64)[:64]
    xh += torch.bincount(a[nz].long(), minlength=64)[:64]
tot = hist.sum().item()
print("entries", tot, "mirror-count levels:", [(y, round(100.0 * c / tot, 3)) for y, c in enumerate(hist.tolist()) if c])
print("count values:", [(y, round(100.0 * c / tot, 3)) for y, c in enumerate(xh.tolist()) if c])
