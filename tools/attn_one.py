import math, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
B, S = int(os.environ.get('A1_B', 8)), int(os.environ.get('A1_S', 2048))
nh, nkv, hd = int(os.environ.get('A1_NH', 16)), int(os.environ.get('A1_NKV', os.environ.get('A1_NH', 16))), int(os.environ.get('A1_HD', 128))
CAUSAL = os.environ.get('A1_CAUSAL', '1') == '1'
qkv = torch.randn(B * S, (nh + 2 * nkv) * hd, device="cuda").to(torch.bfloat16)
q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
for _ in range(3):
    o, lse = K.attn_fwd(q, k, v, B, S, nh, nkv, hd, 1 / math.sqrt(hd), CAUSAL)
if len(sys.argv) > 1:
    do = torch.randn(B * S, nh * hd, device="cuda").to(torch.bfloat16)
    d = torch.empty_like(qkv)
    for _ in range(2):
        K.attn_bwd(q, k, v, o, do, lse, d[:, :nh * hd], d[:, nh * hd:(nh + nkv) * hd], d[:, (nh + nkv) * hd:], B, S, nh, nkv, hd, 1 / math.sqrt(hd), CAUSAL)
torch.cuda.synchronize()
